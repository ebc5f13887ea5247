#!/usr/bin/env python3
"""Development: the first sweeps of a fresh chain on the bench corpus, for a kernel trace
(rocprofv3 --kernel-trace -- python3 tools/early_timeline.py [n_sweeps]); tools/early_timeline_print.py lists the kernels."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import torch
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
    for rep in range(2):          # the second chain is the one to read: the context's workspaces exist
        random.seed(0)
        np.random.seed(0)
        seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
        torch.cuda.synchronize()
        for _ in range(n_sweeps):
            seg.batch_sweep_async()
            torch.cuda.synchronize()
        seg._dk.check_status()
        del seg


if __name__ == "__main__":
    main()
