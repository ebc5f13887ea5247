"""How often does a row's argmax component change from one batch sweep to the next (after the relabelling of
clean_components)?  Sizing question for a score path that VERIFIES the previous sweep's winner against the dense
filter values instead of tracking an index through the top-2 drain: rows whose hint fails fall back to the full path.
Prints, per sweep: rows whose argmax differs from the remapped previous one, rows the pre-filter could not decide,
rows of the full scan."""
import ctypes as C
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    corpus = make_corpus(n_utt, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0)
    np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    dk = seg._dk
    prev = remap = None
    for sw in range(14):
        seg.batch_sweep_async()
        torch.cuda.synchronize()
        out = (C.c_int32 * 2)()
        _abi.check(_abi.lib().segk_kmeans_stage_counts(_abi.ctx(), C.byref(dk.cand), out, _abi.stream()))
        new = dk.cand_k.clone()
        K = int(dk.K.item())
        line = "sweep %2d  K %4d  undecided by the pre-filter %6d (%.2f %%)  full scan %5d" % (
            sw, K, out[0], 100.0 * out[0] / new.numel(), out[1])
        if prev is not None:
            hint = remap[prev.long()]
            ch = (hint != new)
            line += "  argmax changed %7d (%.2f %%)  [without remap %.2f %%]" % (
                int(ch.sum().item()), 100.0 * ch.float().mean().item(), 100.0 * (prev != new).float().mean().item())
        print(line, flush=True)
        prev, remap = new, dk.remap.clone()


if __name__ == "__main__":
    main()
