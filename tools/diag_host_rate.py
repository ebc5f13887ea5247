#!/usr/bin/env python3
"""Development: is a batch sweep bound by the host (Python + launch calls) or by the device?  Enqueues `reps` sweeps without
waiting and reports the host's time per sweep (until the last enqueue returns) next to the wall time per sweep (until the
device has finished).  usage: diag_host_rate.py [n_utterances] [reps]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
corpus = make_corpus(n_utt, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for _ in range(20): seg.batch_sweep_async()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for _ in range(reps): seg.batch_sweep_async()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("utterances %d: host %.1f us per sweep, wall %.1f us per sweep (device idle at the end of the enqueue loop for %.1f us per sweep)" % (
        n_utt, 1e6 * (t1 - t0) / reps, 1e6 * (t2 - t0) / reps, 1e6 * (t2 - t1) / reps))
