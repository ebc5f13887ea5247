"""Development: rows the pre-filter passes to its second stage and rows left for the full scan, bench workload."""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentalist_amd import _abi
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
for utts in (1250, 10000):
    corpus = make_corpus(utts, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    for it in range(8):
        seg.batch_sweep_async()
        counts = (C.c_int32 * 2)()
        _abi.check(_abi.lib().segk_kmeans_stage_counts(_abi.ctx(), C.byref(seg._dk.cand), counts, _abi.stream()))
        print(utts, "sweep", it, "second stage rows", counts[0], "full scan rows", counts[1], "K", int(seg.acoustic_model.components.K))
