#!/usr/bin/env python3
"""Development: duration of the pre-filter launch (library-recorded events) under the SEGK_H1_ABL timing ablations."""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for _ in range(5): seg.batch_sweep_async()
torch.cuda.synchronize()
dk = seg._dk
L, ctx = _abi.lib(), _abi.ctx()
for abl in ("0", "1", "2", "3", "0"):
    os.environ["SEGK_H1_ABL"] = abl
    _abi.check(L.segk_profile_enable(ctx, 1))
    for _ in range(12):
        _abi.check(L.segk_kmeans_filter(ctx, dk._cp(), C.byref(dk.m), None, 0, dk.corpus.n_emb, C.byref(dk.cand), _abi.stream()))
        torch.cuda.synchronize()
    ms = (C.c_float * 64)(); rows = (C.c_int64 * 64)()
    got = L.segk_profile_read(ctx, ms, rows, 64)
    _abi.check(L.segk_profile_enable(ctx, 0))
    v = np.array(ms[2:got])
    print("SEGK_H1_ABL=%s  pre-filter launch: median %.1f us  min %.1f  (%d launches)" % (abl, 1e3 * np.median(v), 1e3 * v.min(), len(v)))
