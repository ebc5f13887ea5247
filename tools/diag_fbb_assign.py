#!/usr/bin/env python3
"""Timing ablations of the batch assign kernel (development tool)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from segmentalist_amd import bigram_acoustic_wordseg as baw
from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
from segmentalist_amd.synth import make_corpus
from segmentalist_amd._abi import check, ptr
D, K = 100, 1000
corpus = make_corpus(10000, D, K, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                covariance_type="fixed", fb_type="unigram", n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                beta_sent_boundary=-1, sync="batch")
seg.batch_sweep_async(); torch.cuda.synchronize()
sw = seg._get_sweeper(); df = seg._df
L, ctx, cp, fp, bp, st = sw._args()
b = 3
check(L.segk_fbb_prepare(ctx, cp, fp, bp, b, st))
def run():
    check(L.segk_fbb_assign(ctx, cp, fp, bp, sw.s_lo, sw.s_n, b, sw._n_utts[b], 7, 1.0, ptr(df.new_tok), ptr(df.n_new), None, 0, st))
for dbg in (0, 1, 2, 4, 3, 7):
    os.environ["SEGK_FBB_DBG"] = str(dbg)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print("dbg=%d: %.3f ms" % (dbg, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
print("tokens in block:", int(df.n_new[sw.utt_range_np[0,b,0]:sw.utt_range_np[0,b,1]].sum().item()) , "of slice 0")
