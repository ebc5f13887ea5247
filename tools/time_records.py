#!/usr/bin/env python3
"""Cost of the per-sweep record keeping against the sweep itself: segment(n) / gibbs_sample(n) in
batch mode (sweep + record metrics) vs batch_sweep_async alone, at BASELINE configs[2], configs[4], configs[1]."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, kmeans_acoustic_wordseg as kaw, unigram_acoustic_wordseg as uaw
from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
from segmentalist_amd.niw import NIW
from segmentalist_amd.synth import make_corpus


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    n = 10
    corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    seg.segment(2)
    def sweeps(k):
        for _ in range(k): seg.batch_sweep_async()
    print("k-means c3: batch_sweep_async %.3f ms, segment() per iteration %.3f ms" % (timed(sweeps, 50), timed(seg.segment, n)))
    fixed = FixedVarPrior(0.002 * np.ones(100), np.zeros(100), 0.002 / 0.05 * np.ones(100))
    kw = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0, sync="batch")
    random.seed(0); np.random.seed(0)
    seg = baw.BigramAcousticWordseg(1000, fixed, {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                    covariance_type="fixed", fb_type="unigram", score_precision="f16", **kw)
    seg.gibbs_sample(2)
    def sweeps2(k):
        for _ in range(k): seg.batch_sweep_async()
    a = timed(sweeps2, 10)
    b = timed(seg.gibbs_sample, n)
    os.environ["SEGK_HOST_METRICS"] = "1"
    c = timed(seg.gibbs_sample, 2)
    os.environ["SEGK_HOST_METRICS"] = "0"
    print("bigram c5 (f16): batch_sweep_async %.3f ms, gibbs_sample() per iteration %.3f ms (host metrics: %.1f ms)" % (a, b, c))
    corpus = make_corpus(1000, 39, 100, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, 100, NIW(np.zeros(39), 0.05, 42, 0.002 * 42 * np.ones(39)), *corpus,
                                     covariance_type="diag", fb_type="standard", **kw)
    seg.gibbs_sample(2)
    def sweeps3(k):
        for _ in range(k): seg.batch_sweep_async()
    a = timed(sweeps3, 20)
    b = timed(seg.gibbs_sample, n)
    os.environ["SEGK_HOST_METRICS"] = "1"
    c = timed(seg.gibbs_sample, 2)
    print("fbgmm diag c2: batch_sweep_async %.3f ms, gibbs_sample() per iteration %.3f ms (host metrics: %.1f ms)" % (a, b, c))


if __name__ == "__main__":
    main()
