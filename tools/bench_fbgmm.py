#!/usr/bin/env python3
"""Throughput of the sequential (reference-chain) FBGMM / bigram Gibbs drivers on BASELINE
config 2 shapes (1 000 utterances, D = 39, K = 100) -- development measurement, not bench.py."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=39)
    ap.add_argument("--K", type=int, default=100)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--cpu-utts", type=int, default=40)
    ap.add_argument("--which", default="diag,fixed,bigram")
    ap.add_argument("--sync", default="sequential")
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--precision", default="f64")
    args = ap.parse_args()
    import torch
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    from oracle import np_oracle as no
    D, K = args.dim, args.K
    corpus = make_corpus(args.utts, D, K, seed=0, N=20, n_slices_max=6)
    keys = sorted(corpus[0])[:args.cpu_utts]
    sub = tuple({k: d[k] for k in keys} for d in corpus)
    fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
    diag = (np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    kw = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0)
    okw = dict(kw)
    if args.sync == "batch":
        kw.update(sync="batch", n_gibbs_blocks=args.blocks)
    pk = dict(score_precision=args.precision) if args.sync == "batch" else {}
    lm = {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}
    for which in args.which.split(","):
        random.seed(0)
        np.random.seed(0)
        t0 = time.perf_counter()
        if which == "diag":
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(*diag), *corpus, covariance_type="diag",
                                             fb_type="standard", **kw)
        elif which == "fixed":
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, FixedVarPrior(*fixed), *corpus,
                                             covariance_type="fixed", fb_type="standard", **kw, **pk)
        else:
            seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), lm, *corpus, covariance_type="fixed",
                                            fb_type="unigram", **kw, **pk)
        torch.cuda.synchronize()
        t_init = time.perf_counter() - t0
        if args.sync == "batch":
            st = []
            for _ in range(args.sweeps + 1):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                seg.batch_sweep_async()
                torch.cuda.synchronize()
                st.append(time.perf_counter() - t1)
            seg._df.check_status()
            st = st[1:]
            cnt, tot, occ = seg._get_sweeper().totals()
            rec = {"components": [occ]}
        else:
            rec = seg.gibbs_sample(args.sweeps)
            st = rec["sample_time"]
        print("%-6s init %.2f s; sweep times %s s -> %.2f sweeps/s (%.1f us/utterance); K=%d"
              % (which, t_init, ["%.3f" % x for x in st], 1.0 / min(st), 1e6 * min(st) / args.utts,
                 rec["components"][-1]), flush=True)
        # oracle on a bounded sample (--cpu-utts 0: device timings only)
        if not keys:
            continue
        random.seed(0)
        np.random.seed(0)
        if which == "diag":
            ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, no.NIW(*diag), *sub, covariance_type="diag",
                                            fb_type="standard", **okw)
        elif which == "fixed":
            ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, no.FixedVarPrior(*fixed), *sub,
                                            covariance_type="fixed", fb_type="standard", **okw)
        else:
            ref = no.BigramAcousticWordseg(K, no.FixedVarPrior(*fixed), lm, *sub, covariance_type="fixed",
                                           fb_type="unigram", **okw)
        t0 = time.perf_counter()
        for i in range(len(keys)):
            ref.gibbs_sample_i(i)
        dt = time.perf_counter() - t0
        print("%-6s oracle (1 core): %.2f ms/utterance -> %.4f sweeps/s" % (which, 1e3 * dt / len(keys),
                                                                          len(keys) / dt / args.utts), flush=True)


if __name__ == "__main__":
    main()
