#!/usr/bin/env python3
"""
Batch-synchronous sweeps against the reference's sequential chain, same corpus, same seed (SURVEY section 7 "Two execution
modes"): objective per sweep, tokens and components -- the statistical tie between the throughput
mode (a different Markov chain, specified only in this repository) and the reference's behaviour.

    python tools/batch_vs_sequential.py [--utts 2000] [--sweeps 10] [--out profiles/r02_batch_vs_sequential.json]
"""
import argparse, json, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def kmeans_curves(n_utt, n_sweeps, D=100, K=1000, minibatches=()):
    """sequential, batch (whole-sweep statistics) and, for every B in `minibatches`, "minibatch_B" (n_batches = B)."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=0, N=20, n_slices_max=6)
    out = {}
    for name, kw in [("sequential", dict(sync="sequential")), ("batch", dict(sync="batch"))] + \
            [("minibatch_%d" % b, dict(sync="batch", n_batches=b)) for b in minibatches]:
        random.seed(0); np.random.seed(0)
        seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=6, init_am_assignments="spread", **kw)
        rec = seg.segment(n_sweeps)
        out[name] = {k: [float(v) for v in rec[k]] for k in ("sum_neg_len_sqrd_norm", "sum_neg_sqrd_norm", "components", "n_tokens")}
        out[name]["sample_time"] = [float(v) for v in rec["sample_time"]]
    return out


def fbgmm_curves(kind, n_utt, n_sweeps, D, K):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=0, N=20, n_slices_max=6)
    out = {}
    for sync in ("sequential", "batch"):
        random.seed(0); np.random.seed(0)
        kw = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                  init_am_assignments="rand", time_power_term=1.0, sync=sync)
        if kind == "bigram":
            prior = FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
            seg = baw.BigramAcousticWordseg(K, prior, {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                            covariance_type="fixed", fb_type="unigram", **kw)
        else:
            prior = NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type="diag", fb_type="standard", **kw)
        rec = seg.gibbs_sample(n_sweeps)
        out[sync] = {k: [float(v) for v in rec[k]] for k in ("log_marg", "log_marg*length", "components", "n_tokens")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2000)
    ap.add_argument("--sweeps", type=int, default=10)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {"kmeans_c3_shape": dict(utterances=a.utts, D=100, K=1000, **kmeans_curves(a.utts, a.sweeps, minibatches=(2, 4, 8))),
           "fbgmm_diag_c2": dict(utterances=1000, D=39, K=100, **fbgmm_curves("diag", 1000, a.sweeps, 39, 100)),
           "bigram_fixed": dict(utterances=500, D=39, K=100, **fbgmm_curves("bigram", 500, a.sweeps, 39, 100))}
    txt = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(txt)
    for name, r in res.items():
        key = "sum_neg_len_sqrd_norm" if "kmeans" in name else "log_marg"
        print(name, key)
        for i, (s, b) in enumerate(zip(r["sequential"][key], r["batch"][key])):
            print("  sweep %2d  sequential %14.4f  batch %14.4f  rel diff %+.4f   K %4d / %4d   tokens %6d / %6d"
                  % (i, s, b, (b - s) / abs(s), r["sequential"]["components"][i], r["batch"]["components"][i],
                     r["sequential"]["n_tokens"][i], r["batch"]["n_tokens"][i]))
        for mb in sorted(k for k in r if k.startswith("minibatch_")):
            print("  %-12s last sweep %14.4f  rel diff to sequential %+.4f   K %4d   tokens %6d   %.2f ms per sweep"
                  % (mb, r[mb][key][-1], (r[mb][key][-1] - r["sequential"][key][-1]) / abs(r["sequential"][key][-1]),
                     r[mb]["components"][-1], r[mb]["n_tokens"][-1], 1e3 * float(np.median(r[mb]["sample_time"][1:]))))


if __name__ == "__main__":
    main()
