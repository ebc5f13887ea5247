#!/usr/bin/env python3
"""Per-stage timing of one batch sweep on the bench workload (HIP events on the launch stream),
plus the score kernel alone at a few row counts.  Development tool, not part of the product."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(t)), float(t.min())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--K", type=int, default=1000)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--score-only", action="store_true")
    args = ap.parse_args()
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(args.utts, args.dim, args.K, seed=0, N=20, n_slices_max=6)
    random.seed(0)
    np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(args.K, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    dk = seg._dk
    sw = seg._get_sweeper()
    n_emb = seg._corpus.n_emb
    for _ in range(args.sweeps):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    dk.check_status()
    print("K after %d sweeps: %d" % (args.sweeps, seg.acoustic_model.components.K))

    # score kernel alone at several row counts
    for nb in ("2", "1"):
        os.environ["SEGK_SCORE_NB"] = nb
        for n in [n_emb, 1048576, 524288, 393216, 262144, 131072, 65536, 16384]:
            if n > n_emb:
                continue
            med, mn = timeit(lambda: dk.score_rows(row0=0, n=n))
            tf = 2.0 * n * args.K * args.dim / (med * 1e-3) / 1e12
            print("NB=%s score n=%8d  median %.3f ms  min %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)"
                  % (nb, n, med, mn, tf, 100 * tf / 157.3))
    del os.environ["SEGK_SCORE_NB"]
    for dbg in (0, 2, 4, 8, 6, 12, 14):      # timing-only ablations: 2 no barrier, 4 no drain, 8 no staging
        os.environ["SEGK_SCORE_DBG"] = str(dbg)
        for n in (1048576, 131072):
            med, mn = timeit(lambda: dk.score_rows(row0=0, n=n))
            tf = 2.0 * n * args.K * args.dim / (med * 1e-3) / 1e12
            print("dbg=%2d n=%8d  median %.3f ms  %.1f TFLOP/s (%.1f%%)" % (dbg, n, med, tf, 100 * tf / 157.3))
    os.environ["SEGK_SCORE_DBG"] = "0"
    if args.score_only:
        return

    # stage by stage (state is advanced between stages by running real sweeps)
    import ctypes as C
    from segmentalist_amd import _abi
    from segmentalist_amd._abi import check, ptr
    L, ctx, cp, mp = dk._L, dk._ctx, dk._cp(), C.byref(dk.m)
    pt = sw.part
    st = _abi.stream()
    dk.status.zero_()
    dk.score_rows()
    med, mn = timeit(lambda: dk.segment(seg._dev_bounds, 0, 6, 0.0), reps=5, warm=1)
    torch.cuda.synchronize()
    nb = int(dk.status[1].item())
    print("segment: median %.3f ms min %.3f ms; brute-forced spans per launch: %.0f of %d (%.3f%%)"
          % (med, mn, nb / 6.0, n_emb, 100.0 * nb / 6.0 / n_emb))
    stages = [
        ("score", lambda: dk.score_rows()),
        ("segment", lambda: dk.segment(seg._dev_bounds, 0, 6, 0.0)),
        ("prepare", lambda: dk.prepare()),
    ]
    for name, fn in stages:
        med, mn = timeit(fn, reps=5, warm=1)
        print("%-10s median %.3f ms  min %.3f ms" % (name, med, mn))
    # batch statistics stages one by one (state left as the last full sweep produced it; finalize relabels new_k
    # in place, so it is timed once on a state its partials call has just prepared)
    def partials():
        check(L.segk_kmeans_batch_partials(ctx, cp, mp, ptr(sw.blk_lo), pt.nbl, ptr(dk.new_tok), ptr(dk.new_k),
                                           ptr(dk.n_flag), ptr(dk.out_total), ptr(sw.sorted), ptr(sw.koff),
                                           ptr(sw.pack), sw.cap, ptr(dk.out_scalars), st))

    def back():
        dk.score_rows()
        dk.segment(seg._dev_bounds, 0, 6, 0.0)
        partials()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sw._enqueue_back()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    med, mn = timeit(partials, reps=5, warm=1)
    print("%-10s median %.3f ms  min %.3f ms" % ("partials", med, mn))
    ts = sorted(back() for _ in range(5))
    print("%-10s median %.3f ms  min %.3f ms" % ("finalize+post", ts[2], ts[0]))
    for dbg in (1, 2, 4, 7):       # timing-only ablations inside the partials kernel
        os.environ["SEGK_PART_DBG"] = str(dbg)
        med, mn = timeit(partials, reps=5, warm=1)
        print("partials dbg=%d median %.3f ms" % (dbg, med))
    os.environ["SEGK_PART_DBG"] = "0"
    partials()
    med, mn = timeit(lambda: seg.batch_sweep_async(), reps=10, warm=2)
    print("full sweep: median %.3f ms  min %.3f ms" % (med, mn))
    t0 = time.perf_counter()
    for _ in range(20):
        seg.batch_sweep_async()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue per sweep %.1f us; total %.3f ms/sweep" % (1e6 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))


if __name__ == "__main__":
    main()
