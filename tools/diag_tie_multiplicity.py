#!/usr/bin/env python3
"""Development: how many components lie within the split-precision margin of the best one, for the rows the
full scan gets (shard-sized and full corpus)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
for utts in (10000,):
    corpus = make_corpus(utts, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    for _ in range(6):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    dk = seg._dk
    nq = int(dk.cand_count.item())
    q = dk.cand_queue[:nq].cpu().numpy()
    M = dk.means.cpu().numpy().astype(np.float64)
    X = np.concatenate([corpus[0][k] for k in sorted(corpus[0])]).astype(np.float64)[q]
    f = X @ M.T - 0.5 * (M * M).sum(1)
    top = f.max(1, keepdims=True)
    Mmax = np.sqrt((M * M).sum(1).max()); xn = np.linalg.norm(X, axis=1)
    tau = 1.25 * (2 * (1.02 * 128 + 16) * 2.0 ** -24 * (xn * Mmax + 0.5 * Mmax ** 2) + (100 / 8 + 13) * 2.0 ** -24 * (xn + Mmax) ** 2)
    cnt = (f >= top - tau[:, None]).sum(1)
    K = int(seg.acoustic_model.components.K)
    best = f.argmax(1)
    print("utts", utts, "queued", nq, "K", K, "contenders histogram (1,2,3,4,5+):", [int((cnt == c).sum()) for c in (1, 2, 3, 4)], int((cnt >= 5).sum()),
          "| rows whose best component is inactive:", int((best >= K).sum()))
    # exact duplicates among the means
    _, inv, counts = np.unique(M.round(12), axis=0, return_inverse=True, return_counts=True)
    print("   duplicate mean rows:", int((counts > 1).sum()), "groups covering", int(counts[counts > 1].sum()), "rows")
    print("   queued rows >= 1048576:", int((q >= 1048576).sum()), " single-contender rows among them:", int(((q >= 1048576) & (cnt == 1)).sum()),
          " single-contender rows below:", int(((q < 1048576) & (cnt == 1)).sum()))
    cf = dk.cand_f.cpu().numpy()[q]
    sel = cnt == 1
    print("   filter gap / tau of single-contender rows (median, min):", float(np.median((cf[sel, 0] - cf[sel, 1]) / tau[sel])), float(((cf[sel, 0] - cf[sel, 1]) / tau[sel]).min()),
          " true gap / tau (median):", float(np.median((np.sort(f[sel], axis=1)[:, -1] - np.sort(f[sel], axis=1)[:, -2]) / tau[sel])))
