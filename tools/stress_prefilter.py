#!/usr/bin/env python3
"""Development: randomized parity sweep of the forced one-product pre-filter path against the C oracle
(max / argmax bit for bit) over dimensions, component counts, scales and tie structures."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SEGK_SCORE_PRE"] = "1"
import numpy as np
import torch
from oracle import c_oracle as co
from segmentalist_amd.kmeans_components import KMeansComponents

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(n_cases):
    D = 4 * rs.randint(2, 33)
    K = int(rs.choice([2, 3, 31, 32, 33, 64, 100, 257, 513, 1000, 1200]))
    n = int(rs.choice([1, 17, 255, 256, 257, 1000, 4096, 20000]))
    scale = float(10.0 ** rs.uniform(-4, 4))
    kind = rs.randint(0, 4)
    K_true = max(1, K // 2)
    mu = rs.randn(K_true, D)
    X = mu[rs.randint(0, K_true, n)] + 0.3 * rs.randn(n, D)
    if kind == 1:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    means = mu[rs.randint(0, K_true, K)] + (0.0 if kind == 2 else 0.05) * rs.randn(K, D)   # kind 2: many exact duplicates
    if kind == 3:
        X = X * 10.0 ** rs.uniform(-6, 0, size=X.shape)
        means = means * 10.0 ** rs.uniform(-6, 0, size=means.shape)
    X = (X * scale).astype(np.float32)
    means = (means * scale).astype(np.float32)
    if n > 3 and K > 2:
        X[1] = means[K - 1]
        X[2] = means[0]
    np.random.seed(0)
    c = KMeansComponents(X, np.zeros(n, dtype=int), K)
    c.dev.means.copy_(torch.from_numpy(means).to(c.dev.means.device))
    c.dev.prepare()
    mx, am, nb = c.dev.exact_max(np.arange(n))
    wmx, wam = co.kmeans_max_argmax(means, X)
    ok = np.array_equal(am, wam) and np.array_equal(mx, wmx)
    bad += not ok
    print("case %2d D=%3d K=%4d n=%5d scale=%.1e kind=%d full-scan rows %5d %s" % (case, D, K, n, scale, kind, nb, "ok" if ok else "MISMATCH"))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
