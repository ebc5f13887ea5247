#!/usr/bin/env python3
"""Development: where does the forced pre-filter disagree with the C oracle?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SEGK_SCORE_PRE"] = "1"
import numpy as np
import torch
from oracle import c_oracle as co
from segmentalist_amd.kmeans_components import KMeansComponents
D, K, n, scale = 100, 1000, 4096, 1.0
rs = np.random.RandomState(D * 1000 + K + 1)
K_true = max(2, K // 2)
mu = rs.randn(K_true, D)
X = mu[rs.randint(0, K_true, n)] + 0.3 * rs.randn(n, D)
X /= np.linalg.norm(X, axis=1, keepdims=True)
X = (X * scale).astype(np.float32)
means = (mu[rs.randint(0, K_true, K)] + 0.05 * rs.randn(K, D))
means /= np.linalg.norm(means, axis=1, keepdims=True)
means = (means * scale).astype(np.float32)
means[K // 2] = means[1]
means[7] = means[6]
X[5] = means[1]
X[9] = means[6]
np.random.seed(0)
c = KMeansComponents(X, np.zeros(n, dtype=int), K)
c.dev.means.copy_(torch.from_numpy(means).to(c.dev.means.device))
c.dev.prepare()
mx, am, nb = c.dev.exact_max(np.arange(n))
wmx, wam = co.kmeans_max_argmax(means, X)
bad = np.nonzero(am != wam)[0]
print("n bad", len(bad), "nbrute", nb)
f = X.astype(np.float64) @ means.astype(np.float64).T - 0.5 * (means.astype(np.float64) ** 2).sum(1)
cf = c.dev.cand_f.cpu().numpy()
for i in bad[:20]:
    top = np.sort(f[i])[::-1]
    print(i, "got", am[i], "want", wam[i], "f got/want", f[i, am[i]], f[i, wam[i]], "gap12", top[0] - top[1], "cand_f", cf[i], "mx", mx[i], wmx[i])
flag = np.nonzero(am >= (1 << 30))[0]
print("rows still marked pending:", len(flag), flag[:40])
