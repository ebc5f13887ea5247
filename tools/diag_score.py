#!/usr/bin/env python3
"""Run the score kernel alone on a large random problem (long dispatches for PMC / clock
diagnosis).  Development tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8 * 1048576
D = int(sys.argv[2]) if len(sys.argv) > 2 else 100
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
from segmentalist_amd.device import DeviceCorpus, DeviceKMeans  # noqa: E402

rs = np.random.RandomState(0)
X = rs.randn(n, D).astype(np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
if os.environ.get("DIAG_ZERO"):            # DVFS check: same instruction stream on trivial operands
    X[:] = 0.0
corpus = DeviceCorpus(X)
assign = -np.ones(n, dtype=np.int64)
assign[:K] = np.arange(K)
dk = DeviceKMeans(corpus, K, assign, X[rs.randint(0, n, K)])
torch.cuda.synchronize()
for _ in range(2):
    dk.score_rows()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dk.score_rows()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
print("score n=%d D=%d K=%d: %.3f ms  %.1f TFLOP/s" % (n, D, K, ms, 2.0 * n * K * D / ms / 1e9))
