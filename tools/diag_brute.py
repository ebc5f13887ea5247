#!/usr/bin/env python3
"""Development: time segk_kmeans_resolve (the full scan of the queued rows) alone, for several queue lengths."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from segmentalist_amd import _abi
from segmentalist_amd.device import DeviceCorpus, DeviceKMeans, ptr
n, D, K = 131250, 100, 1000
rs = np.random.RandomState(0)
X = rs.randn(n, D).astype(np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
corpus = DeviceCorpus(X)
assign = -np.ones(n, dtype=np.int64)
assign[:K] = np.arange(K)
dk = DeviceKMeans(corpus, K, assign, X[rs.randint(0, n, K)])
dk.score_rows()
torch.cuda.synchronize()
print("queued by the filter:", int(dk.cand_count.item()))
dk.cand_queue[:4096] = torch.arange(4096, dtype=torch.int32, device="cuda")
L = _abi.lib()
for nq in (0, 8, 64, 200, 1500, 4096):
    dk.cand_count.fill_(nq)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _abi.check(L.segk_kmeans_resolve(dk._ctx, dk._cp(), C.byref(dk.m), None, 0, n, C.byref(dk.cand), ptr(dk.status), _abi.stream()))
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print("queue %5d rows: resolve %.1f us (median of 12, event to event)" % (nq, float(np.median(ts))))
