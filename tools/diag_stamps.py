#!/usr/bin/env python3
"""Development (library built with -DSEGK_STAMP): s_memtime stamps of the pre-filter kernel's phases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
n, D, K = 1048576, 100, 1000
from segmentalist_amd.device import DeviceCorpus, DeviceKMeans
rs = np.random.RandomState(0)
X = rs.randn(n, D).astype(np.float32)
X /= np.linalg.norm(X, axis=1, keepdims=True)
corpus = DeviceCorpus(X)
assign = -np.ones(n, dtype=np.int64)
assign[:K] = np.arange(K)
dk = DeviceKMeans(corpus, K, assign, X[rs.randint(0, n, K)])
st = torch.zeros(65536 + 8 * 8192, dtype=torch.int64, device="cuda")
os.environ["SEGK_STAMP_PTR"] = hex(st.data_ptr())
for _ in range(3):
    dk.score_rows()
torch.cuda.synchronize()
allst = st.cpu().numpy()
s = allst[:65536].reshape(-1, 8)
s = s[s[:, 0] > 0]
print("workgroups", len(s))
t0 = s[:, 0].min()
d = lambda a, b: (s[:, b] - s[:, a])
for name, a, b in (("prologue+stage0", 0, 1), ("tile loop", 1, 2), ("epilogue", 2, 3), ("tile15 sync wait", 4, 5), ("total", 0, 3)):
    v = d(a, b)
    print("%-18s median %8d  p10 %8d  p90 %8d  (s_memtime ticks)" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
print("kernel span", s[:, 3].max() - t0, "start spread of first 512", np.sort(s[:, 0])[511] - t0)
order = np.argsort(s[:, 0])
print("starts (rel) at ranks 0,511,512,1023,1024,2047:", [int(s[order[i], 0] - t0) for i in (0, 511, 512, 1023, 1024, len(s) - 1)])

p = allst[65536:].reshape(-1, 8)
p = p[p[:, 0] > 0]
print("pair kernel waves", len(p))
for name, a, b in (("wait loads + LDS writes", 0, 1), ("issue next loads + meta", 1, 2), ("score + store", 2, 3), ("step", 0, 3)):
    v = p[:, b] - p[:, a]
    print("%-26s median %8d  p10 %8d  p90 %8d" % (name, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
