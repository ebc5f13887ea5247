#!/usr/bin/env python3
"""Host->device upload time of the bench corpus (X: 1.05 M x 100 float32 = 420 MB) from pageable
and from pinned host memory; used for the PCIe-inclusive note in profiles/README.md."""
import time

import numpy as np
import torch

n, d = 1050000, 100
x = np.random.RandomState(0).randn(n, d).astype(np.float32)
t = torch.from_numpy(x)
tp = t.pin_memory()
dev = torch.empty((n, d), dtype=torch.float32, device="cuda")
for name, src in (("pageable", t), ("pinned", tp)):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%s: %.2f ms  (%.1f GB/s)" % (name, 1e3 * best, x.nbytes / best / 1e9))
