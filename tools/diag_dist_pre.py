"""Development: repeat the 4-rank run of tests/test_gpu_dist.py::test_batch_mode_with_the_prefilter_forced... and, when its
result differs from the single-rank reference, say where (an intermittent mismatch was seen once in ~8 full suites)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_dist import run  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
worlds = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 2]
extra_env = dict(kv.split("=", 1) for kv in sys.argv[3:])
d = tempfile.mkdtemp()
ref = run(1, os.path.join(d, "p0.npz"), env_extra={"SEGK_SCORE_PRE": "0"})
bad = 0
for it in range(n):
    for world in worlds:
        got = run(world, os.path.join(d, "p%d.npz" % world), env_extra=dict({"SEGK_SCORE_PRE": "1"}, **extra_env))
        for k in ref.files:
            if not np.array_equal(ref[k], got[k]):
                bad += 1
                a, b = np.asarray(ref[k]), np.asarray(got[k])
                idx = np.flatnonzero(a.ravel() != b.ravel()) if a.shape == b.shape else []
                if k in ("totals", "n_tokens"):
                    print("iteration %d world %d key %s: ref %s got %s" % (it, world, k, a.ravel(), b.ravel()), flush=True)
    print("iteration", it, "done", flush=True)
print("mismatching keys:", bad)
