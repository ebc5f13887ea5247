#!/usr/bin/env python3
"""Development: phases of k_batch_finalize (SEGK_TSTAMP(3, p), s_memrealtime at 100 MHz) in sweep S of a fresh chain -- the sweeps
with hundreds of flagged tokens (new components founded).  Needs the stamp build (see tools/diag_tail_stamps.py).
usage: diag_finalize_early.py [S ...]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "build_stamp", "libsegk_stamp.so")
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
sweeps = [int(a) for a in sys.argv[1:]] or [2, 3]
corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
st = torch.zeros(8 * 1024 * 8, dtype=torch.int64, device="cuda")
os.environ["SEGK_TSTAMP_PTR"] = hex(st.data_ptr())
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for s in range(1, max(sweeps) + 1):
    st.zero_()
    seg.batch_sweep_async()
    torch.cuda.synchronize()
    if s not in sweeps:
        continue
    v = st.cpu().numpy().reshape(8, 1024, 8).astype(np.float64)[3] / 100.0     # finalize: us
    live = v[:, 0] > 0
    t0 = v[live, 0].min()
    print("sweep %d: K after = %d; %d workgroups; times relative to the first entry, us" % (s, int(seg._dk.K.item()), int(live.sum())))
    for ph in range(7):
        a = v[live, ph]
        a = a[a > 0] - t0
        if len(a):
            print("  phase %d: min %8.2f  mean %8.2f  max %8.2f  (workgroup of the max: %d)" % (ph, a.min(), a.mean(), a.max(), int(np.argmax(np.where(live, v[:, ph], 0)))))
    d = np.where(live[:, None], v, np.nan)
    for ph in range(1, 7):
        dd = d[:, ph] - d[:, ph - 1]
        dd = dd[np.isfinite(dd) & (d[:, ph] > 0) & (d[:, ph - 1] > 0)]
        if len(dd):
            print("  step %d -> %d: mean %8.2f  p90 %8.2f  max %8.2f" % (ph - 1, ph, dd.mean(), np.percentile(dd, 90), dd.max()))
seg._dk.check_status()
