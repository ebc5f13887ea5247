#!/usr/bin/env python3
"""Development: wall-clock stamps (s_memrealtime, 100 MHz) of the phases of the sweep's tail kernels -- segment, sort,
partial sums, finalize, post -- from a library built with -DSEGK_STAMP (kept beside the product build:
  make -C segmentalist_amd/csrc OUT=$PWD/build_stamp/libsegk_stamp.so OBJDIR=/tmp/stampbuild/obj HIPFLAGS="... -DSEGK_STAMP").
usage: diag_tail_stamps.py [n_utterances]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import _abi
_abi.LIB_PATH = os.path.join(ROOT, "build_stamp", "libsegk_stamp.so")
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
corpus = make_corpus(n_utt, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
st = torch.zeros(8 * 1024 * 8, dtype=torch.int64, device="cuda")
os.environ["SEGK_TSTAMP_PTR"] = hex(st.data_ptr())
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for _ in range(12): seg.batch_sweep_async()
torch.cuda.synchronize()
st.zero_()
seg.batch_sweep_async()
torch.cuda.synchronize()
v = st.cpu().numpy().reshape(8, 1024, 8).astype(np.float64) / 100.0          # us
names = ["segment", "sort", "partials", "finalize", "post"]
t0 = v[0][:, 0][v[0][:, 0] > 0].min()
def rel(a):
    a = a[a > 0]
    return (a.min() - t0, a.mean() - t0, a.max() - t0) if len(a) else (float("nan"),) * 3
print("all times in us relative to the first workgroup entry of the segment kernel: min / mean / max over the workgroups (first 1024)")
for k, nm in enumerate(names):
    print("== %s" % nm)
    rows = v[k]
    if k == 1:
        nb = 8
        for lab, sl in (("sort workgroups", slice(0, nb)), ("flags + totals workgroups", slice(nb, 2 * nb))):
            print("  %s" % lab)
            for ph in range(6):
                print("    phase %d: %8.2f %8.2f %8.2f" % ((ph,) + rel(rows[sl, ph])))
        continue
    if k == 2:
        if not (rows[:, 0] > 0).any():
            print("  (fused into the sort kernel)")
            continue
        print("  entry: %8.2f %8.2f %8.2f   last wave exit: %8.2f %8.2f %8.2f   longest list of a wave (tokens): max %d mean %.1f" % (
            rel(rows[:, 0]) + rel(rows[:, 1]) + (int(st.cpu().numpy().reshape(8, 1024, 8)[2][:, 2].max()), st.cpu().numpy().reshape(8, 1024, 8)[2][:, 2].mean())))
        live = rows[:, 1] - rows[:, 0]
        live = live[rows[:, 0] > 0]
        print("  workgroup life: mean %.2f  p50 %.2f  p90 %.2f  max %.2f" % (live.mean(), np.percentile(live, 50), np.percentile(live, 90), live.max()))
        continue
    for ph in range(8):
        if (rows[:, ph] > 0).any():
            print("  phase %d: %8.2f %8.2f %8.2f" % ((ph,) + rel(rows[:, ph])))
    if k == 0:
        live = rows[:, 3] - rows[:, 0]
        live = live[rows[:, 0] > 0]
        print("  workgroup life: mean %.2f max %.2f; gathers %.2f, DP %.2f, stores %.2f (wave 0 of each workgroup)" % (
            live.mean(), live.max(), (rows[:, 1] - rows[:, 0])[rows[:, 0] > 0].mean(), (rows[:, 2] - rows[:, 1])[rows[:, 0] > 0].mean(),
            (rows[:, 3] - rows[:, 2])[rows[:, 0] > 0].mean()))
