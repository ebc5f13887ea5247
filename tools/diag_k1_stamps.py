#!/usr/bin/env python3
"""Development (library built with -DSEGK_STAMP: `make -C segmentalist_amd/csrc clean; make ... HIPFLAGS+=-DSEGK_STAMP`):
where a wave of k_kmeans_top2_rs spends its cycles -- waiting for the prefetched rows, inside the tile loops, in all."""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
if os.environ.get("SEGK_LIB_PATH"):          # a -DSEGK_STAMP build kept beside the product build (build_stamp/libsegk_stamp.so)
    _abi.LIB_PATH = os.environ["SEGK_LIB_PATH"]
from segmentalist_amd.synth import make_corpus
corpus = make_corpus(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for _ in range(6): seg.batch_sweep_async()
torch.cuda.synchronize()
dk = seg._dk
st = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["SEGK_STAMP_PTR"] = hex(st.data_ptr())
if os.environ.get("ZERO") == "1":          # clock check: the same instruction stream on all-zero operands
    dk.corpus.Xb3[64:].zero_()
    dk.tiles_b3.view(torch.uint8)[4096:].zero_()
for _ in range(12):
    dk.score_rows(row0=0, n=dk.corpus.n_emb, hint_remap=dk.remap)
torch.cuda.synchronize()
v = st.cpu().numpy().reshape(-1, 8)
v = v[v[:, 3] > 0]
print("waves %d  groups/wave %.1f" % (len(v), v[:, 3].mean()))
print("cycles per wave: total %.0f  tile loops %.0f (%.1f %%)  row waits %.0f (%.1f %%)" % (
    v[:, 2].mean(), v[:, 1].mean(), 100 * v[:, 1].sum() / v[:, 2].sum(), v[:, 0].mean(), 100 * v[:, 0].sum() / v[:, 2].sum()))
print("tile loop cycles per group: mean %.0f  (16 tiles x 14 MFMAs x 32 = 7168 at full rate) -> %.1f cycles per MFMA" % (
    (v[:, 1] / v[:, 3]).mean(), (v[:, 1] / v[:, 3]).mean() / 224))
print("row wait cycles per group: mean %.0f  max wave %.0f" % ((v[:, 0] / v[:, 3]).mean(), (v[:, 0] / v[:, 3]).max()))
print("fill + start-up cycles per wave: mean %.0f" % v[:, 4].mean())
r0, r1 = v[:, 5].min(), v[:, 6].max()
print("realtime (100 MHz): kernel span %.1f us; wave starts spread %.1f us; wave ends spread %.1f us; mean wave life %.1f us -> clock %.2f GHz" % (
    (r1 - r0) / 100.0, (v[:, 5].max() - r0) / 100.0, (r1 - v[:, 6].min()) / 100.0, ((v[:, 6] - v[:, 5]).mean()) / 100.0,
    (v[:, 2] + v[:, 4]).mean() / ((v[:, 6] - v[:, 5]).mean() / 100.0) / 1e3))
print("start-up: entry -> first rows issued %.0f cycles, LDS fill (loads + writes + barrier) %.0f cycles" % ((v[:, 7] >> 32).mean(), (v[:, 7] & 0xffffffff).mean()))
# wave ends by XCD (workgroup i runs on XCD i % 8) and by position inside the XCD: is the end spread systematic?
wg = np.arange(len(v)) // 4
end = (v[:, 6] - r0) / 100.0
cyc = v[:, 2] + v[:, 4]
print("per XCD: mean / min / max wave end (us) and mean cycles per wave")
for x in range(8):
    sel = (wg % 8) == x
    print("  XCD %d: %.1f / %.1f / %.1f   cycles %.0f   clock %.3f GHz" % (x, end[sel].mean(), end[sel].min(), end[sel].max(), cyc[sel].mean(),
          cyc[sel].mean() / ((v[sel, 6] - v[sel, 5]).mean() / 100.0) / 1e3))
