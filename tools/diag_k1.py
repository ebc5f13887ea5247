#!/usr/bin/env python3
"""Development: duration of k_kmeans_top2_rs (library-recorded events) on the headline corpus under the environment given."""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
if os.environ.get("SEGK_LIB_PATH"):          # a -DSEGK_STAMP build kept beside the product build (build_stamp/libsegk_stamp.so)
    _abi.LIB_PATH = os.environ["SEGK_LIB_PATH"]
from segmentalist_amd.synth import make_corpus
n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
corpus = make_corpus(n_utt, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
for _ in range(6): seg.batch_sweep_async()
torch.cuda.synchronize()
dk = seg._dk
L, ctx = _abi.lib(), _abi.ctx()
for env in sys.argv[2:] or [""]:
    for kv in env.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            os.environ[k] = v
    if "ZERO=1" in env:          # clock check: the same instruction stream on all-zero operands (results meaningless)
        dk.corpus.Xb3[64:].zero_()
        dk.tiles_b3.view(torch.uint8)[4096:].zero_()
    _abi.check(L.segk_profile_enable(ctx, 1))
    for _ in range(14):
        dk.score_rows(row0=0, n=dk.corpus.n_emb, hint_remap=dk.remap)
        torch.cuda.synchronize()
    ms = (C.c_float * 64)(); rows = (C.c_int64 * 64)()
    got = L.segk_profile_read(ctx, ms, rows, 64)
    _abi.check(L.segk_profile_enable(ctx, 0))
    v = np.array(ms[2:got])
    print("%-40s K1 median %.1f us  min %.1f  (%d launches, kind %d)" % (env, 1e3 * np.median(v), 1e3 * v.min(), len(v), L.segk_profile_last_kind(ctx)), flush=True)
    for kv in env.split(","):
        if "=" in kv:
            os.environ.pop(kv.split("=")[0], None)
