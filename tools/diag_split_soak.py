"""Development: batch sweeps at a shard size with the second stage split over component ranges (the default there) against the
unsplit kernel (SEGK_SP2_SPLIT=1) and against the split-precision filter alone (SEGK_SCORE_PRE=0): identical state."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n_sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 30


def run(env):
    for k in ("SEGK_SP2_SPLIT", "SEGK_SCORE_PRE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    corpus = make_corpus(U, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    for _ in range(n_sweeps):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    seg._dk.check_status()
    c = seg.acoustic_model.components
    return seg.utterances.boundaries.copy(), c.assignments.copy(), c.means.copy(), c.mean_numerators.copy(), c.counts.copy(), c.K


ref = run({"SEGK_SCORE_PRE": "0"})
for name, env in (("default (pre-filter, split second stage)", {}), ("unsplit second stage", {"SEGK_SP2_SPLIT": "1"})):
    got = run(env)
    same = all(np.array_equal(a, b) for a, b in zip(ref[:5], got[:5])) and ref[5] == got[5]
    print("%d utterances, %d sweeps, %s: %s (K = %d)" % (U, n_sweeps, name, "identical to the split-precision filter alone" if same else "DIFFERENT", got[5]))
