"""Development: the persistent sequential chain against the three-launch form on the same corpus and seeds."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus

U, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nsw = int(sys.argv[4]) if len(sys.argv) > 4 else 2


def run(chain):
    os.environ["SEGK_SEQ_CHAIN"] = "1" if chain else "0"
    corpus = make_corpus(U, D, K, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=6, init_am_assignments="spread")
    out = []
    for s in range(nsw):
        try:
            rec = seg.segment(1)
        except Exception as e:
            print("chain" if chain else "plain", "sweep", s, "raised", repr(e)[:200])
            rec = None
        c = seg.acoustic_model.components
        out.append(dict(b=seg.utterances.boundaries.copy(), a=c.assignments.copy(), m=c.means.copy(), n=c.mean_numerators.copy(),
                        cnt=c.counts.copy(), K=c.K, rec=rec))
    return out


ref = run(False)
got = run(True)
for s in range(nsw):
    r, g = ref[s], got[s]
    print("sweep", s, "K", r["K"], g["K"], "rec", r["rec"] and r["rec"]["sum_neg_len_sqrd_norm"], g["rec"] and g["rec"]["sum_neg_len_sqrd_norm"])
    db = np.flatnonzero((r["b"] != g["b"]).any(axis=1))
    print("  utterances with different boundaries:", db.size, db[:10])
    print("  assignments differ:", int((r["a"] != g["a"]).sum()), "means rows differ:", int((r["m"] != g["m"]).any(axis=1).sum()),
          "numerators rows differ:", int((r["n"] != g["n"]).any(axis=1).sum()), "counts differ:", int((r["cnt"] != g["cnt"]).sum()))
