"""Development (diag_chain_dsweep.py D [K_max [f64]]): the persistent FBGMM chain against the four launches per utterance at one D, sweep by sweep: which utterances'
log-probabilities and which spans' scores differ (bits)."""
import os, random, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
from segmentalist_amd.niw import NIW
from segmentalist_amd.synth import make_corpus
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
DT = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else np.float32
corpus = make_corpus(24, D, K, seed=4, ragged=True, n_slices_max=5, N_range=(3, 6), dtype=DT)
prior = NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
out = {}
for mode in ("1", "0"):
    os.environ["SEGK_FB_CHAIN"] = mode
    random.seed(3); np.random.seed(3)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type="diag", fb_type="standard",
                                     n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0,
                                     wip=0.0, init_am_assignments="rand", time_power_term=1.0)
    snaps = []
    for s in range(4):
        rec = seg.gibbs_sample(1, anneal_schedule=None) if False else seg.gibbs_sample(1)
        df = seg._df
        snaps.append(dict(K=int(df.K.item()), lp=df.out_logprob.cpu().numpy().copy(), sc=df.score.cpu().numpy().copy(), sa=df.stat_a.cpu().numpy().copy(),
                          b=seg.utterances.boundaries.copy(), rec=rec["log_marg*length"][0]))
    out[mode] = snaps
for s in range(4):
    a, b = out["1"][s], out["0"][s]
    dl = np.nonzero(a["lp"] != b["lp"])[0]
    ds = np.nonzero((a["sc"] != b["sc"]) & ~(np.isnan(a["sc"]) & np.isnan(b["sc"])))[0]
    print("sweep %d (K %d): record equal %s; utterances with a different log-probability: %s; spans with a different score: %d %s; stats equal %s, boundaries equal %s"
          % (s, a["K"], a["rec"] == b["rec"], dl.tolist(), len(ds), [(int(i), float(a["sc"][i]), float(b["sc"][i] - a["sc"][i])) for i in ds[:6]],
             np.array_equal(a["sa"], b["sa"]), np.array_equal(a["b"], b["b"])), flush=True)
