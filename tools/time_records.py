import os, random, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from segmentalist_amd import bigram_acoustic_wordseg as baw
from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
from segmentalist_amd.synth import make_corpus
D, K = 100, 1000
corpus = make_corpus(10000, D, K, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                covariance_type="fixed", fb_type="unigram", n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                beta_sent_boundary=-1, sync="batch")
t0 = time.time(); rec = seg.gibbs_sample(2); print("gibbs_sample(2) with records: %.2f s" % (time.time() - t0), rec["sample_time"], rec["log_marg"], rec["components"])
