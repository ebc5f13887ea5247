#!/usr/bin/env python3
"""cProfile of the host side of two sequential (reference-chain) k-means sweeps on the headline corpus after two warm ones:
how much of record['sample_time'] is the library call and how much the driver around it (shuffle, order upload, record sums)."""
import cProfile, pstats, random, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread")
seg.segment(2)
pr = cProfile.Profile(); pr.enable(); rec = seg.segment(2); pr.disable()
print(rec["sample_time"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
