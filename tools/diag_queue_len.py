import sys, random
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
for utts in (1250, 10000):
    corpus = make_corpus(utts, 100, 1000, seed=0, N=20, n_slices_max=6)
    random.seed(0); np.random.seed(0)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
    for _ in range(6):
        seg.batch_sweep_async()
    torch.cuda.synchronize()
    print(utts, "queued rows after 6 sweeps:", int(seg._dk.cand_count.item()), "K", int(seg.acoustic_model.components.K))
