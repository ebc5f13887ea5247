import random, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from segmentalist_amd import kmeans_acoustic_wordseg as kaw
from segmentalist_amd.synth import make_corpus
corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
random.seed(0); np.random.seed(0)
seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_max=6, init_am_assignments="spread", sync="batch")
pt = seg._get_sweeper().part
for it in range(12):
    seg.batch_sweep_async()
    torch.cuda.synchronize()
    nf = seg._dk.n_flag.cpu().numpy()
    per_block = [int(nf[pt.bounds[b]:pt.bounds[b + 1]].sum()) for b in range(8)]
    print(it + 1, int(seg._dk.K.item()), per_block, flush=True)
seg._dk.check_status()
