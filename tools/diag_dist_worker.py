"""Development: tests/dist_worker.py's run with a check after every sweep, on every rank, that the float32 tile image of the
means (segk_kmeans.tiles: what the full scan k_kmeans_brute_ls reads) holds the means of that moment."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_sweeps = int(sys.argv[1])
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(96, 24, 40, seed=3, ragged=True, n_slices_max=5, N_range=(4, 14))
    random.seed(11)
    np.random.seed(11)
    seg = kaw.SegmentalKMeansWordseg(40, *corpus, n_slices_max=5, init_am_assignments="rand", sync="batch", n_stat_blocks=8)
    dk = seg._dk
    K, D = 40, 24
    n_tiles = (K + 31) // 32
    stride = dk.tiles.numel() // n_tiles if dk.tiles.numel() % n_tiles == 0 else None
    tot = []
    X = seg._corpus.X.cpu().numpy().astype(np.float32)[:, :D]
    prev_k = dk.cand_k.cpu().numpy().copy()
    for sw in range(n_sweeps):
        m_before = dk.means.cpu().numpy().reshape(K, D).astype(np.float32).copy()
        rec = seg.segment(1)
        torch.cuda.synchronize()
        # every local row's argmax / max of -|x - m|^2 (float32, numpy's summation) against what the sweep's score call left
        ck, cs = dk.cand_k.cpu().numpy(), dk.cand_s.cpu().numpy()
        nbad = nstale = 0
        wrong = []
        q = dk.cand_queue.cpu().numpy()
        for r in range(X.shape[0]):
            dl = m_before - X[r]
            sc = -(dl * dl).sum(axis=1)
            k = int(np.argmax(sc))
            if ck[r] != k or np.float32(cs[r]) != sc[k]:
                nbad += 1
                nstale += int(ck[r] == prev_k[r])
                wrong.append((r, int(ck[r]), float(cs[r]), k, float(sc[k]), float(sc[ck[r]]) if 0 <= ck[r] < K else None))
        if nbad:
            pos = {int(v): i for i, v in enumerate(q[:X.shape[0]])}
            print("WRONG rank %d sweep %d: K=%d count=%d; (row, queue position, got k, got s, want k, want s, ref score of got k): %s"
                  % (rank, sw, int(dk.K.item()) if hasattr(dk, "K") else -1, int(dk.cand_count.item()),
                     [(w[0], pos.get(w[0], -1)) + w[1:] for w in wrong[:12]]), flush=True)
            print("WRONGPOS rank %d sweep %d: queue positions of the wrong rows: %s" % (rank, sw, sorted(pos.get(w[0], -1) for w in wrong)), flush=True)
        if nbad:
            print("SCORES rank %d sweep %d: %d of %d rows differ from the reference argmax (%d of them hold the previous sweep's label)"
                  % (rank, sw, nbad, X.shape[0], nstale), flush=True)
        prev_k = ck.copy()
        tot.append(rec["sum_neg_len_sqrd_norm"][-1])
        t = dk.tiles.cpu().numpy()
        m = dk.means.cpu().numpy().reshape(K, D)
        G = (D + 3) // 4
        st = (G * 128 + 32 + 1023) // 1024 * 1024
        bad = 0
        for k in range(K):
            for d in range(D):
                v = t[(k >> 5) * st + (d >> 2) * 128 + ((((d >> 1) & 1) * 32 + (k & 31)) << 1) + (d & 1)]
                bad += int(v != m[k, d])
        if bad:
            print("TILES rank %d sweep %d: %d of %d elements of the fp32 tile image differ from the means" % (rank, sw, bad, K * D), flush=True)
    print("TOTALS rank %d %s" % (rank, " ".join("%.8f" % x for x in tot)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
