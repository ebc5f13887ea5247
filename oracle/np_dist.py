"""
np_dist.py -- the batch-synchronous k-means sweep (np_oracle.kmeans_batch_sweep) split over
ranks exactly as the product splits it: rank r owns a contiguous run of statistics blocks,
exchanges (1) the flagged tokens and (2) the packed block partials with all-gathers, and every
rank replays the clamp / combines the partials in the same fixed order.  TEST INFRASTRUCTURE:
used by tests/test_dist_cpu.py under torch.distributed (gloo, world_size 2) to check that the
protocol reproduces the single-process specification bit for bit.
"""
import numpy as np

from . import np_oracle as no


def kmeans_batch_sweep_rank(seg, n_blocks, rank, world, all_gather_object):
    u, c = seg.utterances, seg.acoustic_model.components
    D = u.D
    nbl = n_blocks // world
    bb = no.block_bounds(D, n_blocks)
    lo, hi = bb[rank * nbl], bb[(rank + 1) * nbl]
    # ---- local: score + DP + tokens against the frozen means
    toks, totals = {}, {}
    for i in range(lo, hi):
        N = u.lengths[i]
        tri = (N * N + N) // 2
        vec = seg.get_vec_embed_neg_len_sqrd_norms(u.vec_ids[i, :tri], u.durations[i, :tri])
        totals[i], bnd = no.forward_backward_kmeans_viterbi(vec, N, seg.n_slices_min, seg.n_slices_max, i)
        u.boundaries[i, :N] = bnd
        new = u.get_segmented_embeds_i(i)
        toks[i] = (new, [int(k) for k in c.get_max_assignments(new)])
    # ---- exchange 1: tokens whose argmax is an inactive row, in token order
    K = c.K
    flags = [(i, t, k) for i in range(lo, hi) for t, k in enumerate(toks[i][1]) if k >= K]
    all_flags = all_gather_object(flags)
    for r, fl in enumerate(all_flags):                      # rank order == utterance order
        for (i, t, k) in fl:
            if k > K:
                k = K
            if k == K:
                K += 1
            if r == rank:
                toks[i][1][t] = k
    c.K = K
    # ---- local block partials, sequential in token order
    parts = []
    for b in range(rank * nbl, (rank + 1) * nbl):
        s = np.zeros((c.K_max, c.D), np.float64)
        n = np.zeros(c.K_max, np.int64)
        tot = np.float64(0.)
        for i in range(bb[b], bb[b + 1]):
            for e, k in zip(*toks[i]):
                s[k] += c.X[e]
                n[k] += 1
            tot += totals[i]
        parts.append((s, n, tot))
    # ---- exchange 2: all block partials; fixed tree on every rank
    all_parts = [p for rp in all_gather_object(parts) for p in rp]
    c.mean_numerators = no.tree_sum([p[0] for p in all_parts])
    c.counts = no.tree_sum([p[1] for p in all_parts])
    total = no.tree_sum([p[2] for p in all_parts])
    for k in range(c.K):
        if c.counts[k] != 0:
            c.means[k] = c.mean_numerators[k] / c.counts[k]
    # ---- clean_components with a relabel table, then the local tokens' final labels
    c.assignments[:] = -1
    for i in range(lo, hi):
        for e, k in zip(*toks[i]):
            c.assignments[e] = k
    c.clean_components()          # relabels c.assignments (local tokens) exactly like the remap table
    return total
