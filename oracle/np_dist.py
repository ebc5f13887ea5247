"""
np_dist.py -- the batch-synchronous k-means sweep (np_oracle.kmeans_batch_sweep) split over
ranks exactly as the product splits it: rank r owns a contiguous run of statistics blocks and
contributes ONE record to ONE all-gather per sweep -- per block the partial sums / counts of its
tokens whose component is already active, the block's total, and the (normally empty) list of its
tokens whose argmax is an inactive row, in token order.  Every rank then replays the
`k > K -> K` clamp over the flagged tokens of all blocks in global token order, sums the new
components' rows itself (the embedding matrix is replicated; a component founded this sweep
consists of flagged tokens only, an older one of un-flagged tokens only) and combines all blocks
in the same fixed tree.  TEST INFRASTRUCTURE: used by tests/test_dist_cpu.py under
torch.distributed (gloo, world_size 2) to check that the protocol reproduces the single-process
specification bit for bit.
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
    # ---- local block records: partials of the un-flagged tokens (sequential, token order) + flag lists
    K0 = c.K
    record = []
    for b in range(rank * nbl, (rank + 1) * nbl):
        s = np.zeros((c.K_max, c.D), np.float64)
        n = np.zeros(c.K_max, np.int64)
        tot = np.float64(0.)
        flags = []
        for i in range(bb[b], bb[b + 1]):
            for t, (e, k) in enumerate(zip(*toks[i])):
                if k >= K0:
                    flags.append((i, t, k, e))
                else:
                    s[k] += c.X[e]
                    n[k] += 1
            tot += totals[i]
        record.append((s, n, tot, flags))
    # ---- THE exchange; everything below is replicated arithmetic on identical inputs
    blocks = [blk for rec in all_gather_object(record) for blk in rec]       # rank order == block order
    K = K0
    resolved = []                                     # (block, final label, embedding row) in global token order
    for b, (_, _, _, flags) in enumerate(blocks):
        for (i, t, k, e) in flags:
            if k > K:
                k = K
            if k == K:
                K += 1
            resolved.append((b, k, e))
            if lo <= i < hi:
                toks[i][1][t] = k
    c.K = K
    part_sum = [blk[0].copy() for blk in blocks]
    part_cnt = [blk[1].copy() for blk in blocks]
    for (b, k, e) in resolved:                        # sequential per (block, new component), token order
        part_sum[b][k] += c.X[e]
        part_cnt[b][k] += 1
    c.mean_numerators = no.tree_sum(part_sum)
    c.counts = no.tree_sum(part_cnt)
    total = no.tree_sum([blk[2] for blk in blocks])
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
