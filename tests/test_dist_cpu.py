"""
N > 1 path on CPU (gloo, world_size 2): the rank-split batch sweep protocol (partition, flagged
token replay, packed partials, fixed combine tree) reproduces the single-process specification
bit for bit, and the product's exchange helper moves the rows it should.
"""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import np_dist, np_oracle as no
    from segmentalist_amd.comm import TorchComm, get_comm

    def ago(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    corpus = cases.chain_corpus(30, 6, 12, 4242, True, 0, 5, "float32")
    random.seed(9)
    np.random.seed(9)
    seg = no.SegmentalKMeansWordseg(12, *corpus, n_slices_max=5, init_am_assignments="rand")
    c = seg.acoustic_model.components
    totals = []
    for it in range(3):
        totals.append(np_dist.kmeans_batch_sweep_rank(seg, 8, rank, world, ago))
    # merge the rank-local pieces for comparison
    a = torch.from_numpy(c.assignments.copy())
    dist.all_reduce(a, op=dist.ReduceOp.MAX)
    b = torch.from_numpy(seg.utterances.boundaries.astype(np.uint8))
    gathered = [torch.empty_like(b) for _ in range(world)]
    dist.all_gather(gathered, b)
    bb = no.block_bounds(seg.utterances.D, 8)
    nbl = 8 // world
    full = b.clone()
    for r in range(world):
        full[bb[r * nbl]:bb[(r + 1) * nbl]] = gathered[r][bb[r * nbl]:bb[(r + 1) * nbl]]
    # the product's communicator on CPU tensors (gloo branch of segmentalist_amd/comm.py)
    rows = torch.zeros((world, 5), dtype=torch.float64)
    rows[rank] = torch.arange(5, dtype=torch.float64) + 10 * rank
    comm = get_comm()
    assert isinstance(comm, TorchComm) and (comm.rank, comm.world, comm.backend) == (rank, world, "gloo")
    comm.all_gather_rows(rows, rows[rank])
    assert all(torch.equal(rows[r], torch.arange(5, dtype=torch.float64) + 10 * r) for r in range(world))
    mx = torch.full((4,), -1, dtype=torch.int32)
    mx[rank] = 7 + rank
    comm.all_reduce_max(mx)
    assert mx.tolist() == [7 + r if r < world else -1 for r in range(4)]
    assert comm.all_gather_object(("r", rank)) == [("r", r) for r in range(world)]
    if rank == 0:
        np.savez(os.path.join(out_dir, "dist.npz"), assignments=a.numpy(), boundaries=full.numpy(),
                 means=c.means, mean_numerators=c.mean_numerators, counts=c.counts, K=np.array(c.K),
                 totals=np.array(totals))
    dist.destroy_process_group()


def test_rank_split_protocol_equals_single_process_spec(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "dist.npz"))
    from oracle import np_oracle as no
    corpus = cases.chain_corpus(30, 6, 12, 4242, True, 0, 5, "float32")
    random.seed(9)
    np.random.seed(9)
    seg = no.SegmentalKMeansWordseg(12, *corpus, n_slices_max=5, init_am_assignments="rand")
    c = seg.acoustic_model.components
    totals = [no.kmeans_batch_sweep(seg, n_blocks=8) for _ in range(3)]
    assert np.array_equal(got["boundaries"].astype(bool), seg.utterances.boundaries)
    assert np.array_equal(got["assignments"], c.assignments)
    assert np.array_equal(got["means"], c.means)
    assert np.array_equal(got["mean_numerators"], c.mean_numerators)
    assert np.array_equal(got["counts"], c.counts)
    assert int(got["K"]) == c.K
    assert np.array_equal(got["totals"], np.array(totals))


# ------------------------------------------------------------------ FBGMM / bigram batch sampler
def _fb_make(kind):
    from oracle import np_oracle as no
    corpus = cases.chain_corpus(26, 6, 9, 777, True, 0, 5, "float32")
    random.seed(4)
    np.random.seed(4)
    kw = dict(n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
              init_am_assignments="rand", time_power_term=1.0)
    if kind == "bigram":
        return no.BigramAcousticWordseg(9, no.FixedVarPrior(*cases.fixed_prior_params(6)), dict(cases.BIGRAM_LM),
                                        *corpus, covariance_type="fixed", fb_type="unigram", **kw)
    return no.UnigramAcousticWordseg(no.FBGMM, 1.0, 9, no.NIW(*cases.diag_prior_params(6)), *corpus,
                                     covariance_type="diag", fb_type="standard", **kw)


def _fb_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import np_fbgmm_batch as nb

    def ago(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    res = {}
    for kind in ("diag", "bigram"):
        seg = _fb_make(kind)
        st = nb.FbgmmBatch(seg, n_gibbs_blocks=3, n_stat_blocks=4, seed=21, rank=rank, world=world, all_gather_object=ago)
        lps = [st.sweep(sw) for sw in range(3)]
        # merge the rank-local pieces: every rank owns the utterances (and rows) of its slices
        own = [i for s in range(st.s_lo, st.s_hi) for b in range(st.B) for i in range(*st.ranges[s][b])]
        pieces = ago({i: (seg.utterances.boundaries[i].copy(), [(e, st.slot[e]) for e in st._tokens(i)],
                          [lp[i] for lp in lps]) for i in own})
        if rank == 0:
            bounds = np.zeros_like(seg.utterances.boundaries)
            slot = -np.ones_like(st.slot)
            lp = np.zeros((3, seg.utterances.D))
            for part in pieces:
                for i, (bd, toks, l) in part.items():
                    bounds[i] = bd
                    for e, k in toks:
                        slot[e] = k
                    lp[:, i] = l
            res[kind + "_bounds"], res[kind + "_slot"], res[kind + "_lp"] = bounds, slot, lp
            res[kind + "_cnt"] = st.stats_excluding(-1)[0]
            if kind == "bigram":
                res["bigram_big"] = st.big
    if rank == 0:
        np.savez(os.path.join(out_dir, "fb.npz"), **res)
    dist.destroy_process_group()


def test_fbgmm_batch_rank_split_equals_single_process_spec(tmp_path):
    """World-2 gloo run of the blocked-Gibbs protocol (per-step exchange of the block's partial sums
    and, with a language model, transcripts) against the single-process specification."""
    from oracle import np_fbgmm_batch as nb
    world = 2
    mp.spawn(_fb_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "fb.npz"))
    for kind in ("diag", "bigram"):
        seg = _fb_make(kind)
        st = nb.FbgmmBatch(seg, n_gibbs_blocks=3, n_stat_blocks=4, seed=21)
        lps = np.stack([st.sweep(sw) for sw in range(3)])
        assert np.array_equal(got[kind + "_bounds"], seg.utterances.boundaries), kind
        assert np.array_equal(got[kind + "_slot"], st.slot), kind
        assert np.array_equal(got[kind + "_lp"], lps), kind
        assert np.array_equal(got[kind + "_cnt"], st.stats_excluding(-1)[0]), kind
        if kind == "bigram":
            assert np.array_equal(got["bigram_big"], st.big)
