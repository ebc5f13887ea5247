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
    from segmentalist_amd.device import all_gather_rows

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
    # the product's exchange helper on CPU tensors (gloo branch)
    rows = torch.zeros((world, 5), dtype=torch.float64)
    rows[rank] = torch.arange(5, dtype=torch.float64) + 10 * rank
    all_gather_rows(rows, rows[rank])
    assert all(torch.equal(rows[r], torch.arange(5, dtype=torch.float64) + 10 * r) for r in range(world))
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
