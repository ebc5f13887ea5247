"""CPU tests of the host-side mirror (A13): Utterances / process_embeddings / RNG indirection /
partitioning, against the oracle's restatement (itself pinned to the reference)."""
import random

import numpy as np

from oracle import np_oracle as no
from segmentalist_amd import rng
from segmentalist_amd.device import Partition
from segmentalist_amd.utterances import Utterances, process_embeddings
from tests.golden import cases


def test_process_embeddings_matches_oracle():
    emb, vid, dur, lm = cases.chain_corpus(9, 4, 3, 5, True, 0, 4, "float32")
    E1, V1, L1 = process_embeddings(emb, vid)
    E2, V2, L2 = no.process_embeddings(emb, vid)
    assert L1 == L2 and np.array_equal(E1, E2) and E1.dtype == E2.dtype
    for a, b in zip(V1, V2):
        assert np.array_equal(a, b)
    assert V1.row_start[-1] == E1.shape[0]


def test_utterances_init_matches_oracle_and_consumes_same_rng(golden):
    g = golden("chains")
    for chain in cases.KMEANS_CHAINS:
        name, n_utt, D, K, seed, ragged, N, nmax, dtype = chain
        emb, vid, dur, lm = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
        E, V, labels = process_embeddings(emb, vid)
        args = ([len(lm[i]) for i in labels], V, [dur[i] for i in labels], [lm[i] for i in labels])
        np.random.seed(1)
        u = Utterances(*args, p_boundary_init=0.5, n_slices_min=0, n_slices_max=nmax)
        after = np.random.rand()
        np.random.seed(1)
        o = no.Utterances(*args, p_boundary_init=0.5, n_slices_min=0, n_slices_max=nmax)
        assert after == np.random.rand()
        assert np.array_equal(u.boundaries, o.boundaries)
        assert np.array_equal(u.boundaries, g["%s_spread_init_bounds" % name])
        assert np.array_equal(u.vec_ids, o.vec_ids)
        assert np.array_equal(u.durations, o.durations, equal_nan=True)
        for i in range(u.D):
            assert list(u.get_segmented_embeds_i(i)) == list(o.get_segmented_embeds_i(i))
            assert u.get_segmented_landmark_indices(i) == o.get_segmented_landmark_indices(i)


def test_min_duration_and_seed_boundaries():
    emb, vid, dur, lm = cases.chain_corpus(5, 3, 2, 8, True, 0, 4, "float32")
    E, V, labels = process_embeddings(emb, vid)
    args = ([len(lm[i]) for i in labels], V, [dur[i] for i in labels], [lm[i] for i in labels])
    seeds = [[l[0] + 1, l[-1]] for l in args[3]]
    u = Utterances(*args, seed_boundaries=seeds, min_duration=9)
    o = no.Utterances(*args, seed_boundaries=seeds, min_duration=9)
    assert np.array_equal(u.boundaries, o.boundaries)
    assert np.array_equal(u.durations, o.durations, equal_nan=True)
    u0 = Utterances(*args, p_boundary_init=0)
    assert u0.boundaries.sum() == u0.D


def test_py2_shuffle_is_selectable():
    random.seed(7)
    a = list(range(10))
    rng.set_shuffle("py2")
    try:
        rng.shuffle(a)
    finally:
        rng.set_shuffle("py3")
    random.seed(7)
    b = list(range(10))
    no.shuffle_py2(b)
    assert a == b
    random.seed(7)
    c = list(range(10))
    rng.shuffle(c)
    random.seed(7)
    d = list(range(10))
    random.shuffle(d)
    assert c == d


def test_partition_covers_everything_once():
    row_start = np.arange(0, 1001 * 7, 7)
    for world in (1, 2, 4, 8):
        lo = 0
        for r in range(world):
            p = Partition(1000, row_start, n_blocks=8, rank=r, world=world)
            assert p.utt_lo == lo
            lo = p.utt_hi
            assert p.row_lo == 7 * p.utt_lo and p.row_hi == 7 * p.utt_hi
            assert len(p.local_bounds) == 8 // world + 1
        assert lo == 1000
    assert list(Partition(1000, row_start, 8, 0, 1).bounds) == no.block_bounds(1000, 8)


def test_virtual_ranks_communicator_behaves_like_a_process_group():
    """tests/virtual_ranks.py (eight ranks in one process for the one-GPU box): the three collectives of
    segmentalist_amd/comm.py on CPU tensors, and a rank that skips a collective fails the run instead of passing."""
    import pytest
    import torch
    from tests.virtual_ranks import VirtualWorld

    def fn(comm):
        rows = torch.zeros((comm.world, 3), dtype=torch.float64)
        rows[comm.rank] = torch.arange(3, dtype=torch.float64) + 10 * comm.rank
        comm.all_gather_rows(rows, rows[comm.rank])
        mx = torch.full((comm.world,), -1, dtype=torch.int32)
        mx[comm.rank] = comm.rank
        comm.all_reduce_max(mx)
        objs = comm.all_gather_object({"rank": comm.rank})
        return rows, mx, objs

    for rows, mx, objs in VirtualWorld(8).run(fn):
        assert all(torch.equal(rows[r], torch.arange(3, dtype=torch.float64) + 10 * r) for r in range(8))
        assert mx.tolist() == list(range(8))
        assert [o["rank"] for o in objs] == list(range(8))

    def skips(comm):
        if comm.rank != 3:
            comm.all_gather_object(comm.rank)
        return comm.rank

    with pytest.raises(RuntimeError, match="never entered"):
        VirtualWorld(4, timeout=2).run(skips)


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE must not die on an assertion: it starts `python -m torch.distributed.run
    --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (never an exec) with a 127.0.0.1 rendezvous and relays
    its output; with fewer GPUs than ranks (here: none) the child is told to use gloo."""
    import io
    import os
    import subprocess
    import sys
    import pytest
    import bench
    seen = {}

    class FakeProc(object):
        stdout = io.StringIO('{"metric": "x"}\n')

        def wait(self):
            return 0

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return FakeProc()

    class FakeRun(object):
        stdout = "0\n"

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: FakeRun())
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("SEGK_BENCH_BACKEND", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["SEGK_BENCH_BACKEND"] == "gloo" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_host_init_stats_of_a_sharded_corpus_equal_the_reference_constructor():
    """device._host_init_stats (what the ranks of a sharded corpus upload instead of running k_kmeans_init_stats, which needs
    every row on the device) against the oracle's KMeansComponents.__init__ (kmeans_components.py:79-81: add_item for k
    ascending, i ascending): means, numerators, counts and K bit for bit, float32 and float64, with unassigned rows, an
    empty slot in the middle of the labels' range excluded by construction (labels are consecutive) and empty trailing slots."""
    from segmentalist_amd.device import _host_init_stats
    for dtype in (np.float32, np.float64):
        rs = np.random.RandomState(3)
        n, D, K_max = 700, 13, 40
        X = rs.randn(n, D).astype(dtype)
        a = rs.randint(0, 31, n)
        a[rs.rand(n) < 0.2] = -1
        a[:31] = np.arange(31)                       # every label 0..30 occurs: consecutive, slots 31..39 stay empty
        np.random.seed(11)
        ref = no.KMeansComponents(X, a.copy(), K_max)
        means, numer, counts, K = _host_init_stats(X, a, K_max, ref.random_means)
        assert K == ref.K == 31
        assert means.dtype == ref.means.dtype and np.array_equal(means, ref.means)
        assert np.array_equal(numer, ref.mean_numerators)
        assert np.array_equal(counts, ref.counts)


def test_complete_band_tables_only_when_no_embedding_lies_outside_the_window():
    """Utterances.complete_band_tables: the banded image of the span tables for the FBGMM kernels, which read nothing else --
    handed over only when every embedding of the triangle is inside the band."""
    from segmentalist_amd.synth import make_corpus
    from segmentalist_amd.utterances import Utterances, process_embeddings
    mats, vec_ids_dict, durations_dict, landmarks_dict = make_corpus(12, 4, 6, seed=1, ragged=True, n_slices_max=4, N_range=(3, 9))
    _, vec_ids, labels = process_embeddings(mats, vec_ids_dict)
    u = Utterances([len(landmarks_dict[i]) for i in labels], vec_ids, [durations_dict[i] for i in labels],
                   [landmarks_dict[i] for i in labels], p_boundary_init=0.5, n_slices_min=0, n_slices_max=4)
    band = u.complete_band_tables(4)
    assert band is not None
    ids, dur = band
    assert ids.shape == (u.D, u.N_max, 4) and dur.shape == ids.shape
    tri = np.asarray(u.vec_ids)
    assert np.count_nonzero(ids >= 0) == np.count_nonzero(tri >= 0)
    for i in range(u.D):
        for t in range(1, u.N_max + 1):
            for w in range(4):
                s = t - 1 - w
                want = tri[i, t * (t - 1) // 2 + s] if s >= 0 else -1
                assert ids[i, t - 1, w] == want
    assert u.complete_band_tables(3) is None            # spans of four slices have embeddings: a window of three drops them
    assert u.complete_band_tables(0) is None and u.complete_band_tables(u.N_max) is None


def test_parallel_forms_of_the_finalize_kernels_serial_loops_equal_the_serial_loops():
    """k_batch_finalize (csrc/segk_stats.hip) runs two loops of the reference in parallel; the formulas it relies on, restated
    here in a few lines each, against the serial loops on random inputs (the kernel itself is held to the oracle by the GPU
    parity tests; this pins the derivations, including the corner cases: adjacent holes, holes at the top, no holes, every row
    a hole, labels far beyond K).
    (1) clean_components on indices (kmeans_components.py:129-151, 263-266): holes walked in descending order, each filled
        from the last active position.  Parallel: hole number i (1 = highest) is filled from position P = K1 - i; while P is
        itself a hole, number j, it holds what hole j received: continue at K1 - j.
    (2) add_item's `k > K -> K` clamp (kmeans_components.py:102-106) over the flagged tokens in order: token q founds a
        component iff its raw label >= K + (founders before it); per 64 tokens the founders' mask is the fixed point of
        mask -> [raw >= K + popcount(mask below)], reached from any start in at most 64 rounds."""
    rs = np.random.RandomState(5)
    for trial in range(300):
        K1 = int(rs.randint(1, 200))
        p_hole = rs.choice([0.0, 0.05, 0.3, 0.7, 1.0])
        hole = rs.rand(K1) < p_hole
        # serial walk
        pos = list(range(K1))
        K = K1
        order = []
        for k in range(K1 - 1, -1, -1):
            if hole[k]:
                K -= 1
                if k != K:
                    pos[k] = pos[K]
                order.append(k)
        # parallel form
        above = np.concatenate([np.cumsum(hole[::-1])[::-1][1:], [0]]).astype(int)      # holes at positions > k
        pos2 = list(range(K1))
        holes2 = [None] * int(hole.sum())
        for k in range(K1):
            if not hole[k]:
                continue
            i = above[k] + 1
            holes2[i - 1] = k
            p = K1 - i
            if p == k:
                continue
            while hole[p]:
                p = K1 - (above[p] + 1)
            pos2[k] = p
        assert holes2 == order
        assert K == K1 - int(hole.sum())
        assert pos2[:K] == pos[:K], (trial, K1)
        for k in order:                     # (what the relabel table is built from: holes below the final K)
            if k < K:
                assert pos2[k] == pos[k]
    for trial in range(300):
        Kb = int(rs.randint(0, 50))
        n = int(rs.randint(0, 300))
        raw = Kb + rs.randint(0, rs.choice([1, 3, 40, 1000]), size=n)
        K = Kb
        want = []
        for r in raw:
            k = min(int(r), K)
            if k == K:
                K += 1
            want.append(k)
        K2 = Kb
        got = []
        for q0 in range(0, n, 64):
            kr = raw[q0:q0 + 64]
            m = len(kr)
            mask = kr >= K2
            for _ in range(66):
                below = np.concatenate([[0], np.cumsum(mask)[:-1]]) if m else np.zeros(0, int)
                m2 = kr >= K2 + below
                if np.array_equal(m2, mask):
                    break
                mask = m2
            else:
                raise AssertionError("no fixed point within 66 rounds")
            below = np.concatenate([[0], np.cumsum(mask)[:-1]]) if m else np.zeros(0, int)
            got.extend(np.minimum(kr, K2 + below).tolist())
            K2 += int(mask.sum())
        assert got == want and K2 == K, trial
