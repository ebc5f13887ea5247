"""Multi-rank batch mode of the PRODUCT: 1, 2 and 4 ranks (gloo, all ranks on the one GPU of
the test box) must produce bit-identical state -- the fixed summation tree and the replayed
`k > K -> K` clamp make the result independent of the sharding."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(world, out, n_sweeps=3, worker="dist_worker.py", extra=(), env_extra=None, backend="gloo"):
    worker = os.path.join(ROOT, "tests", worker)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.update(env_extra or {})
    if world == 1:
        cmd = [sys.executable, worker, out, backend, str(n_sweeps)] + list(extra)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + world), worker, out, backend,
               str(n_sweeps)] + list(extra)
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    return np.load(out)


def test_batch_mode_is_independent_of_the_number_of_ranks(tmp_path):
    ref = run(1, str(tmp_path / "w1.npz"))
    for world in (2, 4):
        got = run(world, str(tmp_path / ("w%d.npz" % world)))
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (world, k)


def test_batch_mode_with_the_corpus_replicated_on_every_rank(tmp_path):
    """shard_corpus=False (every rank keeps all rows on its device, as rounds 1-3 did; needed to mix sequential-mode calls
    into a multi-rank run): the same bits as one rank and as the sharded default."""
    ref = run(1, str(tmp_path / "r1.npz"))
    got = run(2, str(tmp_path / "r2.npz"), extra=["--no-shard"])
    for k in ref.files:
        assert np.array_equal(ref[k], got[k]), k


def test_batch_mode_replayed_as_hipgraphs_is_independent_of_the_number_of_ranks(tmp_path):
    """SEGK_SWEEP_GRAPH=1: the sweep captured as one hipGraph (two around the all-gather with several ranks) and
    replayed from the second sweep on -- same bits as the plain launches."""
    ref = run(1, str(tmp_path / "g0.npz"), 4, env_extra={"SEGK_SWEEP_GRAPH": "0"})
    for world in (1, 2):
        got = run(world, str(tmp_path / ("g%d.npz" % world)), 4, env_extra={"SEGK_SWEEP_GRAPH": "1"})
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (world, k)


def test_batch_mode_over_rccl_when_the_box_has_two_gpus(tmp_path):
    """Backend nccl (= RCCL over xGMI): one rank per GPU, the per-sweep all-gather in place on the device buffer
    (comm.TorchComm.all_gather_rows), ensure_assignments / ensure_boundaries as device collectives.  Needs two GPUs: the
    single-GPU test box skips it, a multi-GPU box exercises the RCCL branch the first time it sees this suite."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("backend nccl needs one GPU per rank; this box has %d" % torch.cuda.device_count())
    ref = run(1, str(tmp_path / "n1.npz"))
    got = run(2, str(tmp_path / "n2.npz"), backend="nccl", env_extra={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ref.files:
        assert np.array_equal(ref[k], got[k]), k


def test_checkpoint_written_under_one_world_size_resumes_under_another(tmp_path):
    """state_dict() under 2 ranks holds the complete state (boundaries and assignments of every rank's utterances,
    ADVICE r01): resumed under 1 and under 4 ranks it continues exactly like the uninterrupted single-rank chain."""
    ref = run(1, str(tmp_path / "c_ref.npz"), 3)
    ck = str(tmp_path / "ck.pkl")
    run(2, str(tmp_path / "c_a.npz"), 2, extra=["--save", ck])
    for world in (1, 4):
        got = run(world, str(tmp_path / ("c_b%d.npz" % world)), 1, extra=["--load", ck])
        for k in ("assignments", "means", "mean_numerators", "counts", "K", "boundaries"):
            assert np.array_equal(ref[k], got[k]), (world, k)
        assert ref["totals"][-1] == got["totals"][-1]


def test_batch_mode_with_the_prefilter_forced_is_independent_of_the_number_of_ranks(tmp_path):
    """The same with SEGK_SCORE_PRE=1: every rank scores its row range (first row > 0 on ranks > 0) through
    the one-product pre-filter, the exact pair kernel, the second stage and the full scan; the result
    must still be the single-rank, un-prefiltered one bit for bit."""
    ref = run(1, str(tmp_path / "p0.npz"), env_extra={"SEGK_SCORE_PRE": "0"})
    for world in (1, 2, 4):
        got = run(world, str(tmp_path / ("p%d.npz" % world)), env_extra={"SEGK_SCORE_PRE": "1"})
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (world, k)


@pytest.mark.parametrize("kind", ["diag", "bigram"])
def test_fbgmm_batch_mode_is_independent_of_the_number_of_ranks(tmp_path, kind):
    """The blocked-Gibbs sampler of the FBGMM / bigram drivers on 1, 2 and 4 ranks: boundaries,
    assignments, statistics, LM tables and record values bit-identical."""
    ref = run(1, str(tmp_path / "f1.npz"), 2, "dist_worker_fbgmm.py", [kind])
    for world in (2, 4):
        got = run(world, str(tmp_path / ("f%d.npz" % world)), 2, "dist_worker_fbgmm.py", [kind])
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (world, k)


def test_bigram_batch_matrix_core_mode_is_independent_of_the_number_of_ranks(tmp_path):
    """score_precision="f16" (matrix-core span scores and token likelihoods): every row's score is
    independent of how the rows are grouped into launches, so 1 / 2 / 4 ranks still coincide bit for bit."""
    ref = run(1, str(tmp_path / "h1.npz"), 2, "dist_worker_fbgmm.py", ["bigram", "f16"])
    for world in (2, 4):
        got = run(world, str(tmp_path / ("h%d.npz" % world)), 2, "dist_worker_fbgmm.py", ["bigram", "f16"])
        for k in ref.files:
            assert np.array_equal(ref[k], got[k]), (world, k)


def test_bench_self_launches_its_ranks_and_reports_the_measured_collective(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher (the form the driver uses for N = 1): bench.py starts the two ranks as a
    child torchrun job before touching the GPU, the ranks share the card over gloo (one GPU here), rank 0's JSON line comes
    back through the parent and carries the measured per-sweep all-gather."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "1",
           "--min-seconds", "0", "--cpu-utts", "0", "--utts", "400", "--K", "64", "--dim", "16"]
    res = subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    cm = out["config"]["collective_measured"]
    assert cm["world_size"] == 2 and cm["all_gather_us_per_sweep_max_over_ranks"] > 0
    import torch
    assert cm["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
