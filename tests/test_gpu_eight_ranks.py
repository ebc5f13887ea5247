"""
configs[3] -- the headline corpus sharded over EIGHT ranks -- and the FBGMM / bigram batch samplers on eight ranks,
bit-identical to one rank.

A one-GPU test box admits at most six processes on its card, so the eight ranks run as eight threads of this process
behind the communicator interface of segmentalist_amd/comm.py (tests/virtual_ranks.py: strictly one rank runs at a time,
a collective completes when every rank has entered it).  What runs per rank is the product unchanged: the 8-way
Partition, the per-shard launch plan (a 1 250-utterance shard takes the "pre-filter from 98 k rows + split second stage"
plan of segk_kmeans_api.hip), the packed record, the replicated finalize; only the transport of the record is a device
copy instead of RCCL (torch.distributed transports: tests/test_gpu_dist.py with gloo at 1 / 2 / 4 ranks, nccl when the box
has two GPUs).
"""
import random

import numpy as np
import pytest

from tests.virtual_ranks import VirtualWorld

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def _kmeans_run(corpus, K, n_sweeps, comm, counts=None, n_batches=1):
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    random.seed(11)
    np.random.seed(11)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                     init_am_assignments="spread", wip=0, sync="batch", n_stat_blocks=8, process_group=comm,
                                     n_batches=n_batches)
    rec = seg.segment(n_sweeps)
    c = seg.acoustic_model.components
    if counts is not None:           # which stages of the score path this rank's launches went through
        import torch
        from segmentalist_amd import _abi
        out = (_abi.C.c_int32 * 2)()
        _abi.check(_abi.lib().segk_kmeans_stage_counts(_abi.ctx(), _abi.C.byref(seg._dk.cand), out, _abi.stream()))
        counts.append((seg._get_sweeper().part.rank, int(out[0]), int(out[1])))
    # every rank reads the collective attributes in the same order
    state = dict(assignments=c.assignments.copy(), boundaries=seg.utterances.boundaries.copy(), means=c.means.copy(),
                 mean_numerators=c.mean_numerators.copy(), counts=c.counts.copy(), K=c.K,
                 totals=list(rec["sum_neg_len_sqrd_norm"]), n_tokens=list(rec["n_tokens"]),
                 components=list(rec["components"]))
    return state


def _same(a, b):
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


def test_headline_corpus_on_eight_ranks_equals_one_rank(gpu):
    """configs[3]: 10 000 utterances x 105 spans, D = 100, K = 1000; rank r owns 1 250 utterances (131 250 rows)."""
    from segmentalist_amd.comm import SingleComm
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(10000, 100, 1000, seed=0, N=20, n_slices_max=6)
    one = _kmeans_run(corpus, 1000, 3, SingleComm())
    gpu.cuda.empty_cache()
    counts = []
    world = VirtualWorld(8)
    eight = world.run(lambda comm: _kmeans_run(corpus, 1000, 3, comm, counts))
    for r in range(8):
        _same(one, eight[r])           # the complete state on EVERY rank, not only on rank 0
    assert one["components"][-1] < 1000 and len(set(one["totals"])) == 3        # the chain moved
    # every shard went through the pre-filter (its second stage saw rows) -- the per-shard launch plan, not the
    # whole-corpus one and not the split-precision filter alone
    assert sorted(c[0] for c in counts) == list(range(8))
    assert all(c[1] > 0 for c in counts), counts


def test_minibatch_sweeps_on_eight_ranks_equal_one_rank(gpu):
    """n_batches = 4 (four all-gathers per sweep, every rank resegmenting a quarter of its block per step) at a size where the
    hinted score path runs per step on one rank (2 000 utterances: 52 500 rows per step) and the small-launch paths on eight."""
    from segmentalist_amd.comm import SingleComm
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(2000, 100, 1000, seed=1, N=20, n_slices_max=6)
    one = _kmeans_run(corpus, 1000, 3, SingleComm(), n_batches=4)
    eight = VirtualWorld(8).run(lambda comm: _kmeans_run(corpus, 1000, 3, comm, n_batches=4))
    for r in range(8):
        _same(one, eight[r])
    whole = _kmeans_run(corpus, 1000, 3, SingleComm(), n_batches=1)
    assert whole["totals"] != one["totals"]                  # a different chain than the whole-sweep batch mode


@pytest.mark.parametrize("kind,prec", [("diag", "f64"), ("bigram", "f64"), ("bigram", "f16"), ("diag", "f32")])
def test_fbgmm_batch_samplers_on_eight_ranks_equal_one_rank(gpu, kind, prec):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.comm import SingleComm
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K = 24, 40
    corpus = make_corpus(200, D, K, seed=3, ragged=True, n_slices_max=5, N_range=(4, 14))

    def run(comm):
        random.seed(11)
        np.random.seed(11)
        kw = dict(n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                  init_am_assignments="rand", time_power_term=1.0, sync="batch", n_gibbs_blocks=3, n_stat_blocks=8,
                  batch_seed=5, score_precision=prec, process_group=comm)
        fixed = (0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
        if kind == "bigram":
            seg = baw.BigramAcousticWordseg(K, FixedVarPrior(*fixed), {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5},
                                            *corpus, covariance_type="fixed", fb_type="unigram", **kw)
        else:
            seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)),
                                             *corpus, covariance_type="diag", fb_type="standard", **kw)
        rec = seg.gibbs_sample(2)
        c = seg.acoustic_model.components
        sw = seg._get_sweeper()
        state = dict(assignments=c.assignments.copy(), counts=c.counts.copy(), K=c.K, stat_a=c.dev.stat_a.cpu().numpy(),
                     stat_b=c.dev.stat_b.cpu().numpy(), log_marg=list(rec["log_marg"]), lml=list(rec["log_marg*length"]),
                     partials=sw.partials.cpu().numpy(), boundaries=seg.utterances.boundaries.copy())
        if kind == "bigram":
            state["unigram"], state["bigram"] = seg.lm.unigram_counts.copy(), seg.lm.bigram_counts.copy()
        return state

    one = run(SingleComm())
    eight = VirtualWorld(8).run(run)
    for r in range(8):
        _same(one, eight[r])
