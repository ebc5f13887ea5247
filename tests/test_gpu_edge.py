"""Edge cases of the drivers on the GPU against trajectories captured from the reference
(tests/golden/edge.npz): 1- and 2-landmark utterances, spans below min_duration (NaN durations),
n_slices_min = 1, a single initial span (p_boundary_init = 0), seed boundaries snapped to landmarks.
Also the same corpora through the batch modes against their specifications."""
import random

import numpy as np
import numpy.testing as npt
import pytest

from oracle import np_fbgmm_batch as nb
from oracle import np_oracle as no
from tests.golden import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


def _product_mods(**extra):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd import unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW

    def wrap(cls):
        return (lambda *a, **k: cls(*a, **dict(k, **extra))) if extra else cls
    return dict(SegmentalKMeansWordseg=wrap(kaw.SegmentalKMeansWordseg), UnigramAcousticWordseg=wrap(uaw.UnigramAcousticWordseg),
                BigramAcousticWordseg=wrap(baw.BigramAcousticWordseg), FBGMM=fbgmm.FBGMM, FixedVarPrior=FixedVarPrior, NIW=NIW)


def _oracle_mods():
    return dict(SegmentalKMeansWordseg=no.SegmentalKMeansWordseg, UnigramAcousticWordseg=no.UnigramAcousticWordseg,
                BigramAcousticWordseg=no.BigramAcousticWordseg, FBGMM=no.FBGMM, FixedVarPrior=no.FixedVarPrior, NIW=no.NIW)


@pytest.mark.parametrize("case", cases.EDGE_CHAINS, ids=[c[0] for c in cases.EDGE_CHAINS])
def test_edge_case_chains_vs_reference(gpu, golden, case):
    g = golden("edge")
    name, driver = case[0], case[1]
    random.seed(1)
    np.random.seed(1)
    seg = cases.edge_build(_product_mods(), case)
    c = seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, g[name + "_init_bounds"])
    assert np.array_equal(c.assignments, g[name + "_init_assign"])
    for it in range(3):
        rec = seg.segment(1) if driver == "kmeans" else seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, g[name + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[name + "_assign"][it]), it
        key = "sum_neg_len_sqrd_norm" if driver == "kmeans" else "log_marg"
        npt.assert_allclose(rec[key][0], g[name + "_rec_" + key][it], rtol=1e-8)
        assert rec["components"][0] == g[name + "_rec_components"][it]
        assert rec["n_tokens"][0] == g[name + "_rec_n_tokens"][it]


@pytest.mark.parametrize("case", [c for c in cases.EDGE_CHAINS if c[1] != "kmeans"],
                         ids=[c[0] for c in cases.EDGE_CHAINS if c[1] != "kmeans"])
def test_edge_cases_batch_sampler_vs_specification(gpu, case):
    random.seed(2)
    np.random.seed(2)
    ref = cases.edge_build(_oracle_mods(), case)
    spec = nb.FbgmmBatch(ref, n_gibbs_blocks=2, n_stat_blocks=2, seed=3)
    random.seed(2)
    np.random.seed(2)
    seg = cases.edge_build(_product_mods(sync="batch", n_gibbs_blocks=2, n_stat_blocks=2, batch_seed=3), case)
    for sw in range(3):
        lp = spec.sweep(sw)
        seg.batch_sweep_async()
        gpu.cuda.synchronize()
        seg._df.check_status()
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        assert np.array_equal(seg._get_sweeper().slot.cpu().numpy(), spec.slot), sw
        npt.assert_allclose(seg._df.out_logprob.cpu().numpy(), lp, rtol=1e-9)


@pytest.mark.parametrize("case", [c for c in cases.EDGE_CHAINS if c[1] == "kmeans"],
                         ids=[c[0] for c in cases.EDGE_CHAINS if c[1] == "kmeans"])
def test_edge_cases_kmeans_batch_vs_specification(gpu, case):
    random.seed(2)
    np.random.seed(2)
    ref = cases.edge_build(_oracle_mods(), case)
    random.seed(2)
    np.random.seed(2)
    seg = cases.edge_build(_product_mods(sync="batch", n_stat_blocks=2), case)
    for sw in range(3):
        tot = no.kmeans_batch_sweep(ref, n_blocks=2)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), sw
        c, rc = seg.acoustic_model.components, ref.acoustic_model.components
        assert np.array_equal(c.assignments, rc.assignments), sw
        assert np.array_equal(c.means, rc.means), sw
        assert rec["sum_neg_len_sqrd_norm"][0] == tot


def test_batch_sweep_with_thousands_of_new_components_in_one_sweep(gpu):
    """a first sweep with K << K_max.  The inactive rows of `means` are data points
    (kmeans_components.py:75-76, 149-166), so nearly every new token of the first batch sweep has an inactive row as its
    argmax and founds a component through add_item's `k > K -> K` clamp (:102-106): several thousand flagged tokens in one
    sweep, far beyond the 2048 the finalize kernel keeps in LDS (the rest goes through the context's overflow arrays).
    Bit-exact against the specification sweep; a per-block cap that is too small raises instead of corrupting silently."""
    from segmentalist_amd import _abi, kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    n_utt, D, K_max = 1500, 8, 4000
    # utterances longer than the window have no embedding for their single initial span: very few initial tokens
    corpus = make_corpus(n_utt, D, 200, seed=77, ragged=True, n_slices_max=6, N_range=(5, 12))
    kw = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.0, init_am_assignments="rand", wip=0)
    random.seed(2); np.random.seed(2)
    ref = no.SegmentalKMeansWordseg(K_max, *corpus, **kw)
    random.seed(2); np.random.seed(2)
    seg = kaw.SegmentalKMeansWordseg(K_max, *corpus, sync="batch", n_stat_blocks=8, **kw)
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    assert cr.K < K_max // 8                       # at most one token per utterance at the start: most rows are inactive
    K0 = cr.K
    for it in range(2):
        want = no.kmeans_batch_sweep(ref, n_blocks=8)
        rec = seg.segment(1)
        if it == 0:
            assert cr.K - K0 > 2048, (K0, cr.K)    # more components founded in ONE sweep than the LDS list holds
            assert int(seg._dk.n_flag.sum().item()) > 3000
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(cd.assignments, cr.assignments), it
        assert cd.K == cr.K
        assert np.array_equal(cd.counts, cr.counts)
        assert np.array_equal(cd.mean_numerators, cr.mean_numerators), it
        assert np.array_equal(cd.means, cr.means), it
        assert rec["sum_neg_len_sqrd_norm"][0] == want
    # the per-block cap is the one limit left, and exceeding it is loud
    random.seed(2); np.random.seed(2)
    small = kaw.SegmentalKMeansWordseg(K_max, *corpus, sync="batch", n_stat_blocks=8, flag_cap=64, **kw)
    with pytest.raises(_abi.SegkError, match="flag_cap"):
        small.segment(1)


@pytest.mark.parametrize("n_range,nmax,sync,n_blocks", [((3, 34), 6, "sequential", 1), ((3, 45), 6, "batch", 8), ((3, 80), 6, "sequential", 1),
                                                        ((3, 80), 6, "batch", 8), ((30, 80), 10, "batch", 4), ((30, 80), 10, "sequential", 1),
                                                        ((60, 80), 70, "sequential", 1), ((60, 80), 70, "batch", 4)])
def test_utterances_beyond_the_fast_kernels_landmark_limits(gpu, n_range, nmax, sync, n_blocks):
    """More than 32 landmarks (the persistent sequential chain and the chain's update kernel stop applying), more than 64 (the
    eight-lane segment kernel too) and windows of more than eight slices: the fall-backs behind the fast kernels, against the
    specification bit for bit -- boundaries, assignments, K, means and the record total over three sweeps, sequential mode (the
    visiting order shuffled from the same stream position on both sides) and batch mode."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(40, 16, 30, seed=11, ragged=True, n_slices_max=nmax, N_range=n_range)
    random.seed(5); np.random.seed(5)
    ref = no.SegmentalKMeansWordseg(30, *corpus, n_slices_max=nmax, init_am_assignments="spread")
    random.seed(5); np.random.seed(5)
    kw = dict(sync="batch", n_stat_blocks=n_blocks) if sync == "batch" else {}
    seg = kaw.SegmentalKMeansWordseg(30, *corpus, n_slices_max=nmax, init_am_assignments="spread", **kw)
    assert seg.utterances.boundaries.shape[1] > 32
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    for it in range(3):
        if sync == "batch":
            want = no.kmeans_batch_sweep(ref, n_blocks=n_blocks)
        else:
            st = random.getstate()
            want = ref.segment(1)["sum_neg_len_sqrd_norm"][0]
            random.setstate(st)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(cd.assignments, cr.assignments), it
        assert cd.K == cr.K
        assert np.array_equal(cd.means, cr.means), it
        assert rec["sum_neg_len_sqrd_norm"][0] == want, it


@pytest.mark.parametrize("D,K,sync,n_blocks,n_batches", [(16, 30, "sequential", 1, 1), (100, 130, "batch", 8, 1), (24, 40, "batch", 4, 3),
                                                         (7, 9, "batch", 2, 1)])
def test_float64_corpora_end_to_end(gpu, D, K, sync, n_blocks, n_batches):
    """float64 embeddings (the reference's dtype when the caller passes doubles: every score and statistic in float64, none of
    the matrix-core filters apply) through the sequential chain, whole-sweep batches and mini-batches, against the specification
    bit for bit over three sweeps."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(120, D, K, seed=13, ragged=True, n_slices_max=6, N_range=(3, 14), dtype=np.float64)
    random.seed(5); np.random.seed(5)
    ref = no.SegmentalKMeansWordseg(K, *corpus, n_slices_max=6, init_am_assignments="spread")
    random.seed(5); np.random.seed(5)
    kw = dict(sync="batch", n_stat_blocks=n_blocks, n_batches=n_batches) if sync == "batch" else {}
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=6, init_am_assignments="spread", **kw)
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    assert cd.means.dtype == np.float64
    totals = np.zeros(ref.utterances.D)
    for it in range(3):
        if sync == "batch":
            want = no.kmeans_batch_sweep(ref, n_blocks=n_blocks) if n_batches == 1 else no.kmeans_minibatch_sweep(ref, n_blocks, n_batches, totals)
        else:
            st = random.getstate()
            want = ref.segment(1)["sum_neg_len_sqrd_norm"][0]
            random.setstate(st)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(cd.assignments, cr.assignments), it
        assert cd.K == cr.K
        assert np.array_equal(cd.means, cr.means), it
        assert rec["sum_neg_len_sqrd_norm"][0] == want, it


@pytest.mark.parametrize("n_utt,D,K,n_blocks,n_batches,n_range", [(5, 8, 4, 8, 1, (3, 9)), (1, 8, 3, 8, 1, (3, 9)), (3, 8, 1, 2, 1, (1, 3)),
                                                                  (9, 5, 6, 8, 4, (1, 4)), (2, 3, 2, 1, 1, (1, 2)), (17, 129, 7, 3, 2, (2, 6))])
def test_batch_mode_on_tiny_corpora(gpu, n_utt, D, K, n_blocks, n_batches, n_range):
    """Fewer utterances than statistics blocks (empty blocks), one utterance, one component, one- and two-landmark utterances,
    three to 129 dimensions, more mini-batches than a block has utterances: whole-sweep and mini-batch steps against the
    specification bit for bit."""
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(n_utt, D, K, seed=31, ragged=True, n_slices_max=4, N_range=n_range)
    random.seed(5); np.random.seed(5)
    ref = no.SegmentalKMeansWordseg(K, *corpus, n_slices_max=4, init_am_assignments="spread")
    random.seed(5); np.random.seed(5)
    seg = kaw.SegmentalKMeansWordseg(K, *corpus, n_slices_max=4, init_am_assignments="spread", sync="batch", n_stat_blocks=n_blocks,
                                     n_batches=n_batches)
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    totals = np.zeros(ref.utterances.D)
    for it in range(3):
        want = no.kmeans_batch_sweep(ref, n_blocks=n_blocks) if n_batches == 1 else no.kmeans_minibatch_sweep(ref, n_blocks, n_batches, totals)
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(cd.assignments, cr.assignments), it
        assert cd.K == cr.K
        assert np.array_equal(cd.means, cr.means), it
        assert rec["sum_neg_len_sqrd_norm"][0] == want, it
