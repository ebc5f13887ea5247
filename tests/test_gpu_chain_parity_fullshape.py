"""
The reference's own chains (sync="sequential", the API default -- the only mode whose results ARE the reference's) at the
BASELINE shapes, HIP path against the oracle chain on the same seeds and the same process-global `random` stream:

  * SegmentalKMeansWordseg.segment (kmeans_acoustic_wordseg.py:225-332, 393-399) at configs[2] shape -- D = 100,
    K = 1000, 20 landmarks, 300 utterances, 2 sweeps -- through the persistent kernel (SEGK_SEQ_CHAIN=1) and through
    the three launches per utterance (=0): boundaries, assignments, means, numerators, counts, K and the record values
    bit for bit;
  * UnigramAcousticWordseg + FBGMM diag / fixed (unigram_acoustic_wordseg.py:252-360, fbgmm.py:422-463) and
    BigramAcousticWordseg (bigram_acoustic_wordseg.py:386-551) at configs[1] shape -- D = 39, K = 100, 20 landmarks,
    200 utterances, 2 sweeps: boundaries, assignments, counts (and LM tables) exact, record values 1e-8.

The small golden chains (tests/test_gpu_kmeans.py, test_gpu_fbgmm.py, test_gpu_bigram.py) pin the same code to outputs of
the reference itself at D <= 16, K <= 12; these pin it at the sizes users and bench.py run.  The oracle takes ~8 s / ~5 s
per case on one host core.
"""
import random

import numpy as np
import numpy.testing as npt
import pytest

from oracle import np_oracle as no

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


@pytest.fixture(scope="module")
def kmeans_reference_chain():
    """The oracle's chain at configs[2] shape, computed once for both device forms: list of per-sweep states."""
    from segmentalist_amd.synth import make_corpus
    corpus = make_corpus(300, 100, 1000, seed=0, N=20, n_slices_max=6)
    no.set_shuffle("py3")
    random.seed(3)
    np.random.seed(3)
    ref = no.SegmentalKMeansWordseg(1000, *corpus, n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                    init_am_assignments="spread", wip=0)
    c = ref.acoustic_model.components
    states = [dict(bounds=ref.utterances.boundaries.copy(), assign=c.assignments.copy(), random_means=c.random_means.copy())]
    for it in range(2):
        random.seed(100 + it)
        rec = ref.segment(1)
        states.append(dict(bounds=ref.utterances.boundaries.copy(), assign=c.assignments.copy(), means=c.means.copy(),
                           numer=c.mean_numerators.copy(), counts=c.counts.copy(), K=c.K, rec=rec))
    return corpus, states


@pytest.mark.parametrize("chain_kernel", ["1", "0"], ids=["persistent_kernel", "three_launches"])
def test_kmeans_sequential_chain_headline_shape_vs_oracle(gpu, monkeypatch, kmeans_reference_chain, chain_kernel):
    from segmentalist_amd import kmeans_acoustic_wordseg as kaw
    corpus, want = kmeans_reference_chain
    monkeypatch.setenv("SEGK_SEQ_CHAIN", chain_kernel)
    random.seed(3)
    np.random.seed(3)
    seg = kaw.SegmentalKMeansWordseg(1000, *corpus, n_slices_min=0, n_slices_max=6, p_boundary_init=0.5,
                                     init_am_assignments="spread", wip=0)      # sync="sequential" is the default
    c = seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, want[0]["bounds"])
    assert np.array_equal(c.assignments, want[0]["assign"])
    assert np.array_equal(c.random_means, want[0]["random_means"])
    for it in range(2):
        random.seed(100 + it)
        rec = seg.segment(1)
        w = want[it + 1]
        assert np.array_equal(seg.utterances.boundaries, w["bounds"]), it
        assert np.array_equal(c.assignments, w["assign"]), it
        assert c.K == w["K"], it
        got_means = c.means
        assert got_means.dtype == w["means"].dtype == np.float32
        assert np.array_equal(got_means, w["means"]), it          # all K_max rows: stale and random rows included
        assert np.array_equal(c.mean_numerators, w["numer"]), it
        assert np.array_equal(c.counts, w["counts"]), it
        assert rec["sum_neg_len_sqrd_norm"][0] == w["rec"]["sum_neg_len_sqrd_norm"][0], it      # bit for bit
        assert rec["n_tokens"][0] == w["rec"]["n_tokens"][0] and rec["components"][0] == w["rec"]["components"][0]
        assert np.isclose(rec["sum_neg_sqrd_norm"][0], w["rec"]["sum_neg_sqrd_norm"][0], rtol=1e-10, atol=0)
    # the chain really moved (a frozen state would compare equal trivially)
    assert not np.array_equal(want[1]["bounds"], want[0]["bounds"]) and not np.array_equal(want[2]["assign"], want[1]["assign"])


def _fb_build(mods, kind, corpus, D, K, seed):
    random.seed(seed)
    np.random.seed(seed)
    args = dict(n_slices_min=0, n_slices_max=6, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                init_am_assignments="rand", time_power_term=1.0)
    fixed = mods["FixedVarPrior"](0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D))
    if kind == "bigram":
        return mods["BigramAcousticWordseg"](K, fixed, {"type": "smooth", "intrp_lambda": 0.1, "a": 0.5, "b": 0.5}, *corpus,
                                             covariance_type="fixed", fb_type="unigram", **args)
    prior = fixed if kind == "fixed" else mods["NIW"](np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    return mods["UnigramAcousticWordseg"](mods["FBGMM"], 1.0, K, prior, *corpus, covariance_type=kind, fb_type="standard",
                                          **args)


@pytest.mark.parametrize("kind", ["diag", "fixed", "bigram"])
def test_fbgmm_serial_chains_config2_shape_vs_oracle(gpu, kind):
    from segmentalist_amd import bigram_acoustic_wordseg as baw, fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K, n_utt = 39, 100, 200
    corpus = make_corpus(n_utt, D, K, seed=0, N=20, n_slices_max=6)
    no.set_shuffle("py3")
    ref = _fb_build(dict(FixedVarPrior=no.FixedVarPrior, NIW=no.NIW, FBGMM=no.FBGMM,
                         UnigramAcousticWordseg=no.UnigramAcousticWordseg,
                         BigramAcousticWordseg=no.BigramAcousticWordseg), kind, corpus, D, K, 7)
    seg = _fb_build(dict(FixedVarPrior=FixedVarPrior, NIW=NIW, FBGMM=fbgmm.FBGMM,
                         UnigramAcousticWordseg=uaw.UnigramAcousticWordseg,
                         BigramAcousticWordseg=baw.BigramAcousticWordseg), kind, corpus, D, K, 7)
    rc, c = ref.acoustic_model.components, seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries)
    assert np.array_equal(c.assignments, rc.assignments)
    moved = False
    for it in range(2):
        # both sides consume the process-global `random` stream (shuffle, one uniform per backward step and per
        # assignment): run them from one state and check that they leave it in the same place
        st = random.getstate()
        before = ref.utterances.boundaries.copy()
        rec_ref = ref.gibbs_sample(1)
        after_ref = random.random()
        random.setstate(st)
        rec = seg.gibbs_sample(1)
        assert random.random() == after_ref, "the device chain consumed a different number of uniforms"
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(c.assignments, rc.assignments), it
        assert c.K == rc.K and np.array_equal(c.counts, rc.counts), it
        if kind == "bigram":
            assert np.array_equal(seg.lm.unigram_counts, ref.lm.unigram_counts), it
            assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts), it
        for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z"]:
            npt.assert_allclose(rec[k][0], rec_ref[k][0], rtol=1e-8, err_msg=k)
        assert rec["components"][0] == rec_ref["components"][0] and rec["n_tokens"][0] == rec_ref["n_tokens"][0]
        moved = moved or not np.array_equal(before, ref.utterances.boundaries)
    assert moved
    # the statistics the next sweep would score against
    names = (["m_N_numerators", "S_N_partials", "log_prod_vars", "inv_vars"] if kind == "diag" else
             ["mu_N_numerators", "precision_Ns", "log_prod_precision_preds", "precision_preds"])
    for nm in names:
        npt.assert_allclose(getattr(c, nm)[:c.K], getattr(rc, nm)[:rc.K], rtol=1e-10, atol=1e-300, err_msg=nm)
