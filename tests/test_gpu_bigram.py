"""GPU parity of the bigram segmenter (BASELINE config 5) through the C ABI: BigramSmoothLM counts
resident on the device, LM priors inside segk_fbgmm_score / segk_fbgmm_assign, LM rewiring in
del_component -- against golden chains generated from the reference (tests/golden/bigram.npz)
and against the oracle on fresh seeds.  Integer state bit-exact; log-likelihoods 1e-8 relative
(the north-star tolerance is 1e-4)."""
import random

import numpy as np
import numpy.testing as npt
import pytest

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


def _make(mod, prior_cls, chain, seed=1, **kw):
    name, n_utt, D, K, cseed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, cseed, ragged, N, nmax, dtype)
    random.seed(seed)
    np.random.seed(seed)
    prior = prior_cls(*cases.fixed_prior_params(D))
    args = dict(covariance_type=cov, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1,
                lms=1.0, wip=0.0, fb_type="unigram", init_am_assignments="rand", time_power_term=1.0)
    args.update(kw)
    return mod.BigramAcousticWordseg(K, prior, dict(cases.BIGRAM_LM), *corpus, **args)


@pytest.mark.parametrize("chain", cases.BIGRAM_CHAINS, ids=[c[0] for c in cases.BIGRAM_CHAINS])
def test_bigram_chain_vs_reference(gpu, golden, chain):
    from segmentalist_amd import bigram_acoustic_wordseg as baw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    g = golden("bigram")
    name = chain[0]
    seg = _make(baw, FixedVarPrior, chain)
    c = seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, g[name + "_init_bounds"])
    assert np.array_equal(c.assignments, g[name + "_init_assign"])
    assert np.array_equal(seg.lm.unigram_counts, g[name + "_init_unigram"])
    assert np.array_equal(seg.lm.bigram_counts, g[name + "_init_bigram"])
    for it in range(4):
        rec = seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, g[name + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[name + "_assign"][it]), it
        assert np.array_equal(seg.lm.unigram_counts, g[name + "_unigram"][it]), it
        assert np.array_equal(seg.lm.bigram_counts, g[name + "_bigram"][it]), it
        for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z"]:
            npt.assert_allclose(rec[k][0], g[name + "_rec_" + k][it], rtol=1e-8, err_msg=k)
        assert rec["components"][0] == g[name + "_rec_components"][it]


@pytest.mark.parametrize("kw", [dict(lms=0.7, wip=-0.3, time_power_term=1.3), dict(anneal=True)],
                         ids=["lms_wip_tpt", "anneal_am"])
def test_bigram_chain_vs_oracle_other_settings(gpu, kw):
    """Settings the golden chains do not cover: lms / wip / time_power_term and annealed assignment."""
    from segmentalist_amd import bigram_acoustic_wordseg as baw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    kw = dict(kw)
    anneal = kw.pop("anneal", False)
    chain = ("x", 12, 8, 9, 77, True, 0, 5, "float32", "fixed")
    no.set_shuffle("py3")
    ref = _make(no, no.FixedVarPrior, chain, seed=5, **kw)
    seg = _make(baw, FixedVarPrior, chain, seed=5, **kw)
    c = seg.acoustic_model.components
    for it in range(3):
        # the oracle draws from the same process-global stream: run the two sides from one state
        st, nst = random.getstate(), np.random.get_state()
        if anneal:
            order = list(range(ref.utterances.D))
            no._shuffle(order)
            for i in order:
                ref.gibbs_sample_i(i, 2.0, True)
        else:
            ref.gibbs_sample(1)
        random.setstate(st)
        np.random.set_state(nst)
        if anneal:
            seg.gibbs_sample(1, anneal_schedule=None, anneal_end_temp_inv=2.0, anneal_gibbs_am=True)
        else:
            seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(c.assignments, ref.acoustic_model.components.assignments), it
        assert np.array_equal(seg.lm.unigram_counts, ref.lm.unigram_counts), it
        assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts), it
        npt.assert_allclose(seg.log_marg(), ref.log_marg(), rtol=1e-8)


def test_bigram_per_embedding_api_and_assignments_only(gpu):
    """log_marg_i_embed_unigram / gibbs_sample_inside_loop_i_embed one at a time, then an
    assignments_only sweep (boundaries frozen) against the oracle's per-embedding functions."""
    from segmentalist_amd import bigram_acoustic_wordseg as baw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    chain = ("y", 8, 6, 7, 31, True, 0, 4, "float32", "fixed")
    no.set_shuffle("py3")
    ref = _make(no, no.FixedVarPrior, chain, seed=9)
    seg = _make(baw, FixedVarPrior, chain, seed=9)
    rc, c = ref.acoustic_model.components, seg.acoustic_model.components
    embeds = [e for e in ref.utterances.get_segmented_embeds_i(2) if e != -1]
    # scoring with everything assigned
    for e in embeds:
        npt.assert_allclose(seg.log_marg_i_embed_unigram(e), ref.log_marg_i_embed_unigram(e), rtol=1e-10)
    # remove utterance 2, re-add its segments one at a time with the bigram prior
    ref.lm.remove_counts_from_utterance(ref.get_unsup_transcript_i(2))
    seg.lm.remove_counts_from_utterance(seg.get_unsup_transcript_i(2))
    for e in embeds:
        rc.del_item(e)
        c.del_item(e)
    assert np.array_equal(seg.lm.unigram_counts, ref.lm.unigram_counts)
    assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts)
    jr = jp = None
    for e in embeds:
        st = random.getstate()
        jr = ref.gibbs_sample_inside_loop_i_embed(e, jr, 1, u=random.random())
        random.setstate(st)
        jp = seg.gibbs_sample_inside_loop_i_embed(e, jp)
        assert jr == jp
    ref.lm.counts_from_utterance(ref.get_unsup_transcript_i(2))
    seg.lm.counts_from_utterance(seg.get_unsup_transcript_i(2))
    assert np.array_equal(c.assignments, rc.assignments)
    assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts)

    # assignments_only: boundaries stay, every utterance is re-assigned in shuffled order
    before = seg.utterances.boundaries.copy()
    st = random.getstate()
    order = list(range(ref.utterances.D))
    no._shuffle(order)
    for i in order:
        ref.lm.remove_counts_from_utterance(ref.get_unsup_transcript_i(i))
        es = [e for e in ref.utterances.get_segmented_embeds_i(i) if e != -1]
        for e in es:
            rc.del_item(e)
        j = None
        for e in es:
            j = ref.gibbs_sample_inside_loop_i_embed(e, j, 1, u=random.random())
        ref.lm.counts_from_utterance(ref.get_unsup_transcript_i(i))
    random.setstate(st)
    rec = seg.gibbs_sample(1, assignments_only=True)
    assert rec["log_marg*length"][0] == 0
    assert np.array_equal(seg.utterances.boundaries, before)
    assert np.array_equal(c.assignments, rc.assignments)
    assert np.array_equal(seg.lm.unigram_counts, ref.lm.unigram_counts)
    assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts)


def test_reference_test_bigram_lms_on_device_counts(gpu):
    """tests/test_bigram_lms.py:13-76 of the reference against the device-backed count tables."""
    from segmentalist_amd.bigram_lms import BigramSmoothLM
    lm = BigramSmoothLM(0.1, 1, 2, 5)
    lm.counts_from_data([[1, 1, 3, 4, 0], [4, 4], [1, 0, 2, 2, 2, 2, 3, 1], [3, 3, 1]])
    npt.assert_almost_equal(lm.prob_i_given_j(1, 3), 0.1 * lm.prob_i(1) + 0.9 * (2. + 2. / 5) / (4 + 2))
    npt.assert_almost_equal(lm.prob_i(1), (5. + 1. / 5) / (18 + 1))
    pv, pj = lm.prob_vec_i(), lm.prob_vec_given_j(3)
    for i in range(5):
        assert pv[i] == lm.prob_i(i)
        npt.assert_almost_equal(pj[i], lm.prob_i_given_j(i, 3))
        npt.assert_almost_equal(lm.log_prob_vec_i()[i], np.log(lm.prob_i(i)))
        npt.assert_almost_equal(lm.log_prob_vec_given_j(3)[i], np.log(lm.prob_i_given_j(i, 3)))
    lm.remove_counts_from_utterance([3, 3, 1])
    assert lm.unigram_counts.sum() == 15 and lm.bigram_counts[3, 3] == 0


def test_persistent_chain_with_a_language_model_equals_the_launches_per_utterance(gpu, monkeypatch):
    """segk_fbgmm_sequential_sweep with a language model attached (every workgroup replays the updates on its own copy of
    the bigram counts) against the six launches per utterance (SEGK_FB_CHAIN=0) from identical states: boundaries,
    assignments, statistics, both count tables, K, the record values and the position of the RNG stream bit for bit, over
    sweeps in which components empty (their rows and columns of the bigram table move)."""
    from segmentalist_amd import bigram_acoustic_wordseg as baw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.synth import make_corpus
    D, K = 12, 30
    corpus = make_corpus(60, D, K, seed=4, ragged=True, n_slices_max=5, N_range=(3, 14))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SEGK_FB_CHAIN", mode)
        random.seed(3)
        np.random.seed(3)
        seg = baw.BigramAcousticWordseg(K, FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D)),
                                        dict(cases.BIGRAM_LM), *corpus, covariance_type="fixed", fb_type="unigram",
                                        n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=0.8,
                                        wip=-0.1, init_am_assignments="rand", time_power_term=1.0)
        rec = seg.gibbs_sample(4, anneal_schedule="linear", anneal_gibbs_am=True)
        df = seg._df
        out[mode] = dict(b=seg.utterances.boundaries.copy(), a=seg.acoustic_model.components.assignments.copy(),
                         sa=df.stat_a.cpu().numpy(), sb=df.stat_b.cpu().numpy(), pr=df.pred.cpu().numpy(),
                         cn=df.counts.cpu().numpy(), uni=np.array(seg.lm.unigram_counts), big=np.array(seg.lm.bigram_counts),
                         K=int(df.K.item()), rec={k: list(v) for k, v in rec.items() if k != "sample_time"}, rnd=random.random())
    assert min(out["1"]["rec"]["components"]) < K, "no component emptied: the test does not cover the relaunches"
    assert out["1"]["big"].sum() > 0
    for k in out["1"]:
        if isinstance(out["1"][k], np.ndarray):
            assert np.array_equal(out["1"][k], out["0"][k]), k
        else:
            assert out["1"][k] == out["0"][k], k


@pytest.mark.parametrize("n_utt,D,K,n_range,nmax", [(20, 300, 8, (3, 14), 5), (24, 12, 9, (3, 80), 6), (24, 12, 9, (30, 80), 10),
                                                   (30, 520, 6, (3, 6), 5)])
def test_bigram_chain_vs_oracle_wide_rows_and_long_utterances(gpu, n_utt, D, K, n_range, nmax):
    """Shapes the golden chains do not reach: D = 300 and D = 520 (more dimensions than threads in the update kernels; beyond
    the tabulated log prior predictive), utterances of up to 80 landmarks and a window of ten slices (the launches per
    utterance instead of the persistent kernel) -- boundaries, assignments and the language model's counts equal to the
    oracle's, log_marg within 1e-8 relative, over three sweeps from the same stream positions."""
    from segmentalist_amd import bigram_acoustic_wordseg as baw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.synth import make_corpus
    no.set_shuffle("py3")
    corpus = make_corpus(n_utt, D, K, seed=77, ragged=True, n_slices_max=nmax, N_range=n_range)
    sides = []
    for mod, pc in ((no, no.FixedVarPrior), (baw, FixedVarPrior)):
        random.seed(5); np.random.seed(5)
        args = dict(covariance_type="fixed", n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1,
                    lms=1.0, wip=0.0, fb_type="unigram", init_am_assignments="rand", time_power_term=1.0)
        sides.append(mod.BigramAcousticWordseg(K, pc(*cases.fixed_prior_params(D)), dict(cases.BIGRAM_LM), *corpus, **args))
    ref, seg = sides
    for it in range(3):
        st, nst = random.getstate(), np.random.get_state()
        ref.gibbs_sample(1)
        random.setstate(st)
        np.random.set_state(nst)
        seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries), it
        assert np.array_equal(seg.acoustic_model.components.assignments, ref.acoustic_model.components.assignments), it
        assert np.array_equal(seg.lm.unigram_counts, ref.lm.unigram_counts), it
        assert np.array_equal(seg.lm.bigram_counts, ref.lm.bigram_counts), it
        npt.assert_allclose(seg.log_marg(), ref.log_marg(), rtol=1e-8)
