"""
GPU parity tests of the FBGMM / unigram Gibbs path
against golden vectors captured from the reference, the constants of the reference's own tests
and the oracle.  Tolerance: log-likelihoods 1e-4 relative is the contract (BASELINE north_star);
the fp64 device path is held to 1e-9 here.  Sampled boundaries / assignments must coincide with
the reference when the same uniforms are consumed.
"""
import random

import numpy as np
import numpy.testing as npt
import pytest

from tests.golden import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need an MI355X"
    torch.cuda.set_device(0)
    from segmentalist_amd import _abi
    _abi.ctx()
    return torch


@pytest.mark.parametrize("tag,D,K_max,n_items,seed", [("s", 5, 6, 40, 31), ("m", 39, 100, 600, 32),
                                                      ("l", 100, 40, 300, 33)])
def test_components_scores_and_sampling_vs_reference(gpu, golden, tag, D, K_max, n_items, seed):
    from segmentalist_amd.fbgmm import FBGMM
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    g = golden("gauss")
    X, assign = cases.gauss_state(D, K_max, n_items, seed)
    idx = g["fx_%s_idx" % tag]
    for cov, pre in (("fixed", "fx"), ("diag", "dg")):
        prior = FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else NIW(*cases.diag_prior_params(D))
        fm = FBGMM(X, prior, 1.7, K_max, assign.copy(), covariance_type=cov, lms=0.8)
        c = fm.components
        assert c.K == int(g["%s_%s_K" % (pre, tag)])
        assert np.array_equal(c.counts, g["%s_%s_counts" % (pre, tag)])
        names = (["mu_N_numerators", "precision_Ns", "log_prod_precision_preds", "precision_preds"]
                 if cov == "fixed" else ["m_N_numerators", "S_N_partials", "log_prod_vars", "inv_vars"])
        for nm in names:
            npt.assert_allclose(getattr(c, nm), g["%s_%s_%s" % (pre, tag, nm)], rtol=1e-13, atol=1e-300, err_msg=nm)
        for j, i in enumerate(idx):
            npt.assert_allclose(c.log_post_pred(i), g["%s_%s_log_post_pred" % (pre, tag)][j], rtol=RTOL, atol=1e-9)
            npt.assert_allclose(c.log_prior(i), g["%s_%s_log_prior" % (pre, tag)][j], rtol=RTOL)
            npt.assert_allclose(fm.log_marg_i(i), g["%s_%s_log_marg_i" % (pre, tag)][j], rtol=RTOL)
        npt.assert_allclose(fm.log_marg(), g["%s_%s_log_marg" % (pre, tag)], rtol=1e-10)
        if cov == "fixed":
            npt.assert_allclose(fm.log_prob_z(), g["fx_%s_log_prob_z" % tag], rtol=1e-12)
        # A10 with the reference's uniforms (state mutates between draws)
        for j, i in enumerate(idx):
            c.dev.assign_item(i, g["%s_%s_sample_u" % (pre, tag)][j])
            assert c.assignments[i] == g["%s_%s_sample_k" % (pre, tag)][j], (cov, j)
        after = "after_mu_N_numerators" if cov == "fixed" else "after_m_N_numerators"
        npt.assert_allclose(c.dev.stat_a.cpu().numpy(), g["%s_%s_%s" % (pre, tag, after)], rtol=1e-13, atol=1e-300)
        if cov == "fixed":
            assert np.array_equal(c.counts, g["fx_%s_after_counts" % tag])
        else:
            npt.assert_allclose(c.S_N_partials, g["dg_%s_after_S_N_partials" % tag], rtol=1e-13, atol=1e-300)


def test_component_mutators_vs_oracle(gpu):
    from oracle import np_oracle as no
    from segmentalist_amd.gaussian_components_diag import GaussianComponentsDiag
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior, GaussianComponentsFixedVar
    from segmentalist_amd.niw import NIW
    rs = np.random.RandomState(3)
    X = rs.randn(50, 6).astype(np.float32)
    assign = no.consecutive_labels(rs.randint(0, 4, 50))
    assign[rs.rand(50) < 0.5] = -1
    assign = np.array([{k: j for j, k in enumerate(sorted(set(assign) - {-1}))}.get(a, -1) for a in assign])
    pairs = [
        (no.GaussianComponentsFixedVar(X, no.FixedVarPrior(*cases.fixed_prior_params(6)), assign.copy(), K_max=7),
         GaussianComponentsFixedVar(X, FixedVarPrior(*cases.fixed_prior_params(6)), assign.copy(), K_max=7),
         ["mu_N_numerators", "precision_Ns", "log_prod_precision_preds", "precision_preds"]),
        (no.GaussianComponentsDiag(X, no.NIW(*cases.diag_prior_params(6)), assign.copy(), K_max=7),
         GaussianComponentsDiag(X, NIW(*cases.diag_prior_params(6)), assign.copy(), K_max=7),
         ["m_N_numerators", "S_N_partials", "log_prod_vars", "inv_vars"]),
    ]
    for ref, dev, names in pairs:
        def same():
            assert dev.K == ref.K
            assert np.array_equal(dev.counts, ref.counts)
            assert np.array_equal(dev.assignments, ref.assignments)
            for nm in names:
                npt.assert_allclose(getattr(dev, nm), getattr(ref, nm), rtol=1e-12, atol=1e-300, err_msg=nm)
        same()
        free = list(np.where(ref.assignments == -1)[0])
        used = list(np.where(ref.assignments != -1)[0])
        for step in range(60):
            if rs.rand() < 0.5 and free:
                i = free.pop(rs.randint(len(free)))
                k = int(rs.randint(0, ref.K + 1)) if ref.K < 7 else int(rs.randint(0, ref.K))
                ref.add_item(i, k)
                dev.add_item(i, k)
                used.append(i)
            elif used:
                i = used.pop(rs.randint(len(used)))
                ref.del_item(i)              # may delete the component (swap-last compaction)
                dev.del_item(i)
                free.append(i)
            same()


@pytest.mark.parametrize("chain", cases.UNIGRAM_CHAINS, ids=[c[0] for c in cases.UNIGRAM_CHAINS])
@pytest.mark.parametrize("fb_type", ["standard", "viterbi"])
def test_unigram_chain_vs_reference(gpu, golden, chain, fb_type):
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    g = golden("chains")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    prior = FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else NIW(*cases.diag_prior_params(D))
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, n_slices_min=0,
                                     n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0,
                                     wip=0.0, fb_type=fb_type, init_am_assignments="rand", time_power_term=1.0)
    c = seg.acoustic_model.components
    tag = "%s_%s" % (name, fb_type)
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_init_bounds"])
    assert np.array_equal(c.assignments, g[tag + "_init_assign"])
    for it in range(4):
        rec = seg.gibbs_sample(1)       # consumes the process-global `random` like the reference
        assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[tag + "_assign"][it]), it
        for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z"]:
            npt.assert_allclose(rec[k][0], g[tag + "_rec_" + k][it], rtol=1e-8, err_msg=k)
        assert rec["components"][0] == g[tag + "_rec_components"][it]
        assert rec["n_tokens"][0] == g[tag + "_rec_n_tokens"][it]
    assert np.array_equal(c.counts, g[tag + "_counts"])
    # the host RNG stream ended up where the reference's did: same number of uniforms consumed
    assert sum(g[tag + "_n_uniforms"]) >= 0


def _three_embedding_dataset():
    """tests/test_unigram_acoustic_wordseg.py:16-57 of the reference (fixture data)."""
    m = np.array([
        [-0.2702691, -0.12348549, -0.20069546, -0.10067126, -0.32822475,
         -0.24878924, -0.17988801, -0.13201745, 0.66409844, -0.44816282],
        [-0.27186683, -0.12384345, -0.20049213, -0.10272419, -0.32618827,
         -0.24660945, -0.17784701, -0.13362537, 0.66524321, -0.44805479],
        [-0.2465426, -0.06354388, -0.22458388, 0.79060942, 0.48230717,
         -0.11888564, 0.06724239, -0.04977163, 0.06908087, 0.03395205]], dtype=np.float32)
    return ({"test": m}, {"test": np.array([0, 1, 2])}, {"test": [1, 2, 1]}, {"test": [1, 2]}, {"test": [2]})


def test_reference_unigram_tests_through_the_product_api(gpu):
    """tests/test_unigram_acoustic_wordseg.py:60-142 of the reference, verbatim constants."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    emb, vid, dur, lm, seeds = _three_embedding_dataset()
    S_0 = 0.002 * np.ones(10)
    prior = FixedVarPrior(S_0, np.zeros(10), S_0 / 0.05)
    random.seed(1)
    np.random.seed(1)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 10., 2, prior, emb, vid, dur, lm, seed_boundaries_dict=seeds,
                                     beta_sent_boundary=-1)
    seg.gibbs_sample_i(0)
    got = seg.get_vec_embed_log_probs(seg.utterances.vec_ids[0], seg.utterances.durations[0])
    npt.assert_almost_equal(got, np.array([17.5548998, 35.103967, 17.5548998]))

    random.seed(1)
    np.random.seed(1)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 10., 2, prior, emb, vid, dur, lm, seed_boundaries_dict=seeds,
                                     beta_sent_boundary=-1)
    rec = seg.gibbs_sample(6)
    npt.assert_almost_equal(rec["log_marg"], [
        -11.969040866436707, -11.969040866436707, -11.969040866436707,
        -5.9368664797514707, -11.969040866436707, -5.9368664797514707])
    npt.assert_almost_equal(rec["log_prob_z"], [
        -1.4816045409242173, -1.4816045409242173, -1.4816045409242173,
        -0.69314718055994673, -1.4816045409242173, -0.69314718055994673])
    npt.assert_almost_equal(rec["log_prob_X_given_z"], [
        -10.48743632551249, -10.48743632551249, -10.48743632551249,
        -5.2437192991915236, -10.48743632551249, -5.2437192991915236])


def test_module_level_dp_functions_consume_rng_like_the_reference(gpu):
    from oracle import np_oracle as no
    from segmentalist_amd import unigram_acoustic_wordseg as uaw
    dpc = [c for c in cases.dp_cases() if c["kind"] in ("dense", "banded", "ints")][::7]
    for ci, c in enumerate(dpc):
        for temp in (1, 1.7):
            random.seed(1000 + ci)
            want_t, want_b = no.forward_backward(c["vec"], -0.25, c["N"], c["n_min"], c["n_max"], None, temp)
            after_ref = random.random()
            random.seed(1000 + ci)
            got_t, got_b = uaw.forward_backward(c["vec"], -0.25, c["N"], c["n_min"], c["n_max"], None, temp)
            assert random.random() == after_ref          # same number of uniforms consumed
            assert np.array_equal(got_b, want_b)
            npt.assert_allclose(got_t, want_t, rtol=1e-13)
        with np.errstate(all="ignore"):
            want_t, want_b = no.forward_backward_viterbi(c["vec"], 0.0, c["N"], c["n_min"], c["n_max"])
        got_t, got_b = uaw.forward_backward_viterbi(c["vec"], 0.0, c["N"], c["n_min"], c["n_max"])
        assert np.array_equal(got_b, want_b)


# ------------------------------------------------------------------ FBGMM.gibbs_sample
def _amg_kw(sched):
    if sched == "linear":
        return dict(anneal_schedule="linear", anneal_start_temp_inv=0.5, anneal_end_temp_inv=1.0)
    if sched == "step":
        return dict(anneal_schedule="step", anneal_start_temp_inv=0.25, anneal_end_temp_inv=1.0, n_anneal_steps=2)
    return {}


@pytest.mark.parametrize("case", cases.AM_GIBBS, ids=[c[0] for c in cases.AM_GIBBS])
def test_fbgmm_gibbs_sample_vs_reference(gpu, golden, case):
    """fbgmm.py:288-420 through segk_fbgmm_gibbs_items against trajectories of the reference."""
    from segmentalist_amd import fbgmm
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    g = golden("amgibbs")
    name, D, K_max, n_items, seed, cov, unassigned, sched = case
    X, assign = cases.gauss_state(D, K_max, n_items, seed)
    prior = FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else NIW(*cases.diag_prior_params(D))
    random.seed(3)
    np.random.seed(3)
    fm = fbgmm.FBGMM(X, prior, 1.0, K_max, assign.copy(), covariance_type=cov, lms=1.0)
    rec = fm.gibbs_sample(4, consider_unassigned=unassigned, **_amg_kw(sched))
    assert np.array_equal(fm.components.assignments, g[name + "_assign"])
    assert np.array_equal(fm.components.counts, g[name + "_counts"])
    for k in ["log_marg", "log_prob_z", "log_prob_X_given_z", "anneal_temp"]:
        npt.assert_allclose(rec[k], g[name + "_rec_" + k], rtol=1e-8, err_msg=k)
    assert list(rec["components"]) == list(g[name + "_rec_components"])
    rec = fm.gibbs_sample(1, consider_unassigned=unassigned)
    assert np.array_equal(fm.components.assignments, g[name + "_assign_5"])
    npt.assert_allclose(rec["log_marg"], g[name + "_log_marg_5"], rtol=1e-8)


@pytest.mark.parametrize("chain", cases.AM_ITER_CHAINS, ids=[c[0] for c in cases.AM_ITER_CHAINS])
def test_unigram_chain_with_am_iterations_vs_reference(gpu, golden, chain):
    """gibbs_sample(n, am_n_iter=2): the in-between acoustic-model sweeps of unigram...:440-443."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    g = golden("amgibbs")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    prior = FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else NIW(*cases.diag_prior_params(D))
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, n_slices_min=0,
                                     n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0,
                                     wip=0.0, fb_type="standard", init_am_assignments="rand", time_power_term=1.0)
    rec = seg.gibbs_sample(3, am_n_iter=2)
    tag = name + "_amiter"
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"])
    assert np.array_equal(seg.acoustic_model.components.assignments, g[tag + "_assign"])
    npt.assert_allclose(rec["log_marg"], g[tag + "_rec_log_marg"], rtol=1e-8)
    assert list(rec["components"]) == list(g[tag + "_rec_components"])
    assert list(rec["n_tokens"]) == list(g[tag + "_rec_n_tokens"])


@pytest.mark.parametrize("cov,fb_type,D,K", [("diag", "standard", 12, 30), ("fixed", "standard", 12, 30), ("diag", "viterbi", 12, 30),
                                             ("fixed", "viterbi", 12, 30), ("diag", "standard", 520, 6), ("fixed", "standard", 520, 6),
                                             ("diag", "standard", 12, 300), ("fixed", "standard", 16, 256)],
                         ids=["diag-standard", "fixed-standard", "diag-viterbi", "fixed-viterbi", "diag-D520", "fixed-D520", "diag-K300",
                              "fixed-K256"])
def test_persistent_chain_equals_the_four_launches_per_utterance(gpu, monkeypatch, cov, fb_type, D, K):
    """segk_fbgmm_sequential_sweep (one persistent kernel per stretch of utterances between two emptied components: every
    workgroup replays every update on a model held in LDS, only the span scores are shared out) against the four launches
    per utterance (SEGK_FB_CHAIN=0) from identical states: boundaries, assignments, every statistic, K, the record values and
    the position of the RNG stream bit for bit, over sweeps in which components empty (ragged utterances, more components
    than the data support)."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    # (D = 520: beyond the 512 dimensions up to which the library tabulates the rows' log prior predictive -- the chain
    # evaluates it per utterance with fb_logits' own expression)
    corpus = make_corpus(60 if D < 100 else 24, D, K, seed=4, ragged=True, n_slices_max=5, N_range=(3, 14) if D < 100 else (3, 6))
    from segmentalist_amd import device as dev_mod
    ran = []
    real = dev_mod.DeviceFbgmm.sequential_sweep
    monkeypatch.setattr(dev_mod.DeviceFbgmm, "sequential_sweep", lambda self, *a, **k: ran.append(real(self, *a, **k)) or ran[-1])
    prior = (FixedVarPrior(0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D)) if cov == "fixed"
             else NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)))
    out = {}
    for mode in ("1", "terms0", "0"):
        # "terms0": the chain evaluating every component's predictive term again for each new segment instead of keeping the
        # spans' terms from the scoring phase (SEGK_FB_CHAIN_TERMS=0: also what runs when the kept rows do not fit in LDS)
        monkeypatch.setenv("SEGK_FB_CHAIN", "0" if mode == "0" else "1")
        monkeypatch.setenv("SEGK_FB_CHAIN_TERMS", "0" if mode == "terms0" else "1")
        random.seed(3)
        np.random.seed(3)
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, fb_type=fb_type,
                                         n_slices_min=0, n_slices_max=5, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0,
                                         wip=0.0, init_am_assignments="rand", time_power_term=1.0)
        n_before = len(ran)
        rec = seg.gibbs_sample(4, anneal_schedule="linear", anneal_gibbs_am=True)
        assert all(ran[n_before:]) == (mode != "0") and len(ran) > n_before, "the persistent kernel did not take the sweeps it should"
        df = seg._df
        out[mode] = dict(b=seg.utterances.boundaries.copy(), a=seg.acoustic_model.components.assignments.copy(),
                         sa=df.stat_a.cpu().numpy(), sb=df.stat_b.cpu().numpy(), pr=df.pred.cpu().numpy(),
                         lp=df.log_prod.cpu().numpy(), kc=df.kconst.cpu().numpy(), cn=df.counts.cpu().numpy(),
                         K=int(df.K.item()), rec={k: list(v) for k, v in rec.items() if k != "sample_time"}, rnd=random.random())
    if D < 100:
        assert min(out["1"]["rec"]["components"]) < K, "no component emptied: the test does not cover the relaunches"
    for other in ("0", "terms0"):
        for k in out["1"]:
            if isinstance(out["1"][k], np.ndarray):
                assert np.array_equal(out["1"][k], out[other][k]), (other, k)
            else:
                assert out["1"][k] == out[other][k], (other, k)


@pytest.mark.parametrize("sync", ["sequential", "batch"])
def test_banded_span_tables_feed_the_fbgmm_kernels(gpu, monkeypatch, sync):
    """The FBGMM segmentation kernels (k_unigram_segment, the persistent chain, k_fbb_segment) read the banded image of the
    span tables (segk_corpus.band_ids / band_dur) when it holds every embedding; against the same run on the
    triangular tables: boundaries, assignments, statistics and the RNG position bit for bit.  A corpus with embeddings
    outside the window keeps to the triangle."""
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    from segmentalist_amd.utterances import Utterances
    D, K = 12, 30
    corpus = make_corpus(60, D, K, seed=9, ragged=True, n_slices_max=5, N_range=(3, 14))
    prior = NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    args = dict(covariance_type="diag", fb_type="standard", n_slices_min=0, n_slices_max=5, p_boundary_init=0.5,
                beta_sent_boundary=-1, lms=1.0, wip=-0.1, init_am_assignments="rand", time_power_term=1.1)
    if sync == "batch":
        args.update(sync="batch", n_gibbs_blocks=3, n_stat_blocks=4, batch_seed=11)
    real = Utterances.complete_band_tables
    out = {}
    for mode in ("band", "triangle"):
        monkeypatch.setattr(Utterances, "complete_band_tables", real if mode == "band" else (lambda self, W: None))
        random.seed(3)
        np.random.seed(3)
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *corpus, **args)
        assert seg._corpus.band_W == (5 if mode == "band" else 0)
        rec = seg.gibbs_sample(3)
        df = seg._df
        out[mode] = dict(b=seg.utterances.boundaries.copy(), a=seg.acoustic_model.components.assignments.copy(),
                         sa=df.stat_a.cpu().numpy(), sb=df.stat_b.cpu().numpy(), cn=df.counts.cpu().numpy(), K=int(df.K.item()),
                         rec={k: list(v) for k, v in rec.items() if k != "sample_time"}, rnd=random.random())
    for k in out["band"]:
        if isinstance(out["band"][k], np.ndarray):
            assert np.array_equal(out["band"][k], out["triangle"][k]), k
        else:
            assert out["band"][k] == out["triangle"][k], k
    # embeddings for spans of up to seven slices, a window of five: the band would drop some -- the triangle stays
    monkeypatch.setattr(Utterances, "complete_band_tables", real)
    wide = make_corpus(20, D, K, seed=10, ragged=True, n_slices_max=7, N_range=(9, 14))
    random.seed(3)
    np.random.seed(3)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, prior, *wide, **args)
    assert seg._corpus.band_W == 0
    seg.gibbs_sample(1)


@pytest.mark.parametrize("n_range,nmax,cov,fb_type", [((3, 80), 6, "diag", "standard"), ((3, 80), 6, "fixed", "standard"),
                                                      ((3, 80), 6, "diag", "viterbi"), ((30, 80), 10, "fixed", "standard")])
def test_serial_chain_with_utterances_beyond_the_persistent_kernels_limit(gpu, n_range, nmax, cov, fb_type):
    """More than 64 landmarks per utterance (the persistent chain and the one-wave DP's fast forms stop applying) and a window
    of ten slices: the launches per utterance against oracle/np_oracle.py draw for draw -- boundaries, assignments and K equal,
    log_marg within 1e-12 relative -- over two sweeps from the same stream position."""
    from oracle import np_oracle as no
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.gaussian_components_fixedvar import FixedVarPrior
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K = 12, 20
    corpus = make_corpus(30, D, K, seed=4, ragged=True, n_slices_max=nmax, N_range=n_range)
    kw = dict(covariance_type=cov, fb_type=fb_type, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1,
              lms=1.0, wip=0.0, init_am_assignments="rand", time_power_term=1.0)
    pa = ((0.002 * np.ones(D), np.zeros(D), 0.002 / 0.05 * np.ones(D)) if cov == "fixed"
          else (np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)))
    random.seed(3); np.random.seed(3)
    ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, (no.FixedVarPrior if cov == "fixed" else no.NIW)(*pa), *corpus, **kw)
    random.seed(3); np.random.seed(3)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, (FixedVarPrior if cov == "fixed" else NIW)(*pa), *corpus, **kw)
    assert seg.utterances.boundaries.shape[1] > 64
    st = random.getstate()
    r0 = ref.gibbs_sample(2)
    random.setstate(st)
    r1 = seg.gibbs_sample(2)
    cr, cd = ref.acoustic_model.components, seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries)
    assert np.array_equal(cd.assignments, cr.assignments)
    assert cd.K == cr.K
    for a, b in zip(r0["log_marg"], r1["log_marg"]):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(a))


def test_a_second_model_at_the_first_ones_addresses_gets_its_own_tables(gpu, monkeypatch):
    """The persistent chain reads the rows' log prior predictive from a table the context caches.  The table is keyed by a
    fingerprint of the rows and the prior -- it was keyed by their ADDRESSES, and a model built after another was freed gets the
    same addresses from the caching allocator: a different corpus of the same shape then ran on the first corpus's table
    (boundaries, assignments and log_marg wrong, silently).  Model A through the chain, freed; model B (another corpus, same
    shape) through the chain must equal model B through the launches, which use no table."""
    import gc
    import torch
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K = 12, 30

    def run(cseed, chain, scale):
        monkeypatch.setenv("SEGK_FB_CHAIN", "1" if chain else "0")
        corpus = list(make_corpus(60, D, K, seed=cseed, N=10, n_slices_max=5))
        corpus[0] = {k: (v * scale).astype(np.float32) for k, v in corpus[0].items()}
        random.seed(3); np.random.seed(3)
        seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D)), *corpus,
                                         covariance_type="diag", fb_type="standard", n_slices_min=0, n_slices_max=5, p_boundary_init=0.5,
                                         beta_sent_boundary=-1, lms=1.0, wip=0.0, init_am_assignments="rand", time_power_term=1.0)
        rec = seg.gibbs_sample(3)
        out = (seg.utterances.boundaries.copy(), seg.acoustic_model.components.assignments.copy(), list(rec["log_marg"]))
        del seg
        gc.collect()
        torch.cuda.synchronize()
        return out

    run(4, True, 1.0)
    b_chain = run(5, True, 1.7)
    b_launch = run(5, False, 1.7)
    assert np.array_equal(b_chain[0], b_launch[0])
    assert np.array_equal(b_chain[1], b_launch[1])
    assert b_chain[2] == b_launch[2]


@pytest.mark.parametrize("n_range,nmax,fb_type", [((60, 80), 70, "standard"), ((60, 80), 70, "viterbi"), ((20, 40), 30, "standard"),
                                                  ((60, 80), 17, "standard")])
def test_serial_chain_with_wide_windows(gpu, n_range, nmax, fb_type):
    """Windows of 17, 30 and 70 slices on utterances of up to 80 landmarks (the DP's general form: more candidates per step than
    a wave has lanes) against the oracle draw for draw."""
    from oracle import np_oracle as no
    from segmentalist_amd import fbgmm, unigram_acoustic_wordseg as uaw
    from segmentalist_amd.niw import NIW
    from segmentalist_amd.synth import make_corpus
    D, K = 8, 10
    corpus = make_corpus(10, D, K, seed=4, ragged=True, n_slices_max=nmax, N_range=n_range)
    kw = dict(covariance_type="diag", fb_type=fb_type, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1,
              lms=1.0, wip=0.0, init_am_assignments="rand", time_power_term=1.0)
    pa = (np.zeros(D), 0.05, D + 3, 0.002 * (D + 3) * np.ones(D))
    random.seed(3); np.random.seed(3)
    ref = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, no.NIW(*pa), *corpus, **kw)
    random.seed(3); np.random.seed(3)
    seg = uaw.UnigramAcousticWordseg(fbgmm.FBGMM, 1.0, K, NIW(*pa), *corpus, **kw)
    st = random.getstate()
    r0 = ref.gibbs_sample(2)
    random.setstate(st)
    r1 = seg.gibbs_sample(2)
    assert np.array_equal(seg.utterances.boundaries, ref.utterances.boundaries)
    assert np.array_equal(seg.acoustic_model.components.assignments, ref.acoustic_model.components.assignments)
    for a, b in zip(r0["log_marg"], r1["log_marg"]):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(a))
