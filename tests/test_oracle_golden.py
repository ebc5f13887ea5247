"""
Pins the oracle (oracle/segk_oracle.c and oracle/np_oracle.py) to
  (a) golden vectors captured by running the reference (tests/golden/*.npz), and
  (b) the constants asserted in the reference's own tests
      (/root/reference/segmentalist/tests/*.py; file:line cited per test).
CPU only.
"""
import itertools
import random

import numpy as np
import numpy.testing as npt
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no
from tests.golden import cases


# ------------------------------------------------------------------ A1
@pytest.mark.parametrize("case", cases.A1_CASES, ids=[c[0] for c in cases.A1_CASES])
def test_a1_neg_sqrd_norm_bit_exact(golden, case):
    g = golden("kernels")
    name, D, K, n, dtype = case
    X, means = cases.a1_inputs(*case)
    want = g["a1_%s_scores" % name]
    assert want.dtype == np.dtype(dtype)
    for i in range(n):
        got = co.neg_sqrd_norm(means, X[i])
        assert got.dtype == want.dtype
        assert np.array_equal(got, want[i]), (name, i)
        # and numpy itself, evaluated here, agrees with the reference's stored numbers
        d = means - X[i]
        assert np.array_equal(-(d * d).sum(axis=1), want[i])
    mx, am = co.kmeans_max_argmax(means, X)
    assert np.array_equal(mx, g["a1_%s_max" % name].astype(np.float64))
    assert np.array_equal(am, g["a1_%s_argmax" % name])


def test_pairwise_sum_matches_numpy():
    rs = np.random.RandomState(3)
    for n in [0, 1, 3, 7, 8, 9, 39, 100, 127, 128, 129, 200, 257, 300, 1000, 4099]:
        for dt in (np.float32, np.float64):
            a = (rs.randn(n) * rs.rand() * 100).astype(dt)
            assert co.pairwise_sum(a) == a.sum(), (n, dt)


# ------------------------------------------------------------------ A9
def test_logsumexp_and_draw(golden):
    g = golden("kernels")
    off = 0
    for n, want in zip(g["lse_n"], g["lse_out"]):
        a = g["lse_in"][off:off + n]
        off += n
        npt.assert_allclose(co.logsumexp(a), want, rtol=1e-15, atol=0)
        npt.assert_allclose(no.logsumexp(a), want, rtol=1e-15, atol=0)
    for u, k in zip(g["draw_u"], g["draw_k"]):
        assert co.draw(g["draw_p"], u) == k
        assert no.draw(g["draw_p"], u) == k


# ------------------------------------------------------------------ A6 / A7 / A8
def test_dp_functions_match_reference(golden):
    g = golden("kernels")
    dpc = cases.dp_cases()
    ob = 0
    ou = 0
    n_backtrack = 0
    for ci, c in enumerate(dpc):
        vec, N, n_min, n_max = c["vec"], c["N"], c["n_min"], c["n_max"]
        # A8
        want_b = g["dp_km_bounds"][ob:ob + N]
        tot, b, _ = co.fb_kmeans_viterbi(vec, N, n_min, n_max)
        assert np.array_equal(b, want_b), ("km", ci)
        assert tot == g["dp_km_total"][ci] or (np.isnan(tot) and np.isnan(g["dp_km_total"][ci]))
        with np.errstate(all="ignore"):
            tot2, b2 = no.forward_backward_kmeans_viterbi(vec, N, n_min, n_max)
        assert np.array_equal(b2, want_b)
        assert tot2 == g["dp_km_total"][ci] or np.isnan(tot2)
        # A7
        want_b = g["dp_vt_bounds"][ob:ob + N]
        tot, b, _ = co.fb_viterbi(vec, N, n_min, n_max)
        assert np.array_equal(b, want_b), ("vt", ci)
        assert tot == g["dp_vt_total"][ci] or (np.isnan(tot) and np.isnan(g["dp_vt_total"][ci]))
        # A6, plain and annealed, with the uniforms the reference consumed
        for key, temp in (("fb", 1.0), ("fa", 1.7)):
            u = g["dp_%s_uniforms" % key][ou:ou + N + 1]
            nd = int(g["dp_%s_ndraws" % key][ci])
            want_b = g["dp_%s_bounds" % key][ob:ob + N]
            want_t = g["dp_%s_total" % key][ci]
            if np.isnan(want_t):            # reference asserted (log_prob == -inf)
                _, _, _, _, st = co.forward_backward(vec, -0.25, N, n_min, n_max, temp,
                                                     np.nan_to_num(u, nan=0.5))
                assert st == 1
                n_backtrack += 1
                continue
            tot, b, _, nd_got, st = co.forward_backward(vec, -0.25, N, n_min, n_max, temp,
                                                        np.nan_to_num(u, nan=0.5))
            assert st == 0 and nd_got == nd, (key, ci)
            assert np.array_equal(b, want_b), (key, ci)
            npt.assert_allclose(tot, want_t, rtol=1e-14)
            tot2, b2 = no.forward_backward(vec, -0.25, N, n_min, n_max, None, temp,
                                           uniforms=iter(u[:nd]))
            assert np.array_equal(b2, want_b)
        ob += N
        ou += N + 1
    assert n_backtrack > 0      # the dead-end cases were exercised


# ------------------------------------------------------------------ A2 / A3 / A4 / A10
@pytest.mark.parametrize("tag,D,K_max,n_items,seed", [("s", 5, 6, 40, 31), ("m", 39, 100, 600, 32),
                                                      ("l", 100, 40, 300, 33)])
def test_gaussian_components_match_reference(golden, tag, D, K_max, n_items, seed):
    g = golden("gauss")
    X, assign = cases.gauss_state(D, K_max, n_items, seed)
    idx = g["fx_%s_idx" % tag]
    # ---- fixed variance
    prior = no.FixedVarPrior(*cases.fixed_prior_params(D))
    fm = no.FBGMM(X, prior, 1.7, K_max, assign.copy(), covariance_type="fixed", lms=0.8)
    c = fm.components
    assert c.K == int(g["fx_%s_K" % tag])
    assert np.array_equal(c.counts, g["fx_%s_counts" % tag])
    for nm in ["mu_N_numerators", "precision_Ns", "log_prod_precision_preds", "precision_preds"]:
        assert np.array_equal(getattr(c, nm), g["fx_%s_%s" % (tag, nm)]), nm
    for j, i in enumerate(idx):
        npt.assert_allclose(c.log_post_pred(i), g["fx_%s_log_post_pred" % tag][j], rtol=1e-13)
        npt.assert_allclose(c.log_prior(i), g["fx_%s_log_prior" % tag][j], rtol=1e-13)
        npt.assert_allclose(fm.log_marg_i(i), g["fx_%s_log_marg_i" % tag][j], rtol=1e-13)
        # C twins
        lpp = co.fixedvar_log_post_pred(c.mu_N_numerators, c.precision_Ns, c.log_prod_precision_preds,
                                        c.precision_preds, c.K, X[i])
        npt.assert_allclose(lpp, g["fx_%s_log_post_pred" % tag][j], rtol=1e-13)
        lp = co.fixedvar_log_prior(c.mu_0, c.precision_0, X[i])
        npt.assert_allclose(lp, g["fx_%s_log_prior" % tag][j], rtol=1e-13)
        lm, _ = co.fbgmm_log_marg_i(c.counts, c.K, 1.7, 0.8, lpp, lp)
        npt.assert_allclose(lm, g["fx_%s_log_marg_i" % tag][j], rtol=1e-13)
    npt.assert_allclose(fm.log_marg(), g["fx_%s_log_marg" % tag], rtol=1e-12)
    npt.assert_allclose(fm.log_prob_z(), g["fx_%s_log_prob_z" % tag], rtol=1e-13)
    for j, i in enumerate(idx):
        k = fm.gibbs_sample_inside_loop_i(i, 1, u=g["fx_%s_sample_u" % tag][j])
        assert k == g["fx_%s_sample_k" % tag][j]
    assert np.array_equal(c.counts, g["fx_%s_after_counts" % tag])
    npt.assert_allclose(c.mu_N_numerators, g["fx_%s_after_mu_N_numerators" % tag], rtol=1e-15)
    # ---- diagonal covariance
    prior = no.NIW(*cases.diag_prior_params(D))
    fm = no.FBGMM(X, prior, 1.7, K_max, assign.copy(), covariance_type="diag", lms=0.8)
    c = fm.components
    assert c.K == int(g["dg_%s_K" % tag])
    for nm in ["m_N_numerators", "S_N_partials", "log_prod_vars", "inv_vars"]:
        npt.assert_allclose(getattr(c, nm), g["dg_%s_%s" % (tag, nm)], rtol=1e-14, err_msg=nm)
    for j, i in enumerate(idx):
        npt.assert_allclose(c.log_post_pred(i), g["dg_%s_log_post_pred" % tag][j], rtol=1e-13)
        npt.assert_allclose(c.log_prior(i), g["dg_%s_log_prior" % tag][j], rtol=1e-13)
        npt.assert_allclose(fm.log_marg_i(i), g["dg_%s_log_marg_i" % tag][j], rtol=1e-13)
        lpp = co.diag_log_post_pred(c.m_N_numerators, c.log_prod_vars, c.inv_vars, c.counts,
                                    prior.k_0, prior.v_0, c.K, X[i])
        # lgamma() evaluated directly instead of the reference's tables + cancellation near 0
        npt.assert_allclose(lpp, g["dg_%s_log_post_pred" % tag][j], rtol=1e-11, atol=1e-11)
        lp = co.diag_log_prior(prior.m_0, prior.k_0, prior.v_0, prior.S_0, X[i])
        npt.assert_allclose(lp, g["dg_%s_log_prior" % tag][j], rtol=1e-11, atol=1e-11)
    npt.assert_allclose(fm.log_marg(), g["dg_%s_log_marg" % tag], rtol=1e-12)
    for j, i in enumerate(idx):
        k = fm.gibbs_sample_inside_loop_i(i, 1, u=g["dg_%s_sample_u" % tag][j])
        assert k == g["dg_%s_sample_k" % tag][j]
    npt.assert_allclose(c.m_N_numerators, g["dg_%s_after_m_N_numerators" % tag], rtol=1e-15)
    npt.assert_allclose(c.S_N_partials, g["dg_%s_after_S_N_partials" % tag], rtol=1e-15)


# ------------------------------------------------------------------ chains: k-means (A12, bit exact)
@pytest.mark.parametrize("chain", cases.KMEANS_CHAINS, ids=[c[0] for c in cases.KMEANS_CHAINS])
@pytest.mark.parametrize("init", ["spread", "rand"])
def test_kmeans_wordseg_chain_matches_reference(golden, chain, init):
    g = golden("chains")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    no.set_shuffle("py3")
    seg = no.SegmentalKMeansWordseg(K, *corpus, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                    init_am_assignments=init, wip=0)
    c = seg.acoustic_model.components
    tag = "%s_%s" % (name, init)
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_init_bounds"])
    assert np.array_equal(c.assignments, g[tag + "_init_assign"])
    assert np.array_equal(c.random_means, g[tag + "_random_means"])
    assert c.means.dtype == np.dtype(dtype)
    for it in range(3):
        rec = seg.segment(1)
        assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[tag + "_assign"][it]), it
        assert np.array_equal(c.means, g[tag + "_means"][it]), it
        assert c.K == g[tag + "_K"][it]
        assert rec["sum_neg_len_sqrd_norm"][0] == g[tag + "_rec_sum_neg_len_sqrd_norm"][it]
        assert rec["sum_neg_sqrd_norm"][0] == g[tag + "_rec_sum_neg_sqrd_norm"][it]
        assert rec["n_tokens"][0] == g[tag + "_rec_n_tokens"][it]
    assert np.array_equal(c.mean_numerators, g[tag + "_mean_numerators"])
    assert np.array_equal(c.counts, g[tag + "_counts"])
    recf = seg.acoustic_model.fit(3, consider_unassigned=False)
    assert np.array_equal(c.assignments, g[tag + "_fit_assign"])
    assert np.array_equal(recf["n_mean_updates"], g[tag + "_fit_n_mean_updates"])
    assert np.array_equal(recf["sum_neg_sqrd_norm"], g[tag + "_fit_sum_neg_sqrd_norm"])


# ------------------------------------------------------------------ chains: unigram FBGMM
@pytest.mark.parametrize("chain", cases.UNIGRAM_CHAINS, ids=[c[0] for c in cases.UNIGRAM_CHAINS])
@pytest.mark.parametrize("fb_type", ["standard", "viterbi"])
def test_unigram_wordseg_chain_matches_reference(golden, chain, fb_type):
    g = golden("chains")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    no.set_shuffle("py3")
    prior = (no.FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed"
             else no.NIW(*cases.diag_prior_params(D)))
    seg = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov,
                                    n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                    beta_sent_boundary=-1, lms=1.0, wip=0.0, fb_type=fb_type,
                                    init_am_assignments="rand", time_power_term=1.0)
    c = seg.acoustic_model.components
    tag = "%s_%s" % (name, fb_type)
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_init_bounds"])
    assert np.array_equal(c.assignments, g[tag + "_init_assign"])
    for it in range(4):
        rec = seg.gibbs_sample(1)       # consumes the process-global `random` exactly like the reference
        assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[tag + "_assign"][it]), it
        for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z"]:
            npt.assert_allclose(rec[k][0], g[tag + "_rec_" + k][it], rtol=1e-10, err_msg=k)
        assert rec["components"][0] == g[tag + "_rec_components"][it]
        assert rec["n_tokens"][0] == g[tag + "_rec_n_tokens"][it]
    assert np.array_equal(c.counts, g[tag + "_counts"])


# ------------------------------------------------------------------ config 1: the notebook
def test_notebook_kmeans_trajectory(golden):
    """examples/clustering_examples.ipynb:272-280 -- published log lines."""
    g = golden("notebook")
    X = g["X"]
    np.random.set_state(("MT19937", g["np_state_keys"], int(g["np_state_pos"][0]),
                         int(g["np_state_pos"][1]), float(g["np_state_gauss"])))
    random.setstate((3, tuple(int(v) for v in g["py_state"]), None))
    no.set_shuffle("py2")
    try:
        km = no.KMeans(X, 4, "spread")
    finally:
        no.set_shuffle("py3")
    assert np.array_equal(km.components.assignments, g["kmeans_init_assign"])
    assert np.array_equal(km.components.random_means, g["kmeans_random_means"])
    rec = km.fit(20)
    published = [-618.585465615, -223.041596617, -220.219963349, -219.615938349, -207.450606173,
                 -126.321787187, -109.921387903, -108.302238117, -108.302238117]
    npt.assert_allclose(rec["sum_neg_sqrd_norm"], published, rtol=0, atol=5e-10)
    assert rec["n_mean_updates"] == [69, 18, 1, 1, 4, 11, 4, 1, 0]
    assert np.array_equal(rec["sum_neg_sqrd_norm"], g["kmeans_sum_neg_sqrd_norm"])
    assert np.array_equal(km.components.assignments, g["kmeans_final_assign"])
    assert np.array_equal(km.components.means, g["kmeans_final_means"])


# ------------------------------------------------------------------ constants from the reference's own tests
def _three_embedding_dataset():
    """tests/test_unigram_acoustic_wordseg.py:16-57 (fixture data)."""
    embedding_mat = np.array([
        [-0.2702691, -0.12348549, -0.20069546, -0.10067126, -0.32822475,
         -0.24878924, -0.17988801, -0.13201745, 0.66409844, -0.44816282],
        [-0.27186683, -0.12384345, -0.20049213, -0.10272419, -0.32618827,
         -0.24660945, -0.17784701, -0.13362537, 0.66524321, -0.44805479],
        [-0.2465426, -0.06354388, -0.22458388, 0.79060942, 0.48230717,
         -0.11888564, 0.06724239, -0.04977163, 0.06908087, 0.03395205]], dtype=np.float32)
    vec_ids = np.array([0, 1, 2])
    return ({"test": embedding_mat}, {"test": vec_ids}, {"test": [1, 2, 1]}, {"test": [1, 2]},
            {"test": [2]})


def _ref_prior(D):
    S_0 = 0.002 * np.ones(D)
    return no.FixedVarPrior(S_0, np.zeros(D), S_0 / 0.05)


def test_reference_test_simple_vec_embed_log_probs():
    """tests/test_unigram_acoustic_wordseg.py:60-90."""
    emb, vid, dur, lm, seeds = _three_embedding_dataset()
    random.seed(1)
    np.random.seed(1)
    seg = no.UnigramAcousticWordseg(no.FBGMM, 10., 2, _ref_prior(10), emb, vid, dur, lm,
                                    seed_boundaries_dict=seeds, beta_sent_boundary=-1)
    seg.gibbs_sample_i(0)
    got = seg.get_vec_embed_log_probs(seg.utterances.vec_ids[0], seg.utterances.durations[0])
    npt.assert_almost_equal(got, np.array([17.5548998, 35.103967, 17.5548998]))


def test_reference_test_simple_sampling():
    """tests/test_unigram_acoustic_wordseg.py:93-142."""
    emb, vid, dur, lm, seeds = _three_embedding_dataset()
    random.seed(1)
    np.random.seed(1)
    seg = no.UnigramAcousticWordseg(no.FBGMM, 10., 2, _ref_prior(10), emb, vid, dur, lm,
                                    seed_boundaries_dict=seeds, beta_sent_boundary=-1)
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


def test_reference_test_simple_sampling2():
    """tests/test_unigram_acoustic_wordseg.py:145-231 (two utterances, n_slices_max=2)."""
    m1 = np.array([[1.55329044, 0.82568932, 0.56011276], [1.10640768, -0.41715366, 0.30323529],
                   [1.24183824, -2.39021548, 0.02369367], [1.26094544, -0.27567053, 1.35731148],
                   [1.59711416, -0.54917262, -0.56074459], [-0.4298405, 1.39010761, -1.2608597]],
                  dtype=np.float32)
    m2 = np.array([[1.63075195, 0.25297823, -1.75406467], [-0.59324473, 0.96613426, -0.20922202],
                   [0.97066059, -1.22315308, -0.37979187], [-0.31613254, -0.07262261, -1.04392799],
                   [-1.11535652, 0.33905751, 1.85588856], [-1.08211738, 0.88559445, 0.2924617]],
                  dtype=np.float32)
    n = 3
    vec_ids = -1 * np.ones((n * n + n) // 2, dtype=int)
    e = 0
    for s in range(n):
        for t in range(s + 1, n + 1):
            vec_ids[t * (t - 1) // 2 + s] = e
            e += 1
    emb = {"test1": m1, "test2": m2}
    vid = {"test1": vec_ids, "test2": vec_ids}
    lm = {"test1": [1, 2, 3], "test2": [1, 2, 3]}
    dur = {"test1": [1, 2, 1, 3, 2, 1], "test2": [1, 2, 1, 3, 2, 1]}
    random.seed(1)
    np.random.seed(1)
    seg = no.UnigramAcousticWordseg(no.FBGMM, 10., 2, _ref_prior(3), emb, vid, dur, lm,
                                    p_boundary_init=0.5, beta_sent_boundary=-1, n_slices_max=2)
    rec = seg.gibbs_sample(3)
    npt.assert_almost_equal(rec["log_marg"], [-1520.885395538874, -435.84314783538349, -435.84314783538349])
    npt.assert_almost_equal(rec["log_prob_z"], [-3.641088790277589, -2.7937909298903829, -2.7937909298903829])
    npt.assert_almost_equal(rec["log_prob_X_given_z"],
                            [-1517.2443067485965, -433.04935690549308, -433.04935690549308])


def test_reference_test_kmeans_components():
    """tests/test_kmeans_components.py:13-79 (closed-form cross-checks)."""
    np.random.seed(1)
    D, N, K_true = 4, 11, 4
    z_true = np.random.randint(0, K_true, N)
    mu = np.random.randn(D, K_true) * 4.0
    X = (mu[:, z_true] + np.random.randn(D, N) * 0.7).T
    assignments = no.consecutive_labels(np.random.randint(0, 5, N))
    c = no.KMeansComponents(X, assignments, 5)
    for i in range(N):
        want = [-np.linalg.norm(X[i] - c.mean_numerators[k] / c.counts[k]) ** 2 for k in range(c.K)]
        npt.assert_almost_equal(c.neg_sqrd_norm(i)[:c.K], want)
        npt.assert_almost_equal(co.neg_sqrd_norm(c.means, X[i])[:c.K], want)
    for k in range(c.K):
        npt.assert_almost_equal(np.mean(X[c.assignments == k], axis=0), c.mean_numerators[k] / c.counts[k])


def test_reference_test_fixedvar_log_post_pred_closed_form():
    """tests/test_gaussian_components_fixedvar.py:36-86 style: vectorised == product of normal pdfs."""
    rs = np.random.RandomState(1)
    D, N = 3, 12
    X = rs.randn(N, D).astype(np.float32)
    var = 0.5 * np.ones(D)
    prior = no.FixedVarPrior(var, rs.randn(D), 2.0 * np.ones(D))
    assign = np.array([0, 0, 1, 1, 1, 2, 2, -1, -1, 0, 1, 2])
    c = no.GaussianComponentsFixedVar(X, prior, assign.copy(), K_max=4)
    i = 7
    for k in range(c.K):
        Xk = X[assign == k].astype(np.float64)
        n = len(Xk)
        prec_N = 1. / prior.var_0 + n / var
        mu_N = (prior.mu_0 / prior.var_0 + Xk.sum(axis=0) / var) / prec_N
        var_pred = 1. / prec_N + var
        want = np.sum(-0.5 * (np.log(2 * np.pi) + np.log(var_pred)) - (X[i] - mu_N) ** 2 / (2 * var_pred))
        npt.assert_almost_equal(c.log_post_pred(i)[k], want)


def test_batch_sweep_equals_sequential_for_single_utterance(golden):
    """The batch-synchronous spec degenerates to the reference's segment_i when the batch is
    one utterance and statistics are rebuilt exactly (1 block)."""
    corpus = cases.chain_corpus(1, 4, 3, 77, True, 0, 4, "float32")
    for seed in range(3):
        random.seed(seed)
        np.random.seed(seed)
        a = no.SegmentalKMeansWordseg(3, *corpus, n_slices_max=4, init_am_assignments="rand")
        random.seed(seed)
        np.random.seed(seed)
        b = no.SegmentalKMeansWordseg(3, *corpus, n_slices_max=4, init_am_assignments="rand")
        ta = a.segment_i(0)
        tb = no.kmeans_batch_sweep(b, n_blocks=1)
        assert ta == tb
        assert np.array_equal(a.utterances.boundaries, b.utterances.boundaries)
        assert np.array_equal(a.acoustic_model.components.assignments,
                              b.acoustic_model.components.assignments)
        assert a.acoustic_model.components.K == b.acoustic_model.components.K


# ------------------------------------------------------------------ chains: bigram driver (config 5)
@pytest.mark.parametrize("chain", cases.BIGRAM_CHAINS, ids=[c[0] for c in cases.BIGRAM_CHAINS])
def test_bigram_wordseg_chain_matches_reference(golden, chain):
    g = golden("bigram")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    no.set_shuffle("py3")
    prior = no.FixedVarPrior(*cases.fixed_prior_params(D))
    seg = no.BigramAcousticWordseg(K, prior, dict(cases.BIGRAM_LM), *corpus, covariance_type=cov,
                                   n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                                   beta_sent_boundary=-1, lms=1.0, wip=0.0, fb_type="unigram",
                                   init_am_assignments="rand", time_power_term=1.0)
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
            npt.assert_allclose(rec[k][0], g[name + "_rec_" + k][it], rtol=1e-10, err_msg=k)
        assert rec["components"][0] == g[name + "_rec_components"][it]


def test_reference_test_bigram_lms():
    """tests/test_bigram_lms.py:13-76 of the reference (closed-form checks)."""
    lm = no.BigramSmoothLM(0.1, 1, 2, 5)
    for utt in [[1, 1, 3, 4, 0], [4, 4], [1, 0, 2, 2, 2, 2, 3, 1], [3, 3, 1]]:
        lm.counts_from_utterance(utt)
    npt.assert_almost_equal(lm.prob_i_given_j(1, 3), 0.1 * lm.prob_i(1) + 0.9 * (2. + 2. / 5) / (4 + 2))
    npt.assert_almost_equal(lm.prob_i(1), (5. + 1. / 5) / (18 + 1))
    pv = lm.prob_vec_i()
    pj = lm.prob_vec_given_j(3)
    for i in range(5):
        assert pv[i] == lm.prob_i(i)
        npt.assert_almost_equal(pj[i], lm.prob_i_given_j(i, 3))
        npt.assert_almost_equal(lm.log_prob_vec_i()[i], np.log(lm.prob_i(i)))
    lm.remove_counts_from_utterance([3, 3, 1])
    assert lm.unigram_counts.sum() == 15 and lm.bigram_counts[3, 3] == 0


# ------------------------------------------------------------------ FBGMM.gibbs_sample
def _amg_kw(sched):
    if sched == "linear":
        return dict(anneal_schedule="linear", anneal_start_temp_inv=0.5, anneal_end_temp_inv=1.0)
    if sched == "step":
        return dict(anneal_schedule="step", anneal_start_temp_inv=0.25, anneal_end_temp_inv=1.0, n_anneal_steps=2)
    return {}


@pytest.mark.parametrize("case", cases.AM_GIBBS, ids=[c[0] for c in cases.AM_GIBBS])
def test_fbgmm_gibbs_sample_matches_reference(golden, case):
    g = golden("amgibbs")
    name, D, K_max, n_items, seed, cov, unassigned, sched = case
    X, assign = cases.gauss_state(D, K_max, n_items, seed)
    prior = no.FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else no.NIW(*cases.diag_prior_params(D))
    random.seed(3)
    np.random.seed(3)
    fm = no.FBGMM(X, prior, 1.0, K_max, assign.copy(), covariance_type=cov, lms=1.0)
    rec = fm.gibbs_sample(4, consider_unassigned=unassigned, **_amg_kw(sched))
    assert np.array_equal(fm.components.assignments, g[name + "_assign"])
    assert np.array_equal(fm.components.counts, g[name + "_counts"])
    for k in ["log_marg", "log_prob_z", "log_prob_X_given_z", "anneal_temp"]:
        npt.assert_allclose(rec[k], g[name + "_rec_" + k], rtol=1e-10, err_msg=k)
    assert list(rec["components"]) == list(g[name + "_rec_components"])
    rec = fm.gibbs_sample(1, consider_unassigned=unassigned)
    assert np.array_equal(fm.components.assignments, g[name + "_assign_5"])
    npt.assert_allclose(rec["log_marg"], g[name + "_log_marg_5"], rtol=1e-10)


@pytest.mark.parametrize("chain", cases.AM_ITER_CHAINS, ids=[c[0] for c in cases.AM_ITER_CHAINS])
def test_unigram_chain_with_am_iterations_matches_reference(golden, chain):
    g = golden("amgibbs")
    name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov = chain
    corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
    random.seed(1)
    np.random.seed(1)
    no.set_shuffle("py3")
    prior = no.FixedVarPrior(*cases.fixed_prior_params(D)) if cov == "fixed" else no.NIW(*cases.diag_prior_params(D))
    seg = no.UnigramAcousticWordseg(no.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, n_slices_min=0,
                                    n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0,
                                    wip=0.0, fb_type="standard", init_am_assignments="rand", time_power_term=1.0)
    rec = seg.gibbs_sample(3, am_n_iter=2)
    tag = name + "_amiter"
    assert np.array_equal(seg.utterances.boundaries, g[tag + "_bounds"])
    assert np.array_equal(seg.acoustic_model.components.assignments, g[tag + "_assign"])
    npt.assert_allclose(rec["log_marg"], g[tag + "_rec_log_marg"], rtol=1e-10)
    assert list(rec["components"]) == list(g[tag + "_rec_components"])


# ------------------------------------------------------------------ edge cases of the drivers
def _edge_mods(ns):
    return dict(SegmentalKMeansWordseg=ns.SegmentalKMeansWordseg, UnigramAcousticWordseg=ns.UnigramAcousticWordseg,
                BigramAcousticWordseg=ns.BigramAcousticWordseg, FBGMM=ns.FBGMM, FixedVarPrior=ns.FixedVarPrior,
                NIW=ns.NIW)


@pytest.mark.parametrize("case", cases.EDGE_CHAINS, ids=[c[0] for c in cases.EDGE_CHAINS])
def test_edge_case_chains_match_reference(golden, case):
    """1- and 2-landmark utterances, NaN durations (min_duration), n_slices_min = 1, a single initial
    span, snapped seed boundaries: trajectories of the reference itself."""
    g = golden("edge")
    name, driver = case[0], case[1]
    random.seed(1)
    np.random.seed(1)
    no.set_shuffle("py3")
    seg = cases.edge_build(_edge_mods(no), case)
    c = seg.acoustic_model.components
    assert np.array_equal(seg.utterances.boundaries, g[name + "_init_bounds"])
    assert np.array_equal(c.assignments, g[name + "_init_assign"])
    assert np.array_equal(np.isnan(seg.utterances.durations), g[name + "_durations_nan"])
    for it in range(3):
        rec = seg.segment(1) if driver == "kmeans" else seg.gibbs_sample(1)
        assert np.array_equal(seg.utterances.boundaries, g[name + "_bounds"][it]), it
        assert np.array_equal(c.assignments, g[name + "_assign"][it]), it
        key = "sum_neg_len_sqrd_norm" if driver == "kmeans" else "log_marg"
        npt.assert_allclose(rec[key][0], g[name + "_rec_" + key][it], rtol=1e-10)
        assert rec["components"][0] == g[name + "_rec_components"][it]
