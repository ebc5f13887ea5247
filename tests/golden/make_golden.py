#!/usr/bin/env python3
"""
Generate golden vectors by RUNNING THE REFERENCE (a throw-away py3 translation built by
build_ref.py in a scratch directory outside this repository).  Only data -- inputs that
cannot be re-derived from a seed, and the reference's outputs -- is written, as .npz
files next to this script.  Runnable only where /root/reference exists.

usage: python tests/golden/build_ref.py /tmp/segk_ref && python tests/golden/make_golden.py /tmp/segk_ref
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/segk_ref"
sys.path.insert(0, scratch)

from segmentalist import (bigram_acoustic_wordseg, fbgmm, gaussian_components_diag, gaussian_components_fixedvar,  # noqa: E402
                          kmeans, kmeans_acoustic_wordseg, kmeans_components, niw,
                          unigram_acoustic_wordseg, _cython_utils)

unigram_acoustic_wordseg.i_debug_monitor = -1
bigram_acoustic_wordseg.i_debug_monitor = -1
kmeans_acoustic_wordseg.i_debug_monitor = -1


class UniformLog(object):
    """Records every random.random() the reference consumes."""

    def __enter__(self):
        self.log = []
        self._orig = random.random

        def rec():
            u = self._orig()
            self.log.append(u)
            return u
        random.random = rec
        return self

    def __exit__(self, *a):
        random.random = self._orig


def shuffle_py2(x):
    for i in reversed(range(1, len(x))):
        j = int(random.random() * (i + 1))
        x[i], x[j] = x[j], x[i]


# ----------------------------------------------------------------------------- kernels
def gen_kernels():
    out = {}
    # A1 neg_sqrd_norm
    for name, D, K, n, dtype in cases.A1_CASES:
        X, means = cases.a1_inputs(name, D, K, n, dtype)
        np.random.seed(0)
        c = kmeans_components.KMeansComponents(X, np.zeros(n, dtype=int), K)
        c.means = means.copy()
        S = np.stack([c.neg_sqrd_norm(i) for i in range(n)])
        assert S.dtype == np.dtype(dtype), (S.dtype, dtype)
        out["a1_%s_scores" % name] = S
        out["a1_%s_max" % name] = np.array([c.max_neg_sqrd_norm_i(i) for i in range(n)])
        out["a1_%s_argmax" % name] = np.array([c.argmax_neg_sqrd_norm_i(i) for i in range(n)])
    # A9 logsumexp
    rs = np.random.RandomState(5)
    lse_in = [rs.randn(m) * s for m, s in [(1, 1), (2, 10), (6, 50), (100, 5), (1000, 300)]]
    lse_in.append(np.array([-np.inf, -3.0, -np.inf]))
    out["lse_n"] = np.array([len(a) for a in lse_in])
    out["lse_in"] = np.concatenate(lse_in)
    out["lse_out"] = np.array([_cython_utils.logsumexp(np.ascontiguousarray(a)) for a in lse_in])
    # A9 draw
    p = rs.dirichlet(np.ones(7))
    us = np.concatenate([rs.rand(20), [0.0, 0.999999999999]])
    ks = []
    for u in us:
        orig = random.random
        random.random = lambda: u
        ks.append(_cython_utils.draw(p))
        random.random = orig
    out["draw_p"], out["draw_u"], out["draw_k"] = p, us, np.array(ks)
    # DP functions A6/A7/A8
    dpc = cases.dp_cases()
    km_tot, km_b, vt_tot, vt_b, fb_tot, fb_b, fb_u, fb_nd, fb_ok = [], [], [], [], [], [], [], [], []
    fa_tot, fa_b, fa_u, fa_nd = [], [], [], []
    random.seed(99)
    for c in dpc:
        vec, N, n_min, n_max = c["vec"], c["N"], c["n_min"], c["n_max"]
        with np.errstate(all="ignore"):
            t, b = kmeans_acoustic_wordseg.forward_backward_kmeans_viterbi(vec.copy(), N, n_min, n_max, None)
        km_tot.append(t)
        km_b.append(b)
        with np.errstate(all="ignore"):
            t, b = unigram_acoustic_wordseg.forward_backward_viterbi(vec.copy(), 0.0, N, n_min, n_max, None)
        vt_tot.append(t)
        vt_b.append(b)
        for temp, (T, B, U, ND) in [(1, (fb_tot, fb_b, fb_u, fb_nd)), (1.7, (fa_tot, fa_b, fa_u, fa_nd))]:
            with UniformLog() as ul:
                try:
                    with np.errstate(all="ignore"):
                        t, b = unigram_acoustic_wordseg.forward_backward(
                            vec.copy(), -0.25, N, n_min, n_max, None, temp)
                    ok = True
                except AssertionError:
                    t, b, ok = np.nan, np.zeros(N, bool), False
            if temp == 1:
                fb_ok.append(ok)
            T.append(t)
            B.append(b)
            u = np.full(N + 1, np.nan)
            u[:len(ul.log)] = ul.log
            U.append(u)
            ND.append(len(ul.log))
    out["dp_km_total"] = np.array(km_tot)
    out["dp_km_bounds"] = np.concatenate(km_b)
    out["dp_vt_total"] = np.array(vt_tot)
    out["dp_vt_bounds"] = np.concatenate(vt_b)
    out["dp_fb_total"] = np.array(fb_tot)
    out["dp_fb_bounds"] = np.concatenate(fb_b)
    out["dp_fb_uniforms"] = np.concatenate(fb_u)
    out["dp_fb_ndraws"] = np.array(fb_nd)
    out["dp_fb_ok"] = np.array(fb_ok)
    out["dp_fa_total"] = np.array(fa_tot)
    out["dp_fa_bounds"] = np.concatenate(fa_b)
    out["dp_fa_uniforms"] = np.concatenate(fa_u)
    out["dp_fa_ndraws"] = np.array(fa_nd)
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)
    print("kernels.npz:", len(out), "arrays,", len(dpc), "DP cases")


# ----------------------------------------------------------------------------- gaussians
def gen_gauss():
    out = {}
    for tag, D, K_max, n_items, seed in [("s", 5, 6, 40, 31), ("m", 39, 100, 600, 32), ("l", 100, 40, 300, 33)]:
        X, assign = cases.gauss_state(D, K_max, n_items, seed)
        # fixed variance
        var, mu_0, var_0 = cases.fixed_prior_params(D)
        prior = gaussian_components_fixedvar.FixedVarPrior(var, mu_0, var_0)
        fm = fbgmm.FBGMM(X, prior, 1.7, K_max, assign.copy(), covariance_type="fixed", lms=0.8)
        c = fm.components
        idx = np.where(c.assignments == -1)[0][:6]
        out["fx_%s_idx" % tag] = idx
        out["fx_%s_K" % tag] = np.array(c.K)
        out["fx_%s_counts" % tag] = c.counts.copy()
        out["fx_%s_mu_N_numerators" % tag] = c.mu_N_numerators.copy()
        out["fx_%s_precision_Ns" % tag] = c.precision_Ns.copy()
        out["fx_%s_log_prod_precision_preds" % tag] = c.log_prod_precision_preds.copy()
        out["fx_%s_precision_preds" % tag] = c.precision_preds.copy()
        out["fx_%s_log_post_pred" % tag] = np.stack([c.log_post_pred(i) for i in idx])
        out["fx_%s_log_prior" % tag] = np.array([c.log_prior(i) for i in idx])
        out["fx_%s_log_marg_i" % tag] = np.array([fm.log_marg_i(i) for i in idx])
        out["fx_%s_log_marg" % tag] = np.array(fm.log_marg())
        out["fx_%s_log_prob_z" % tag] = np.array(fm.log_prob_z())
        # assignment sampling A10 (state mutates: record k and u per step)
        ks, us = [], []
        for i in idx:
            with UniformLog() as ul:
                fm.gibbs_sample_inside_loop_i(i)
            ks.append(c.assignments[i])
            us.append(ul.log[0])
        out["fx_%s_sample_k" % tag] = np.array(ks)
        out["fx_%s_sample_u" % tag] = np.array(us)
        out["fx_%s_after_mu_N_numerators" % tag] = c.mu_N_numerators.copy()
        out["fx_%s_after_counts" % tag] = c.counts.copy()
        # diagonal covariance
        m_0, k_0, v_0, S_0 = cases.diag_prior_params(D)
        prior = niw.NIW(m_0, k_0, v_0, S_0)
        fm = fbgmm.FBGMM(X, prior, 1.7, K_max, assign.copy(), covariance_type="diag", lms=0.8)
        c = fm.components
        out["dg_%s_K" % tag] = np.array(c.K)
        out["dg_%s_counts" % tag] = c.counts.copy()
        out["dg_%s_m_N_numerators" % tag] = c.m_N_numerators.copy()
        out["dg_%s_S_N_partials" % tag] = c.S_N_partials.copy()
        out["dg_%s_log_prod_vars" % tag] = c.log_prod_vars.copy()
        out["dg_%s_inv_vars" % tag] = c.inv_vars.copy()
        out["dg_%s_log_post_pred" % tag] = np.stack([c.log_post_pred(i) for i in idx])
        out["dg_%s_log_prior" % tag] = np.array([c.log_prior(i) for i in idx])
        out["dg_%s_log_marg_i" % tag] = np.array([fm.log_marg_i(i) for i in idx])
        out["dg_%s_log_marg" % tag] = np.array(fm.log_marg())
        ks, us = [], []
        for i in idx:
            with UniformLog() as ul:
                fm.gibbs_sample_inside_loop_i(i)
            ks.append(c.assignments[i])
            us.append(ul.log[0])
        out["dg_%s_sample_k" % tag] = np.array(ks)
        out["dg_%s_sample_u" % tag] = np.array(us)
        out["dg_%s_after_m_N_numerators" % tag] = c.m_N_numerators.copy()
        out["dg_%s_after_S_N_partials" % tag] = c.S_N_partials.copy()
    np.savez_compressed(os.path.join(HERE, "gauss.npz"), **out)
    print("gauss.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- chains
def gen_chains():
    out = {}
    for name, n_utt, D, K, seed, ragged, N, nmax, dtype in cases.KMEANS_CHAINS:
        corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
        for init in ["spread", "rand"]:
            random.seed(1)
            np.random.seed(1)
            seg = kmeans_acoustic_wordseg.SegmentalKMeansWordseg(
                K, *corpus, n_slices_min=0, n_slices_max=nmax, p_boundary_init=0.5,
                init_am_assignments=init, wip=0)
            c = seg.acoustic_model.components
            tag = "%s_%s" % (name, init)
            out[tag + "_init_bounds"] = seg.utterances.boundaries.copy()
            out[tag + "_init_assign"] = c.assignments.copy()
            out[tag + "_random_means"] = c.random_means.copy()
            bounds, assigns, means, Ks = [], [], [], []
            rec_all = {}
            for it in range(3):
                rec = seg.segment(1)
                for k, v in rec.items():
                    rec_all.setdefault(k, []).extend(v)
                bounds.append(seg.utterances.boundaries.copy())
                assigns.append(c.assignments.copy())
                means.append(c.means.copy())
                Ks.append(c.K)
            out[tag + "_bounds"] = np.stack(bounds)
            out[tag + "_assign"] = np.stack(assigns)
            out[tag + "_means"] = np.stack(means)
            out[tag + "_mean_numerators"] = c.mean_numerators.copy()
            out[tag + "_counts"] = c.counts.copy()
            out[tag + "_K"] = np.array(Ks)
            for k in ["sum_neg_sqrd_norm", "sum_neg_len_sqrd_norm", "components", "n_tokens"]:
                out[tag + "_rec_" + k] = np.array(rec_all[k])
            # KMeans.fit refinement on the final state (A-next, kmeans.py:97)
            recf = seg.acoustic_model.fit(3, consider_unassigned=False)
            out[tag + "_fit_assign"] = c.assignments.copy()
            out[tag + "_fit_sum_neg_sqrd_norm"] = np.array(recf["sum_neg_sqrd_norm"])
            out[tag + "_fit_n_mean_updates"] = np.array(recf["n_mean_updates"])

    for name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov in cases.UNIGRAM_CHAINS:
        corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
        for fb_type in ["standard", "viterbi"]:
            random.seed(1)
            np.random.seed(1)
            if cov == "fixed":
                prior = gaussian_components_fixedvar.FixedVarPrior(*cases.fixed_prior_params(D))
            else:
                prior = niw.NIW(*cases.diag_prior_params(D))
            seg = unigram_acoustic_wordseg.UnigramAcousticWordseg(
                fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, n_slices_min=0,
                n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
                fb_type=fb_type, init_am_assignments="rand", time_power_term=1.0)
            c = seg.acoustic_model.components
            tag = "%s_%s" % (name, fb_type)
            out[tag + "_init_bounds"] = seg.utterances.boundaries.copy()
            out[tag + "_init_assign"] = c.assignments.copy()
            bounds, assigns, nus = [], [], []
            rec_all = {}
            ulog_all = []
            for it in range(4):
                with UniformLog() as ul:
                    rec = seg.gibbs_sample(1)
                ulog_all.extend(ul.log)
                nus.append(len(ul.log))
                for k, v in rec.items():
                    rec_all.setdefault(k, []).extend(v)
                bounds.append(seg.utterances.boundaries.copy())
                assigns.append(c.assignments.copy())
            out[tag + "_bounds"] = np.stack(bounds)
            out[tag + "_assign"] = np.stack(assigns)
            out[tag + "_n_uniforms"] = np.array(nus)
            out[tag + "_uniforms"] = np.array(ulog_all)
            out[tag + "_counts"] = c.counts.copy()
            for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z", "components",
                      "n_tokens"]:
                out[tag + "_rec_" + k] = np.array(rec_all[k])
    np.savez_compressed(os.path.join(HERE, "chains.npz"), **out)
    print("chains.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- bigram chains (config 5)
def gen_bigram():
    out = {}
    for name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov in cases.BIGRAM_CHAINS:
        corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
        random.seed(1)
        np.random.seed(1)
        if cov == "fixed":
            prior = gaussian_components_fixedvar.FixedVarPrior(*cases.fixed_prior_params(D))
        else:
            prior = niw.NIW(*cases.diag_prior_params(D))
        seg = bigram_acoustic_wordseg.BigramAcousticWordseg(
            K, prior, dict(cases.BIGRAM_LM), *corpus, covariance_type=cov, n_slices_min=0, n_slices_max=nmax,
            p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0, fb_type="unigram",
            init_am_assignments="rand", time_power_term=1.0)
        c = seg.acoustic_model.components
        out[name + "_init_bounds"] = seg.utterances.boundaries.copy()
        out[name + "_init_assign"] = c.assignments.copy()
        out[name + "_init_unigram"] = seg.lm.unigram_counts.copy()
        out[name + "_init_bigram"] = seg.lm.bigram_counts.copy()
        bounds, assigns, unis, bis = [], [], [], []
        rec_all = {}
        for it in range(4):
            rec = seg.gibbs_sample(1)
            for k, v in rec.items():
                rec_all.setdefault(k, []).extend(v)
            bounds.append(seg.utterances.boundaries.copy())
            assigns.append(c.assignments.copy())
            unis.append(seg.lm.unigram_counts.copy())
            bis.append(seg.lm.bigram_counts.copy())
        out[name + "_bounds"] = np.stack(bounds)
        out[name + "_assign"] = np.stack(assigns)
        out[name + "_unigram"] = np.stack(unis)
        out[name + "_bigram"] = np.stack(bis)
        for k in ["log_marg", "log_marg*length", "log_prob_z", "log_prob_X_given_z", "components", "n_tokens"]:
            out[name + "_rec_" + k] = np.array(rec_all[k])
    np.savez_compressed(os.path.join(HERE, "bigram.npz"), **out)
    print("bigram.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- FBGMM.gibbs_sample (8(f).1)
def gen_amgibbs():
    out = {}
    for name, D, K_max, n_items, seed, cov, unassigned, sched in cases.AM_GIBBS:
        X, assign = cases.gauss_state(D, K_max, n_items, seed)
        if cov == "fixed":
            prior = gaussian_components_fixedvar.FixedVarPrior(*cases.fixed_prior_params(D))
        else:
            prior = niw.NIW(*cases.diag_prior_params(D))
        random.seed(3)
        np.random.seed(3)
        fm = fbgmm.FBGMM(X, prior, 1.0, K_max, assign.copy(), covariance_type=cov, lms=1.0)
        assigns, Ks, nus = [], [], []
        rec_all = {}
        kw = {}
        if sched == "linear":
            kw = dict(anneal_schedule="linear", anneal_start_temp_inv=0.5, anneal_end_temp_inv=1.0)
        elif sched == "step":
            kw = dict(anneal_schedule="step", anneal_start_temp_inv=0.25, anneal_end_temp_inv=1.0, n_anneal_steps=2)
        with UniformLog() as ul:
            rec = fm.gibbs_sample(4, consider_unassigned=unassigned, **kw)
        out[name + "_n_uniforms"] = np.array(len(ul.log))
        out[name + "_assign"] = fm.components.assignments.copy()
        out[name + "_counts"] = fm.components.counts.copy()
        for k in ["log_marg", "log_prob_z", "log_prob_X_given_z", "anneal_temp", "components"]:
            out[name + "_rec_" + k] = np.array(rec[k])
        # one more sweep, one iteration at a time, to pin the per-iteration state
        rec = fm.gibbs_sample(1, consider_unassigned=unassigned)
        out[name + "_assign_5"] = fm.components.assignments.copy()
        out[name + "_log_marg_5"] = np.array(rec["log_marg"])
    for name, n_utt, D, K, seed, ragged, N, nmax, dtype, cov in cases.AM_ITER_CHAINS:
        corpus = cases.chain_corpus(n_utt, D, K, seed, ragged, N, nmax, dtype)
        random.seed(1)
        np.random.seed(1)
        if cov == "fixed":
            prior = gaussian_components_fixedvar.FixedVarPrior(*cases.fixed_prior_params(D))
        else:
            prior = niw.NIW(*cases.diag_prior_params(D))
        seg = unigram_acoustic_wordseg.UnigramAcousticWordseg(
            fbgmm.FBGMM, 1.0, K, prior, *corpus, covariance_type=cov, n_slices_min=0,
            n_slices_max=nmax, p_boundary_init=0.5, beta_sent_boundary=-1, lms=1.0, wip=0.0,
            fb_type="standard", init_am_assignments="rand", time_power_term=1.0)
        rec = seg.gibbs_sample(3, am_n_iter=2)
        tag = name + "_amiter"
        out[tag + "_bounds"] = seg.utterances.boundaries.copy()
        out[tag + "_assign"] = seg.acoustic_model.components.assignments.copy()
        for k in ["log_marg", "log_marg*length", "components", "n_tokens"]:
            out[tag + "_rec_" + k] = np.array(rec[k])
    np.savez_compressed(os.path.join(HERE, "amgibbs.npz"), **out)
    print("amgibbs.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- edge cases of the drivers
def gen_edge():
    mods = dict(SegmentalKMeansWordseg=kmeans_acoustic_wordseg.SegmentalKMeansWordseg,
                UnigramAcousticWordseg=unigram_acoustic_wordseg.UnigramAcousticWordseg,
                BigramAcousticWordseg=bigram_acoustic_wordseg.BigramAcousticWordseg, FBGMM=fbgmm.FBGMM,
                FixedVarPrior=gaussian_components_fixedvar.FixedVarPrior, NIW=niw.NIW)
    out = {}
    for case in cases.EDGE_CHAINS:
        name, driver = case[0], case[1]
        random.seed(1)
        np.random.seed(1)
        seg = cases.edge_build(mods, case)
        c = seg.acoustic_model.components
        out[name + "_init_bounds"] = seg.utterances.boundaries.copy()
        out[name + "_init_assign"] = c.assignments.copy()
        out[name + "_durations_nan"] = np.isnan(seg.utterances.durations)
        bounds, assigns = [], []
        recs = {}
        for it in range(3):
            rec = seg.segment(1) if driver == "kmeans" else seg.gibbs_sample(1)
            for k, v in rec.items():
                recs.setdefault(k, []).extend(v)
            bounds.append(seg.utterances.boundaries.copy())
            assigns.append(c.assignments.copy())
        out[name + "_bounds"] = np.stack(bounds)
        out[name + "_assign"] = np.stack(assigns)
        for k in (["sum_neg_len_sqrd_norm", "components", "n_tokens"] if driver == "kmeans"
                  else ["log_marg", "log_marg*length", "components", "n_tokens"]):
            out[name + "_rec_" + k] = np.array(recs[k])
    np.savez_compressed(os.path.join(HERE, "edge.npz"), **out)
    print("edge.npz:", len(out), "arrays")


# ----------------------------------------------------------------------------- notebook (config 1)
def gen_notebook():
    """examples/clustering_examples.ipynb cells, with the python-2 shuffle."""
    out = {}
    random.seed(2)
    np.random.seed(2)
    D, N, K_true = 2, 100, 4
    mu_scale, covar_scale = 4.0, 0.7
    z_true = np.random.randint(0, K_true, N)
    mu = np.random.randn(D, K_true) * mu_scale
    X = (mu[:, z_true] + np.random.randn(D, N) * covar_scale).T
    out["X"] = X.copy()
    alpha, K, n_iter = 1., 4, 20
    var_scale = 0.5
    mu_0 = np.zeros(D)
    k_0 = covar_scale ** 2 / mu_scale ** 2
    var = covar_scale ** 2 * np.ones(D) * var_scale
    var_0 = var / k_0
    prior = gaussian_components_fixedvar.FixedVarPrior(var, mu_0, var_0)
    fm = fbgmm.FBGMM(X, prior, alpha, K, "rand", covariance_type="fixed")
    out["fbgmm_init_assign"] = fm.components.assignments.copy()
    with UniformLog() as ul:
        rec = fm.gibbs_sample(n_iter)
    out["fbgmm_uniforms"] = np.array(ul.log)
    out["fbgmm_log_marg"] = np.array(rec["log_marg"])
    out["fbgmm_final_assign"] = fm.components.assignments.copy()
    # state of both RNG streams just before the k-means cell
    st = np.random.get_state()
    out["np_state_keys"] = st[1].copy()
    out["np_state_pos"] = np.array([st[2], st[3]])
    out["np_state_gauss"] = np.array(st[4])
    pst = random.getstate()
    out["py_state"] = np.array(pst[1], dtype=np.uint64)
    orig = random.shuffle
    random.shuffle = shuffle_py2
    kmeans.random.shuffle = shuffle_py2
    km = kmeans.KMeans(X, K, "spread")
    random.shuffle = orig
    out["kmeans_init_assign"] = km.components.assignments.copy()
    out["kmeans_random_means"] = km.components.random_means.copy()
    rec = km.fit(n_iter)
    out["kmeans_sum_neg_sqrd_norm"] = np.array(rec["sum_neg_sqrd_norm"])
    out["kmeans_n_mean_updates"] = np.array(rec["n_mean_updates"])
    out["kmeans_final_assign"] = km.components.assignments.copy()
    out["kmeans_final_means"] = km.components.means.copy()
    np.savez_compressed(os.path.join(HERE, "notebook.npz"), **out)
    print("notebook.npz: kmeans", rec["sum_neg_sqrd_norm"][:3], rec["n_mean_updates"])
    print("             fbgmm log_marg", out["fbgmm_log_marg"][:4])


if __name__ == "__main__":
    if len(sys.argv) > 2:
        for which in sys.argv[2:]:
            globals()["gen_" + which]()
        sys.exit(0)
    gen_kernels()
    gen_gauss()
    gen_chains()
    gen_bigram()
    gen_amgibbs()
    gen_edge()
    gen_notebook()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
