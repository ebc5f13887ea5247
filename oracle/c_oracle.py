"""
c_oracle.py -- ctypes binding of oracle/libsegk_oracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY (see segk_oracle.c header).  Build with `make -C oracle`
(also done by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsegk_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_pairwise_sum.restype = C.c_double
        L.orc_pairwise_sum.argtypes = [_dp, C.c_int64]
        L.orc_pairwise_sum_f32.restype = C.c_float
        L.orc_pairwise_sum_f32.argtypes = [_fp, C.c_int64]
        L.orc_neg_sqrd_norm_f32.argtypes = [_fp, C.c_int64, C.c_int64, _fp, _fp]
        L.orc_neg_sqrd_norm_f64.argtypes = [_dp, C.c_int64, C.c_int64, _dp, _dp]
        L.orc_kmeans_max_argmax.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                            C.c_int64, _ip, C.c_int64, _dp, _ip]
        L.orc_logsumexp.restype = C.c_double
        L.orc_logsumexp.argtypes = [_dp, C.c_int64]
        L.orc_draw.restype = C.c_int64
        L.orc_draw.argtypes = [_dp, C.c_int64, C.c_double]
        L.orc_build_vec.argtypes = [_ip, _dp, _dp, C.c_int64, C.c_int, C.c_double, C.c_double, _dp]
        L.orc_fb_kmeans_viterbi.restype = C.c_double
        L.orc_fb_kmeans_viterbi.argtypes = [_dp, C.c_int64, C.c_int64, C.c_int64, _bp, _dp]
        L.orc_fb_viterbi.restype = C.c_double
        L.orc_fb_viterbi.argtypes = [_dp, C.c_int64, C.c_int64, C.c_int64, _bp, _dp]
        L.orc_forward_backward.restype = C.c_double
        L.orc_forward_backward.argtypes = [_dp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                                           _dp, _bp, _dp, _ip, C.POINTER(C.c_int)]
        L.orc_fixedvar_log_post_pred.argtypes = [_dp, _dp, _dp, _dp, C.c_int64, C.c_int64, _fp, _dp]
        L.orc_fixedvar_log_prior.restype = C.c_double
        L.orc_fixedvar_log_prior.argtypes = [_dp, _dp, C.c_int64, _fp]
        L.orc_diag_log_post_pred.argtypes = [_dp, _dp, _dp, _ip, C.c_double, C.c_double, C.c_int64,
                                             C.c_int64, _fp, _dp]
        L.orc_diag_log_prior.restype = C.c_double
        L.orc_diag_log_prior.argtypes = [_dp, C.c_double, C.c_double, _dp, C.c_int64, _fp]
        L.orc_fbgmm_log_marg_i.restype = C.c_double
        L.orc_fbgmm_log_marg_i.argtypes = [_ip, C.c_int64, C.c_int64, C.c_double, C.c_double, _dp,
                                           C.c_double, _dp]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def pairwise_sum(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return np.float32(lib().orc_pairwise_sum_f32(_f(a), a.size))
    a = _c(a, np.float64)
    return lib().orc_pairwise_sum(_d(a), a.size)


def neg_sqrd_norm(means, x):
    """A1: dtype follows (means, x) -- both float32 or both float64."""
    K, D = means.shape
    if means.dtype == np.float32 and x.dtype == np.float32:
        m, xx = _c(means, np.float32), _c(x, np.float32)
        out = np.empty(K, np.float32)
        lib().orc_neg_sqrd_norm_f32(_f(m), K, D, _f(xx), _f(out))
        return out
    m, xx = _c(means, np.float64), _c(x, np.float64)
    out = np.empty(K, np.float64)
    lib().orc_neg_sqrd_norm_f64(_d(m), K, D, _d(xx), _d(out))
    return out


def kmeans_max_argmax(means, X, ids=None):
    """max / first argmax of A1 for rows `ids` of X (all rows when None)."""
    K, D = means.shape
    is_f64 = 0 if (means.dtype == np.float32 and X.dtype == np.float32) else 1
    dt = np.float64 if is_f64 else np.float32
    m, XX = _c(means, dt), _c(X, dt)
    if ids is None:
        n, ip = XX.shape[0], None
    else:
        ids = _c(ids, np.int64)
        n, ip = ids.size, _i(ids)
    out_max = np.empty(n, np.float64)
    out_arg = np.empty(n, np.int64)
    lib().orc_kmeans_max_argmax(m.ctypes.data, K, D, XX.ctypes.data, is_f64, XX.shape[1], ip, n,
                                _d(out_max), _i(out_arg))
    return out_max, out_arg


def logsumexp(a):
    a = _c(a, np.float64)
    return lib().orc_logsumexp(_d(a), a.size)


def draw(p, u):
    p = _c(p, np.float64)
    return int(lib().orc_draw(_d(p), p.size, float(u)))


def build_vec(vec_ids, durations, score, use_power, time_power_term, wip):
    vec_ids = _c(vec_ids, np.int64)
    durations = _c(durations, np.float64)
    score = _c(score, np.float64)
    out = np.empty(vec_ids.size, np.float64)
    lib().orc_build_vec(_i(vec_ids), _d(durations), _d(score), vec_ids.size, int(use_power),
                        float(time_power_term), float(wip), _d(out))
    return out


def fb_kmeans_viterbi(vec, N, n_min, n_max):
    vec = _c(vec, np.float64)
    b = np.zeros(N, np.uint8)
    g = np.empty(N, np.float64)
    tot = lib().orc_fb_kmeans_viterbi(_d(vec), N, n_min, n_max, b.ctypes.data_as(_bp), _d(g))
    return tot, b.astype(bool), g


def fb_viterbi(vec, N, n_min, n_max):
    vec = _c(vec, np.float64)
    b = np.zeros(N, np.uint8)
    g = np.empty(N, np.float64)
    tot = lib().orc_fb_viterbi(_d(vec), N, n_min, n_max, b.ctypes.data_as(_bp), _d(g))
    return tot, b.astype(bool), g


def forward_backward(vec, log_p_continue, N, n_min, n_max, anneal_temp, uniforms):
    vec = _c(vec, np.float64)
    uniforms = _c(uniforms, np.float64)
    assert uniforms.size >= N
    b = np.zeros(N, np.uint8)
    a = np.empty(N, np.float64)
    nd = C.c_int64(0)
    st = C.c_int(0)
    tot = lib().orc_forward_backward(_d(vec), float(log_p_continue), N, n_min, n_max, float(anneal_temp),
                                     _d(uniforms), b.ctypes.data_as(_bp), _d(a), C.byref(nd), C.byref(st))
    return tot, b.astype(bool), a, int(nd.value), int(st.value)


def fixedvar_log_post_pred(mu_N_numerators, precision_Ns, log_prod_precision_preds, precision_preds, K, x):
    D = mu_N_numerators.shape[1]
    out = np.empty(K, np.float64)
    x = _c(x, np.float32)
    a, b = _c(mu_N_numerators, np.float64), _c(precision_Ns, np.float64)
    c, d = _c(log_prod_precision_preds, np.float64), _c(precision_preds, np.float64)
    lib().orc_fixedvar_log_post_pred(_d(a), _d(b), _d(c), _d(d), K, D, _f(x), _d(out))
    return out


def fixedvar_log_prior(mu_0, precision_0, x):
    mu_0, precision_0, x = _c(mu_0, np.float64), _c(precision_0, np.float64), _c(x, np.float32)
    return lib().orc_fixedvar_log_prior(_d(mu_0), _d(precision_0), mu_0.size, _f(x))


def diag_log_post_pred(m_N_numerators, log_prod_vars, inv_vars, counts, k_0, v_0, K, x):
    D = m_N_numerators.shape[1]
    out = np.empty(K, np.float64)
    x = _c(x, np.float32)
    a, b, c = _c(m_N_numerators, np.float64), _c(log_prod_vars, np.float64), _c(inv_vars, np.float64)
    n = _c(counts, np.int64)
    lib().orc_diag_log_post_pred(_d(a), _d(b), _d(c), _i(n), float(k_0), float(v_0), K, D, _f(x), _d(out))
    return out


def diag_log_prior(m_0, k_0, v_0, S_0, x):
    m_0, S_0, x = _c(m_0, np.float64), _c(S_0, np.float64), _c(x, np.float32)
    return lib().orc_diag_log_prior(_d(m_0), float(k_0), float(v_0), _d(S_0), m_0.size, _f(x))


def fbgmm_log_marg_i(counts, K, alpha, lms, log_post_pred, log_prior):
    counts = _c(counts, np.int64)
    lpp = _c(log_post_pred, np.float64)
    logits = np.empty(counts.size, np.float64)
    r = lib().orc_fbgmm_log_marg_i(_i(counts), K, counts.size, float(alpha), float(lms), _d(lpp),
                                   float(log_prior), _d(logits))
    return r, logits
