/*
 * segk_oracle.c -- CPU restatement (plain C, fp64) of the segmentalist hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (segmentalist_amd/, the C-ABI
 * library libsegk.so) links, loads or calls this file.  It is used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker*.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced by importing the reference itself (tests/golden/
 * make_golden.py, which runs a throw-away py3 translation of /root/reference) and
 * against the constants asserted by the reference's own tests.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/segmentalist/).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* ------------------------------------------------------------------------- *
 * numpy's pairwise summation for a contiguous double vector (what
 * `(deltas*deltas).sum(axis=1)` executes per row; kmeans_components.py:226,
 * gaussian_components_fixedvar.py:252).  Order: n<8 sequential from 0.; n<=128
 * eight strided accumulators then the fixed combine tree, tail sequential;
 * n>128 split at n/2 rounded down to a multiple of 8.
 * ------------------------------------------------------------------------- */
double orc_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_sum(a, n2) + orc_pairwise_sum(a + n2, n - n2);
    }
}

/* float32 twin: numpy runs the same pairwise routine on float32 data. */
float orc_pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_sum_f32(a, n2) + orc_pairwise_sum_f32(a + n2, n - n2);
    }
}

/* ------------------------------------------------------------------------- *
 * A1  KMeansComponents.neg_sqrd_norm(i)            kmeans_components.py:225-226
 *     deltas = self.means - self.X[i]; return -(deltas*deltas).sum(axis=1)
 *
 * DTYPE CONTRACT (measured on the reference): `means` is created
 * as `random_means.copy()` (kmeans_components.py:75-76) and therefore has the
 * dtype of X.  With float32 embeddings (the wordseg case, unigram_acoustic_
 * wordseg.py:646) the WHOLE expression is float32 arithmetic: means are the
 * float64 quotient mean_numerators/counts rounded to float32 on store (:110),
 * the subtraction, the squaring and numpy's pairwise sum all run in float32 and
 * the result is a float32 vector.  With float64 X everything is float64.
 * ------------------------------------------------------------------------- */
void orc_neg_sqrd_norm_f32(const float *means, int64_t K, int64_t D, const float *x, float *out)
{
    float *sq = (float *)malloc(sizeof(float) * (size_t)(D > 0 ? D : 1));
    for (int64_t k = 0; k < K; k++) {
        const float *m = means + k * D;
        for (int64_t d = 0; d < D; d++) {
            volatile float delta = m[d] - x[d];
            volatile float q = delta * delta;
            sq[d] = q;
        }
        out[k] = -orc_pairwise_sum_f32(sq, D);
    }
    free(sq);
}

void orc_neg_sqrd_norm_f64(const double *means, int64_t K, int64_t D, const double *x, double *out)
{
    double *sq = (double *)malloc(sizeof(double) * (size_t)(D > 0 ? D : 1));
    for (int64_t k = 0; k < K; k++) {
        const double *m = means + k * D;
        for (int64_t d = 0; d < D; d++) {
            double delta = m[d] - x[d];
            sq[d] = delta * delta;
        }
        out[k] = -orc_pairwise_sum(sq, D);
    }
    free(sq);
}

/* max / first-argmax of neg_sqrd_norm for many rows:
 * max_neg_sqrd_norm_i / argmax_neg_sqrd_norm_i      kmeans_components.py:228-232
 * is_f64 selects the dtype of BOTH means and X.  out_max is widened to double
 * (the value the caller stores into the float64 DP vector,
 * kmeans_acoustic_wordseg.py:341). */
void orc_kmeans_max_argmax(const void *means, int64_t K, int64_t D,
                           const void *X, int is_f64, int64_t ldx,
                           const int64_t *ids, int64_t n,
                           double *out_max, int64_t *out_arg)
{
    if (is_f64) {
        double *s = (double *)malloc(sizeof(double) * (size_t)K);
        for (int64_t r = 0; r < n; r++) {
            int64_t e = ids ? ids[r] : r;
            orc_neg_sqrd_norm_f64((const double *)means, K, D, (const double *)X + e * ldx, s);
            int64_t best = 0;
            for (int64_t k = 1; k < K; k++)
                if (s[k] > s[best]) best = k;      /* np.argmax: first maximum */
            out_max[r] = s[best];
            out_arg[r] = best;
        }
        free(s);
    } else {
        float *s = (float *)malloc(sizeof(float) * (size_t)K);
        for (int64_t r = 0; r < n; r++) {
            int64_t e = ids ? ids[r] : r;
            orc_neg_sqrd_norm_f32((const float *)means, K, D, (const float *)X + e * ldx, s);
            int64_t best = 0;
            for (int64_t k = 1; k < K; k++)
                if (s[k] > s[best]) best = k;
            out_max[r] = (double)s[best];
            out_arg[r] = best;
        }
        free(s);
    }
}

/* ------------------------------------------------------------------------- *
 * A9  _cython_utils.logsumexp                        _cython_utils.pyx:13-25
 * ------------------------------------------------------------------------- */
double orc_logsumexp(const double *a, int64_t n)
{
    double mx = a[0], s = 0.0;
    for (int64_t j = 1; j < n; j++)
        if (a[j] > mx) mx = a[j];
    for (int64_t j = 0; j < n; j++) s += exp(a[j] - mx);
    return log(s) + mx;
}

/* A9  _cython_utils.draw / utils.draw with the uniform supplied by the caller
 *     _cython_utils.pyx:75-89, utils.py:10-21 */
int64_t orc_draw(const double *p, int64_t n, double u)
{
    for (int64_t i = 0; i < n; i++) {
        u = u - p[i];
        if (u < 0) return i;
    }
    return n - 1;
}

/* ------------------------------------------------------------------------- *
 * A5  build the DP input vector
 *     SegmentalKMeansWordseg.get_vec_embed_neg_len_sqrd_norms kmeans_acoustic_wordseg.py:334-351
 *     UnigramAcousticWordseg.get_vec_embed_log_probs        unigram_acoustic_wordseg.py:474-511
 * score[j] is the per-embedding score for vec_ids[j] (ignored where id == -1).
 * use_power != 0 applies durations**time_power_term (unigram), else plain
 * multiplication (k-means).
 * ------------------------------------------------------------------------- */
void orc_build_vec(const int64_t *vec_ids, const double *durations,
                   const double *score, int64_t n, int use_power,
                   double time_power_term, double wip, double *out)
{
    for (int64_t j = 0; j < n; j++) {
        double v = NEG_INF;
        if (vec_ids[j] != -1) {
            if (isnan(durations[j])) v = NEG_INF;
            else v = score[j] * (use_power ? pow(durations[j], time_power_term) : durations[j]);
        }
        out[j] = v + wip;
    }
}

/* Window helper shared by the three DPs: python `a[i:i+t][-n_max:cut]` selects
 * s in [lo, hi) where lo = max(0, t-n_max) (n_max==0 -> 0) and hi = t, or
 * t-(n_min-1) when n_min > 1 (n_slices_min_cut).  */
static inline int64_t win_lo(int64_t t, int64_t n_max)
{
    return (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
}

/* ------------------------------------------------------------------------- *
 * A8  forward_backward_kmeans_viterbi      kmeans_acoustic_wordseg.py:449-555
 * Supported n_slices_min in {0,1} (>=2 crashes in the reference).
 * Returns the summed score; boundaries (N bytes) written.  gammas_out (N
 * doubles) optional.
 * ------------------------------------------------------------------------- */
double orc_fb_kmeans_viterbi(const double *vec, int64_t N, int64_t n_min, int64_t n_max,
                             uint8_t *boundaries, double *gammas_out)
{
    (void)n_min;
    int64_t L = N * (N + 1) / 2;
    double *g = (double *)malloc(sizeof(double) * (size_t)N);
    for (int64_t j = 0; j < N; j++) { g[j] = 1.0; boundaries[j] = 0; }
    boundaries[N - 1] = 1;
    g[0] = 0.0;

    int64_t i = 0;
    for (int64_t t = 1; t < N; t++) {           /* :494-506 */
        double best = NEG_INF;
        for (int64_t s = win_lo(t, n_max); s < t; s++) {
            double v = vec[i + s] + g[s];
            if (v > best) best = v;
        }
        g[t] = best;                            /* all -inf -> -inf */
        i += t;
    }
    if (gammas_out) memcpy(gammas_out, g, sizeof(double) * (size_t)N);

    int64_t t = N;
    double total = 0.0;
    for (;;) {                                  /* :510-553 */
        i = (t - 1) * t / 2;
        int64_t lo = win_lo(t, n_max);
        int all_inf = 1;
        for (int64_t s = lo; s < t; s++)
            if (vec[i + s] + g[s] != NEG_INF) { all_inf = 0; break; }
        if (all_inf) {                          /* back-track :516-527 */
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (t - 1) * t / 2;
                lo = win_lo(t, n_max);
                all_inf = 1;
                for (int64_t s = lo; s < t; s++)
                    if (vec[i + s] + g[s] != NEG_INF) { all_inf = 0; break; }
            }
            boundaries[(t - 1 + N) % N] = 1;    /* python index t-1 (t==0 -> -1) */
        }
        /* q_t[::-1]; argmax -> shortest span among ties :529-531.  When t==0
         * was reached the stale (t==1) all -inf window is used: k = 1. */
        int64_t k = 1;
        if (t > 0) {
            double best = NEG_INF;
            int first = 1;
            for (int64_t s = t - 1; s >= lo; s--) {
                double v = vec[i + s] + g[s];
                if (first || v > best) { best = v; k = t - s; first = 0; }
            }
        }
        int64_t idx = i + t - k;
        if (idx < 0) idx += L;                  /* python negative index */
        total += vec[idx];
        if (t - k - 1 < 0) break;
        boundaries[t - k - 1] = 1;
        t = t - k;
    }
    free(g);
    return total;
}

/* ------------------------------------------------------------------------- *
 * A7  forward_backward_viterbi             unigram_acoustic_wordseg.py:759-864
 * (max forward, NO log_p_continue; back-pointer = argmax of the reversed,
 * normalised window -> ties and exp-rounding collapse resolved as in the
 * reference by comparing exp(log_p_k - logsumexp) values.)
 * ------------------------------------------------------------------------- */
double orc_fb_viterbi(const double *vec, int64_t N, int64_t n_min, int64_t n_max,
                      uint8_t *boundaries, double *alphas_out)
{
    (void)n_min;
    int64_t L = N * (N + 1) / 2;
    double *a = (double *)malloc(sizeof(double) * (size_t)N);
    double *w = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    for (int64_t j = 0; j < N; j++) { a[j] = 1.0; boundaries[j] = 0; }
    boundaries[N - 1] = 1;
    a[0] = 0.0;
    int64_t i = 0;
    for (int64_t t = 1; t < N; t++) {
        double best = NEG_INF;
        for (int64_t s = win_lo(t, n_max); s < t; s++) {
            double v = vec[i + s] + a[s];
            if (v > best) best = v;
        }
        a[t] = best;
        i += t;
    }
    if (alphas_out) memcpy(alphas_out, a, sizeof(double) * (size_t)N);

    int64_t t = N;
    double total = 0.0;
    for (;;) {
        i = (t - 1) * t / 2;
        int64_t lo = win_lo(t, n_max);
        int all_inf = 1;
        for (int64_t s = lo; s < t; s++)
            if (vec[i + s] + a[s] != NEG_INF) { all_inf = 0; break; }
        if (all_inf) {
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (t - 1) * t / 2;
                lo = win_lo(t, n_max);
                all_inf = 1;
                for (int64_t s = lo; s < t; s++)
                    if (vec[i + s] + a[s] != NEG_INF) { all_inf = 0; break; }
            }
            boundaries[(t - 1 + N) % N] = 1;
        }
        int64_t k = 1;
        if (t > 0) {
            int64_t n = t - lo;
            for (int64_t s = lo; s < t; s++) w[s - lo] = vec[i + s] + a[s];
            double lse = orc_logsumexp(w, n);            /* :843 */
            double best = 0.0;
            int first = 1;
            for (int64_t s = t - 1; s >= lo; s--) {
                double p = exp(w[s - lo] - lse);
                if (first || p > best) { best = p; k = t - s; first = 0; }
            }
        }
        int64_t idx = i + t - k;
        if (idx < 0) idx += L;
        total += vec[idx];
        if (t - k - 1 < 0) break;
        boundaries[t - k - 1] = 1;
        t = t - k;
    }
    free(a);
    free(w);
    return total;
}

/* ------------------------------------------------------------------------- *
 * A6  forward_backward (forward filtering, backward sampling)
 *                                          unigram_acoustic_wordseg.py:653-756
 * `uniforms` replaces the process-global random.random() stream: one value is
 * consumed per emitted segment (:739 -> _cython_utils.pyx:83); *n_draws returns
 * how many were used.  Return value NaN-safe: the reference asserts the result
 * is not -inf (:753) -> *status = 1 in that case.
 * ------------------------------------------------------------------------- */
double orc_forward_backward(const double *vec, double log_p_continue, int64_t N,
                            int64_t n_min, int64_t n_max, double anneal_temp,
                            const double *uniforms, uint8_t *boundaries,
                            double *alphas_out, int64_t *n_draws, int *status)
{
    (void)n_min;
    int64_t L = N * (N + 1) / 2;
    double *a = (double *)malloc(sizeof(double) * (size_t)N);
    double *w = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    double *p = (double *)malloc(sizeof(double) * (size_t)(N + 1));
    for (int64_t j = 0; j < N; j++) { a[j] = 1.0; boundaries[j] = 0; }
    boundaries[N - 1] = 1;
    a[0] = 0.0;
    int64_t i = 0;
    for (int64_t t = 1; t < N; t++) {           /* :691-703 */
        int64_t lo = win_lo(t, n_max), n = t - lo;
        int all_inf = 1;
        for (int64_t s = lo; s < t; s++) {
            w[s - lo] = vec[i + s] + a[s];
            if (w[s - lo] != NEG_INF) all_inf = 0;
        }
        a[t] = all_inf ? NEG_INF : orc_logsumexp(w, n) + log_p_continue;
        i += t;
    }
    if (alphas_out) memcpy(alphas_out, a, sizeof(double) * (size_t)N);

    int64_t t = N, nd = 0;
    double total = 0.0;
    *status = 0;
    for (;;) {                                  /* :709-751 */
        i = (t - 1) * t / 2;
        int64_t lo = win_lo(t, n_max);
        int all_inf = 1;
        for (int64_t s = lo; s < t; s++)
            if (vec[i + s] + a[s] != NEG_INF) { all_inf = 0; break; }
        if (all_inf) {
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (t - 1) * t / 2;
                lo = win_lo(t, n_max);
                all_inf = 1;
                for (int64_t s = lo; s < t; s++)
                    if (vec[i + s] + a[s] != NEG_INF) { all_inf = 0; break; }
            }
            boundaries[(t - 1 + N) % N] = 1;
        }
        int64_t n, k;
        if (t > 0) {
            n = t - lo;
            for (int64_t s = lo; s < t; s++) w[s - lo] = vec[i + s] + a[s];
        } else {                                 /* stale t==1 window: one -inf */
            n = 1;
            w[0] = NEG_INF;
        }
        double lse = orc_logsumexp(w, n);
        if (anneal_temp != 1.0) {                /* :731-736 */
            for (int64_t j = 0; j < n; j++) p[j] = w[n - 1 - j] - lse;
            double inv = 1. / anneal_temp;
            for (int64_t j = 0; j < n; j++) w[j] = inv * p[j];
            double lse2 = orc_logsumexp(w, n);
            for (int64_t j = 0; j < n; j++) p[j] = exp(w[j] - lse2);
        } else {
            for (int64_t j = 0; j < n; j++) p[j] = exp(w[n - 1 - j] - lse);   /* :738 */
        }
        k = orc_draw(p, n, uniforms[nd]) + 1;    /* :739 */
        nd++;
        int64_t idx = i + t - k;
        if (idx < 0) idx += L;
        total += vec[idx];
        if (t - k - 1 < 0) break;
        boundaries[t - k - 1] = 1;
        t = t - k;
    }
    if (total == NEG_INF) *status = 1;           /* :753 assert False */
    *n_draws = nd;
    free(a); free(w); free(p);
    return total;
}

/* ------------------------------------------------------------------------- *
 * A3  GaussianComponentsFixedVar.log_post_pred(i) gaussian_components_fixedvar.py:242-253
 *     and log_prior(i) :224-231 (-> _log_prod_norm :328-338,
 *     sum_square_a_times_b _cython_utils.pyx:63-70: sequential sum of a*a*b)
 * ------------------------------------------------------------------------- */
void orc_fixedvar_log_post_pred(const double *mu_N_numerators, const double *precision_Ns,
                                const double *log_prod_precision_preds,
                                const double *precision_preds, int64_t K, int64_t D,
                                const float *x, double *out)
{
    double c = -0.5 * (double)D * log(2. * M_PI);
    double *t = (double *)malloc(sizeof(double) * (size_t)(D > 0 ? D : 1));
    for (int64_t k = 0; k < K; k++) {
        for (int64_t d = 0; d < D; d++) {
            double mu = mu_N_numerators[k * D + d] / precision_Ns[k * D + d];
            double delta = mu - (double)x[d];
            t[d] = (delta * delta) * precision_preds[k * D + d];
        }
        out[k] = c + 0.5 * log_prod_precision_preds[k] - 0.5 * orc_pairwise_sum(t, D);
    }
    free(t);
}

double orc_fixedvar_log_prior(const double *mu_0, const double *precision_0, int64_t D,
                              const float *x)
{
    double c = -0.5 * (double)D * log(2. * M_PI);
    double slog = log(precision_0[0]);          /* sum_log :52-58 */
    for (int64_t d = 1; d < D; d++) slog += log(precision_0[d]);
    double ss = 0.0;
    for (int64_t d = 0; d < D; d++) {
        double delta = (double)x[d] - mu_0[d];
        ss += delta * delta * precision_0[d];
    }
    return c + 0.5 * slog - 0.5 * ss;
}

/* ------------------------------------------------------------------------- *
 * A2  GaussianComponentsDiag.log_post_pred(i)   gaussian_components_diag.py:237-259
 *     and log_prior(i) :215-222 (-> _log_prod_students_t :347-360)
 * lgamma tables of the reference (:128-131) are indexed by integer v; here
 * lgamma() is evaluated directly (same values to ~1 ulp).
 * ------------------------------------------------------------------------- */
void orc_diag_log_post_pred(const double *m_N_numerators, const double *log_prod_vars,
                            const double *inv_vars, const int64_t *counts,
                            double k_0, double v_0, int64_t K, int64_t D,
                            const float *x, double *out)
{
    double log_pi = log(M_PI);
    for (int64_t k = 0; k < K; k++) {
        double k_N = k_0 + (double)counts[k];
        double v_N = v_0 + (double)counts[k];
        double g = lgamma((v_N + 1.) / 2.) - lgamma(v_N / 2.);
        double s = 0.0;
        for (int64_t d = 0; d < D; d++) {
            double m = m_N_numerators[k * D + d] / k_N;
            double delta = m - (double)x[d];
            s += log(1. + (delta * delta) * inv_vars[k * D + d] * (1. / v_N));
        }
        out[k] = (double)D * (g - 0.5 * log(v_N) - 0.5 * log_pi)
                 - 0.5 * log_prod_vars[k] - (v_N + 1.) / 2. * s;
    }
}

double orc_diag_log_prior(const double *m_0, double k_0, double v_0, const double *S_0,
                          int64_t D, const float *x)
{
    double log_pi = log(M_PI);
    double lpv = 0.0, s = 0.0;
    for (int64_t d = 0; d < D; d++) {
        double var = (k_0 + 1.) / (k_0 * v_0) * S_0[d];
        lpv += log(var);
        double delta = (double)x[d] - m_0[d];
        s += log(1. + 1. / v_0 * (delta * delta) * (1. / var));
    }
    return (double)D * (lgamma((v_0 + 1.) / 2.) - lgamma(v_0 / 2.) - 0.5 * log(v_0) - 0.5 * log_pi)
           - 0.5 * lpv - (v_0 + 1.) / 2. * s;
}

/* ------------------------------------------------------------------------- *
 * A4  FBGMM.log_marg_i(i)                                   fbgmm.py:256-285
 * given the K-vector log_post_pred and scalar log_prior; writes the K_max
 * logits (before normalisation) if logits_out != NULL.
 * ------------------------------------------------------------------------- */
double orc_fbgmm_log_marg_i(const int64_t *counts, int64_t K, int64_t K_max,
                            double alpha, double lms, const double *log_post_pred,
                            double log_prior, double *logits_out)
{
    int64_t total = counts[0];                  /* sum_ints :41-47 */
    for (int64_t k = 1; k < K_max; k++) total += counts[k];
    double denom = log((double)total + alpha);
    double *z = logits_out ? logits_out : (double *)malloc(sizeof(double) * (size_t)K_max);
    for (int64_t k = 0; k < K_max; k++) {
        z[k] = lms * (log(alpha / (double)K_max + (double)counts[k]) - denom);
        z[k] += (k < K) ? log_post_pred[k] : log_prior;
    }
    double r = orc_logsumexp(z, K_max);
    if (!logits_out) free(z);
    return r;
}
