// segk_metrics.hip -- the per-sweep record metrics of the FBGMM / bigram drivers on the device:
//   FBGMM.log_prob_z                           fbgmm.py:208-225
//   BigramAcousticWordseg.log_prob_z           bigram_acoustic_wordseg.py:287-305
//   GaussianComponentsFixedVar.log_marg        gaussian_components_fixedvar.py:261-296
//   GaussianComponentsDiag.log_marg            gaussian_components_diag.py:271-303
// Before, these were numpy on host snapshots (4 MB of assignments, a stable sort of a million labels, for the
// bigram driver a Python loop over 80 000 tokens): three orders of magnitude more than the sweep they record.
//
// Parity notes.  The fixed-variance metric sums x and x^2 over a component's rows IN THE DTYPE OF X (numpy reduces
// a float32 matrix along axis 0 row after row in float32), so the device does the same: the component's rows in
// ascending order accumulated sequentially in XT.  The lists come from the stable counting sort of the k-means batch
// statistics (k_batch_sort) run over the ROWS (key = assignments[row], blocks of rows instead of blocks of
// utterances; token order would not do: the numbering of an utterance's spans is the corpus's business).
// Everything after those sums is float64, where the order of a sum changes the last bits only (the contract for
// record values is 1e-8).
// The bigram driver's log_prob_z is, in the reference, the sequential Polya-urn probability of the tokens under the
// smoothed unigram model (its loop never advances j_prev): sum_t log((n_t[k_t] + a/K) / (t + a)), which depends on
// the final counts only: sum_k [lgamma(n_k + a/K) - lgamma(a/K)] - [lgamma(T + a) - lgamma(a)].
#include "segk_kmeans_dev.h"
#include "segk_fb_common.h"

// segk_stats.hip
int segk_launch_batch_sort(const segk_corpus *c, const segk_kmeans *m, const int32_t *blk_lo, int n_blocks, const int32_t *new_tok,
                           const int32_t *new_k, const int32_t *n_flag, const double *out_total, int32_t *sorted, int32_t *koff,
                           double *part_tot, int32_t *flags, int cap, double *out_scalars, hipStream_t st);

#define METRIC_MAX_BLOCKS 64

// block bounds (rows) and the sort's dummy K
__global__ void k_metric_setup(int64_t n_emb, int K_max, int n_blocks, int32_t *blk_lo, int32_t *kdummy)
{
    const int b = threadIdx.x;
    if (b == 0) *kdummy = K_max;                       // the sort treats keys >= *K as flagged tokens: none here
    if (b <= n_blocks) blk_lo[b] = (int32_t)(((int64_t)b * n_emb) / n_blocks);
}

// term[k] = log_marg_k, one wave per component
template <typename XT>
__global__ __launch_bounds__(256) void k_metric_log_marg(segk_corpus c, segk_fbgmm f, const int32_t *blk_lo, int n_blocks,
                                                         const int32_t *sorted, const int32_t *koff_all, double *term)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= f.K_max) return;
    const int K = *f.K, D = c.D;
    if (k >= K) {
        if (lane == 0) term[k] = 0.0;
        return;
    }
    const double N = (double)f.counts[k];
    double s = 0.0;
    if (f.cov_type == 0) {
        // gaussian_components_fixedvar.py:261-283: X.sum(axis=0) and np.square(X).sum(axis=0) in the dtype of X, rows ascending
        const XT *X = (const XT *)c.X;
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + lane, dc = d < D ? d : 0;
            XT sx = (XT)0, sxx = (XT)0;
            for (int b = 0; b < n_blocks; b++) {
                const int32_t *koff = koff_all + (int64_t)b * (f.K_max + 1);
                const int64_t p0 = blk_lo[b];              // first row of the block
                const int q0 = koff[k], q1 = koff[k + 1];
                for (int qb = q0; qb < q1; qb += 8) {
                    XT xv[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int64_t e = p0 + sorted[p0 + (qb + q < q1 ? qb + q : q0)];     // clamped: always valid
                        xv[q] = X[e * c.ldx + dc];
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (qb + q < q1) {
                            sx += xv[q];
                            const XT sq = xv[q] * xv[q];
                            sxx += sq;
                        }
                }
            }
            if (d < D) {
                const double p = f.prior_a[d], m0 = f.prior_b[d], p0_ = f.prior_c[d];
                // np.square(X.sum(axis=0)) squares IN THE DTYPE OF X (a float32 array stays float32 under np.square and under
                // the Python scalar 2), only the products with the float64 prior vectors widen
                const XT sx2 = sx * sx;
                const double Sx = (double)sx, Sx2 = (double)sx2, Sxx = (double)sxx;
                const double den = N / p0_ + 1. / p;
                s += (N - 1) / 2. * log(p) - 0.5 * N * 1.8378770664093453 - 0.5 * log(den) - 0.5 * p * Sxx - 0.5 * p0_ * (m0 * m0)
                     + 0.5 * (Sx2 * p / p0_ + (m0 * m0) * p0_ / p + 2 * Sx * m0) / den;
            }
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) term[k] = s;
    } else {
        // gaussian_components_diag.py:271-290
        const double k_N = f.k_0 + N, v_N = f.v_0 + N;
        double ls0 = 0.0, lsn = 0.0;
        for (int d = lane; d < D; d += 64) {
            const double m_N = f.stat_a[(int64_t)k * D + d] / k_N;
            const double S_N = f.stat_b[(int64_t)k * D + d] - k_N * (m_N * m_N);
            ls0 += log(f.prior_a[d]);
            lsn += log(S_N);
        }
        for (int o = 32; o > 0; o >>= 1) {
            ls0 += __shfl_xor(ls0, o);
            lsn += __shfl_xor(lsn, o);
        }
        if (lane == 0)
            term[k] = -N * D / 2. * 1.1447298858494002 + D / 2. * log(f.k_0) - D / 2. * log(k_N) + f.v_0 / 2. * ls0 - v_N / 2. * lsn
                      + D * (lgamma(v_N / 2.) - lgamma(f.v_0 / 2.));
    }
}

// out = {log_prob_z, log_prob_X_given_z, K, n_assigned}
__global__ __launch_bounds__(1024) void k_metric_reduce(segk_fbgmm f, const double *term, int urn, double urn_a, double *out)
{
    __shared__ double red[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K = *f.K;
    const double conc = urn ? urn_a : f.alpha, per = conc / f.K_max;
    double lm = 0.0, lz = 0.0, tot = 0.0;
    for (int k = tid; k < f.K_max; k += blockDim.x) {
        if (k < K) lm += term[k];
        const double n = (double)f.counts[k];
        tot += n;
        if (!urn || n > 0.0) lz += lgamma(n + per) - lgamma(per);        // (an empty component contributes exactly 0 either way)
    }
    for (int o = 32; o > 0; o >>= 1) {
        lm += __shfl_xor(lm, o);
        lz += __shfl_xor(lz, o);
        tot += __shfl_xor(tot, o);
    }
    __shared__ double rt[16];
    if (lane == 0) { red[0][wv] = lm; red[1][wv] = lz; rt[wv] = tot; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) { a += red[0][w]; b += red[1][w]; t += rt[w]; }
        out[0] = lgamma(conc) - lgamma(conc + t) + b;
        out[1] = a;
        out[2] = (double)K;
        out[3] = t;
    }
}

#define RB_MISC_INTS 1024

int segk_rows_by_label(segk_ctx *ctx, const int32_t *labels, int64_t n, int K_max, const int32_t **blk_lo_out, int *n_blocks_out,
                       const int32_t **sorted_out, const int32_t **koff_out, void *stream)
{
    SEGK_REQUIRE(ctx && labels && n > 0 && n < ((int64_t)1 << 31), "segk_rows_by_label arguments");
    SEGK_REQUIRE(K_max > 0 && K_max <= 8192, "K_max <= 8192");
    hipStream_t st = (hipStream_t)stream;
    if (ctx->rb_cap < n) {
        if (ctx->rb_sorted) SEGK_CHECK_HIP(hipFree(ctx->rb_sorted));
        ctx->rb_sorted = nullptr;
        ctx->rb_cap = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->rb_sorted, sizeof(int32_t) * (size_t)n));
        ctx->rb_cap = n;
    }
    if (ctx->rb_K < K_max) {
        if (ctx->rb_koff) SEGK_CHECK_HIP(hipFree(ctx->rb_koff));
        if (ctx->rb_term) SEGK_CHECK_HIP(hipFree(ctx->rb_term));
        ctx->rb_koff = nullptr;
        ctx->rb_term = nullptr;
        ctx->rb_K = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->rb_koff, sizeof(int32_t) * (size_t)METRIC_MAX_BLOCKS * (K_max + 1)));
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->rb_term, sizeof(double) * (size_t)K_max));
        ctx->rb_K = K_max;
    }
    if (!ctx->rb_misc) SEGK_CHECK_HIP(hipMalloc((void **)&ctx->rb_misc, sizeof(int32_t) * RB_MISC_INTS));
    int n_blocks = (int)((n + 16383) / 16384);
    if (n_blocks < 1) n_blocks = 1;
    if (n_blocks > METRIC_MAX_BLOCKS) n_blocks = METRIC_MAX_BLOCKS;
    int32_t *blk_lo = ctx->rb_misc;                       // [68]
    int32_t *kdummy = ctx->rb_misc + 72;
    int32_t *flags = ctx->rb_misc + 80;                   // [64 * 6]
    double *dbl = reinterpret_cast<double *>(ctx->rb_misc + 512);      // part_tot [64], scalars [8]
    hipLaunchKernelGGL(k_metric_setup, dim3(1), dim3(128), 0, st, n, K_max, n_blocks, blk_lo, kdummy);
    // the k-means batch sort over ROWS: "utterances" of one slot each, keys = the label vector itself
    segk_corpus rows{};
    memset(&rows, 0, sizeof(rows));
    rows.N_max = 1;
    segk_kmeans m{};
    memset(&m, 0, sizeof(m));
    m.K = kdummy;
    m.K_max = K_max;
    m.mnorm_max = dbl + 72;
    if (int rc = segk_launch_batch_sort(&rows, &m, blk_lo, n_blocks, nullptr, labels, nullptr, nullptr, ctx->rb_sorted, ctx->rb_koff,
                                        dbl, flags, 1, dbl + 64, st))
        return rc;
    *blk_lo_out = blk_lo;
    *n_blocks_out = n_blocks;
    *sorted_out = ctx->rb_sorted;
    *koff_out = ctx->rb_koff;
    return SEGK_OK;
}

extern "C" {

int32_t segk_fbgmm_record_metrics(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, int32_t urn, double urn_a,
                                  double *out, void *stream)
{
    SEGK_REQUIRE(ctx && c && f && out, "arguments");
    SEGK_REQUIRE(f->cov_type == 0 || f->cov_type == 1, "cov_type");
    SEGK_REQUIRE(f->K_max <= 8192, "K_max <= 8192");
    hipStream_t st = (hipStream_t)stream;
    const int32_t *blk_lo = nullptr, *sorted = nullptr, *koff = nullptr;
    int n_blocks = 0;
    // (also sizes ctx->rb_term; the diagonal metric needs no lists, the call is cheap)
    if (int rc = segk_rows_by_label(ctx, f->assignments, c->n_emb, f->K_max, &blk_lo, &n_blocks, &sorted, &koff, stream)) return rc;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_metric_log_marg<XT>, dim3((f->K_max + 3) / 4), dim3(256), 0, st, *c, *f, blk_lo, n_blocks,
                                       sorted, koff, ctx->rb_term););
    hipLaunchKernelGGL(k_metric_reduce, dim3(1), dim3(1024), 0, st, *f, ctx->rb_term, urn, urn_a, out);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

}  // extern "C"
