// segk_fbgmm.hip -- gfx950 kernels of the FBGMM (collapsed finite Bayesian GMM) Gibbs path,
// sequential-exact mode: everything one utterance needs, in the reference's order, in fp64.
//
//   k_fbgmm_update      A11  del_item / add_item / del_component for fixed-variance and diagonal
//                            components (gaussian_components_fixedvar.py:153-221, _diag.py:162-213)
//   k_fbgmm_score       A2/A3/A4  log_marg_i(e) for a list of embeddings (fbgmm.py:256-285)
//   k_unigram_segment   A5/A6/A7  vec building + forward filtering / backward sampling (or Viterbi)
//                            for one utterance, uniforms taken from a device-resident stream
//   k_fbgmm_assign      A10  gibbs_sample_inside_loop_i / map_assign_i for the new segments of one
//                            utterance, sequentially, statistics updated between segments
//
// Tolerance contract (BASELINE north_star): log-likelihoods within 1e-4 relative of the
// reference; everything here is fp64 with device libm, which lands at ~1e-15.
#include <stdlib.h>
#include <vector>

#include "segk_internal.h"
#include "segk_fb_common.h"
#include "segk_chain_barrier.h"

// The persistent chain (k_fb_chain) keeps the labels of ONE utterance's rows in LDS (f.assignments then addresses that
// staging array as if it began at row 0): a component that empties is relabelled in those rows at once, the rows of all
// other utterances are relabelled between two launches (nobody reads them before), from this log.
struct FbLocal {
    int32_t *asg;          // the staged labels (LDS), entry i = row row0 + i
    int64_t row0;
    int nrows;
    int32_t *relog;        // [2 * FB_RELOG] pairs (from, to), in order
    int *n_relog;
    const double *ktab;    // fb_diag_const(count), count < ktab_n (diagonal components), or NULL
    int64_t ktab_n;
};
#define FB_RELOG 16

// ---------------------------------------------------------------------------------------
// derived statistics of component k (all threads of the block cooperate over D)
//   fixed: precision_pred = P_N * P / (P_N + P); log_prod = sum log(precision_pred)   (:317-325)
//   diag : var = (k_N+1)/(k_N v_N) (S_N_partial - k_N m_N^2); log_prod_vars = sum log var;
//          inv_vars = 1/var                                                            (:332-345)
// ---------------------------------------------------------------------------------------
// x-independent constant of a diagonal component with `cnt` items: D*(lgamma((v_N+1)/2) -
// lgamma(v_N/2) - log(v_N)/2 - log(pi)/2)  (gaussian_components_diag.py:248-251; the reference
// reads the lgamma values from tables indexed by the count, :128-131)
__device__ double fb_diag_const(const segk_fbgmm &f, int D, double cnt)
{
    const double v_N = f.v_0 + cnt;
    return (double)D * (lgamma((v_N + 1.) / 2.) - lgamma(v_N / 2.) - 0.5 * log(v_N) - 0.5 * LOG_PI);
}

__device__ void fb_update_derived(const segk_fbgmm &f, int D, int k, double *red, const FbLocal *loc = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    double part = 0.0;
    if (f.cov_type == 0) {
        for (int d = tid; d < D; d += nt) {
            double pn = f.stat_b[(int64_t)k * D + d], p = f.prior_a[d];
            double pp = pn * p / (pn + p);
            f.pred[(int64_t)k * D + d] = pp;
            part += log(pp);
        }
    } else {
        const double cnt = (double)f.counts[k];
        const double k_N = f.k_0 + cnt, v_N = f.v_0 + cnt;
        for (int d = tid; d < D; d += nt) {
            double m_N = f.stat_a[(int64_t)k * D + d] / k_N;
            double var = (k_N + 1.) / (k_N * v_N) * (f.stat_b[(int64_t)k * D + d] - k_N * (m_N * m_N));
            f.pred[(int64_t)k * D + d] = 1. / var;
            part += log(var);
        }
    }
    // (k_fb_chain: fb_diag_const by count from a table the same function filled once -- two lgamma calls by one thread
    // were 2-4 us of every add_item / del_item; the reference reads its lgamma values from tables indexed by the count
    // too, gaussian_components_diag.py:128-131.  The table is in global memory: the load is issued before the block-wide
    // sum, not after it -- a dependent round trip of 1.5-2 us per add_item / del_item otherwise.)
    const bool from_tab = f.cov_type != 0 && loc && loc->ktab && f.counts[k] >= 0 && f.counts[k] < loc->ktab_n;
    double kc_tab = 0.0;
    if (from_tab && tid == 0) kc_tab = loc->ktab[f.counts[k]];
    double tot = block_sum(part, red);
    if (tid == 0) {
        f.log_prod[k] = tot;
        if (f.cov_type == 0) f.kconst[k] = -0.5 * (double)D * LOG_2PI;
        else if (from_tab) f.kconst[k] = kc_tab;
        else f.kconst[k] = fb_diag_const(f, D, (double)f.counts[k]);
    }
    __syncthreads();
}

template <typename XT>
__device__ void fb_del_component(const segk_corpus &c, const segk_fbgmm &f, int k, int *shK, const FbLocal *loc = nullptr)
{
    // *shK already decremented
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const int K = *shK;
    if (loc && tid == 0) {                       // (k == K too: the log also tells the launcher that a component went)
        const int n = *loc->n_relog;
        if (n < FB_RELOG) { loc->relog[2 * n] = K; loc->relog[2 * n + 1] = k; }
        *loc->n_relog = n + 1;
    }
    if (k != K) {
        for (int d = tid; d < D; d += nt) {
            f.stat_a[(int64_t)k * D + d] = f.stat_a[(int64_t)K * D + d];
            f.stat_b[(int64_t)k * D + d] = f.stat_b[(int64_t)K * D + d];
            f.pred[(int64_t)k * D + d] = f.pred[(int64_t)K * D + d];
        }
        if (loc) {
            for (int i = tid; i < loc->nrows; i += nt)
                if (loc->asg[i] == K) loc->asg[i] = k;
        } else {
            for (int64_t e = tid; e < c.n_emb; e += nt)
                if (f.assignments[e] == K) f.assignments[e] = k;
        }
    }
    __syncthreads();
    for (int d = tid; d < D; d += nt) {
        f.stat_a[(int64_t)K * D + d] = 0.0;
        f.stat_b[(int64_t)K * D + d] = 0.0;
        f.pred[(int64_t)K * D + d] = 0.0;
    }
    if (tid == 0) {
        if (k != K) {
            f.log_prod[k] = f.log_prod[K];
            f.kconst[k] = f.kconst[K];
            f.counts[k] = f.counts[K];
        }
        f.log_prod[K] = 0.0;
        f.kconst[K] = 0.0;
        f.counts[K] = 0;
    }
    if (f.lm_unigram) {          // gaussian_components_fixedvar.py:204-221, same statement order
        const int KM = f.K_max;
        __syncthreads();
        if (k != K) {
            if (tid == 0) f.lm_unigram[k] = f.lm_unigram[K];
            for (int q = tid; q < KM; q += nt) f.lm_bigram[(int64_t)k * KM + q] = f.lm_bigram[(int64_t)K * KM + q];
            __syncthreads();
            for (int q = tid; q < KM; q += nt) f.lm_bigram[(int64_t)q * KM + k] = f.lm_bigram[(int64_t)q * KM + K];
            __syncthreads();
        }
        if (tid == 0) f.lm_unigram[K] = 0;
        for (int q = tid; q < KM; q += nt) f.lm_bigram[(int64_t)K * KM + q] = 0;
        __syncthreads();
        for (int q = tid; q < KM; q += nt) f.lm_bigram[(int64_t)q * KM + K] = 0;
    }
    __syncthreads();
}

// x and x^2 as the reference sees them: X[i] is widened to double; the diagonal model caches
// np.square(X) in the dtype of X (gaussian_components_diag.py:125) -> float32 square for f32 data
template <typename XT>
__device__ __forceinline__ double x_sq(XT x)
{
    XT q = x * x;
    return (double)q;
}

template <typename XT>
__device__ void fb_add_item(const segk_corpus &c, const segk_fbgmm &f, int64_t e, int k_in, int *shK, int *sh_i,
                            double *red, const FbLocal *loc = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const XT *X = (const XT *)c.X;
    if (tid == 0) {
        *sh_i = (k_in == *shK) ? 1 : 0;
        if (k_in == *shK) *shK = *shK + 1;
    }
    __syncthreads();
    const int is_new = *sh_i;
    const int k = k_in;
    for (int d = tid; d < D; d += nt) {
        const double x = (double)X[e * c.ldx + d];
        double a = f.stat_a[(int64_t)k * D + d], b = f.stat_b[(int64_t)k * D + d];
        if (f.cov_type == 0) {
            if (is_new) { a = f.prior_c[d] * f.prior_b[d]; b = f.prior_c[d]; }     // precision_0*mu_0, precision_0
            a += f.prior_a[d] * x;
            b += f.prior_a[d];
        } else {
            if (is_new) { a = f.k_0 * f.prior_b[d]; b = f.prior_a[d] + f.k_0 * (f.prior_b[d] * f.prior_b[d]); }
            a += x;
            b += x_sq<XT>(X[e * c.ldx + d]);
        }
        f.stat_a[(int64_t)k * D + d] = a;
        f.stat_b[(int64_t)k * D + d] = b;
    }
    if (tid == 0) {
        f.counts[k] += 1;
        f.assignments[e] = k;
    }
    __syncthreads();
    fb_update_derived(f, D, k, red, loc);
}

template <typename XT>
__device__ void fb_del_item(const segk_corpus &c, const segk_fbgmm &f, int64_t e, int *shK, int *sh_i, double *red,
                            const FbLocal *loc = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const XT *X = (const XT *)c.X;
    __shared__ int sh_cnt0;
    if (tid == 0) {
        int k = f.assignments[e];
        if (k != -1) {
            f.counts[k] -= 1;
            f.assignments[e] = -1;
            sh_cnt0 = (f.counts[k] == 0) ? 1 : 0;
        }
        *sh_i = k;
    }
    __syncthreads();
    const int k = *sh_i;
    if (k == -1) return;
    if (sh_cnt0) {
        if (tid == 0) *shK = *shK - 1;
        __syncthreads();
        fb_del_component<XT>(c, f, k, shK, loc);
    } else {
        for (int d = tid; d < D; d += nt) {
            const double x = (double)X[e * c.ldx + d];
            if (f.cov_type == 0) {
                f.stat_a[(int64_t)k * D + d] -= f.prior_a[d] * x;
                f.stat_b[(int64_t)k * D + d] -= f.prior_a[d];
            } else {
                f.stat_a[(int64_t)k * D + d] -= x;
                f.stat_b[(int64_t)k * D + d] -= x_sq<XT>(X[e * c.ldx + d]);
            }
        }
        __syncthreads();
        fb_update_derived(f, D, k, red, loc);
    }
}

// op 0: delete the OLD segments of utterance `utt` (listed from the boundaries + vec_ids)
// op 1: add_item(item, k_item)   op 2: del_item(item)   op 4: del_component(k_item)
// (fb_nt(f) threads, like the assignment kernel and the persistent chain: fb_update_derived's block-wide sum of D logarithms
// associates by the number of threads once D > 256 -- launched with 256 threads whatever the bank's width, the removal of an
// utterance's old segments left log_prod a last bit away from what the other kernels compute from the same statistics, and the
// span scores of that utterance with it: tools/diag_chain_dsweep.py)
template <typename XT>
__global__ __launch_bounds__(512) void k_fbgmm_update(segk_corpus c, segk_fbgmm f, int op, int utt, int64_t item, int k_item,
                               const uint8_t *boundaries)
{
    __shared__ int shK, sh_i;
    __shared__ double red[512];
    if (threadIdx.x == 0) shK = *f.K;
    __syncthreads();
    if (op == 0) {
        const int N = c.lengths[utt];
        const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
        const int32_t *vid = c.vec_ids + (int64_t)utt * triMax;
        const uint8_t *bnd = boundaries + (int64_t)utt * c.N_max;
        int jp = 0;
        for (int j = 0; j < N; j++) {
            if (bnd[j]) {               // uniform: every thread reads the same byte
                int id = vid[(j + 1) * j / 2 + jp];
                jp = j + 1;
                if (id >= 0) fb_del_item<XT>(c, f, id, &shK, &sh_i, red);
            }
        }
    } else if (op == 1) {
        fb_add_item<XT>(c, f, item, k_item, &shK, &sh_i, red);
    } else if (op == 2) {
        fb_del_item<XT>(c, f, item, &shK, &sh_i, red);
    } else if (op == 4) {
        if (threadIdx.x == 0) shK = shK - 1;
        __syncthreads();
        fb_del_component<XT>(c, f, k_item, &shK);
    } else if ((op == 5 || op == 6) && f.lm_unigram) {
        // lm.remove_counts_from_utterance / lm.counts_from_utterance over the CURRENT transcript
        // (bigram_lms.py:98-114): sequential integer updates, one thread
        // integer adds commute, so utterances may go in parallel (utt < 0: the whole corpus,
        // set_lm_counts, bigram_acoustic_wordseg.py:271-276)
        const long long sgn = (op == 6) ? 1 : -1;
        const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
        const int u_lo = utt < 0 ? (int)threadIdx.x : utt, u_hi = utt < 0 ? c.n_utt : utt + 1;
        const int u_step = utt < 0 ? (int)blockDim.x : 1;
        for (int u = u_lo; u < u_hi && (utt < 0 || threadIdx.x == 0); u += u_step) {
            const int N = c.lengths[u];
            const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
            const uint8_t *bnd = boundaries + (int64_t)u * c.N_max;
            int jp = 0, kprev = -1;
            for (int j = 0; j < N; j++)
                if (bnd[j]) {
                    int64_t id = vid[(j + 1) * j / 2 + jp];
                    jp = j + 1;
                    if (id < 0) id += c.n_emb;          // python assignments[-1]
                    int k = f.assignments[id];
                    if (k < 0) k += f.K_max;            // python unigram_counts[-1]
                    atomicAdd((unsigned long long *)&f.lm_unigram[k], (unsigned long long)sgn);
                    if (kprev >= 0)
                        atomicAdd((unsigned long long *)&f.lm_bigram[(int64_t)kprev * f.K_max + k], (unsigned long long)sgn);
                    kprev = k;
                }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *f.K = shK;
}

// ---------------------------------------------------------------------------------------
// per-component log predictive of embedding x (A2/A3) and the prior predictive
// ---------------------------------------------------------------------------------------
// sum over the dimensions d0, d0+dstep, ... of the x-dependent terms of component k
template <typename XT>
__device__ __forceinline__ double fb_pred_sum(const segk_fbgmm &f, int D, int k, const XT *x, int d0, int dstep)
{
    // the statistics of eight of the lane's dimensions are fetched together (a load pair per loop iteration was a dependent
    // round trip per dimension: ten of them per token and lane at D = 39 with four lanes per component); the terms and
    // their order are unchanged
    double s = 0.0;
    const double *sa = f.stat_a + (int64_t)k * D, *sb = f.stat_b + (int64_t)k * D, *pp = f.pred + (int64_t)k * D;
    if (f.cov_type == 0) {      // gaussian_components_fixedvar.py:242-253
        for (int dd = d0; dd < D; dd += 8 * dstep) {
            double a[8], b[8], p[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = dd + j * dstep;
                a[j] = b[j] = p[j] = 1.0;
                if (d < D) { a[j] = sa[d]; b[j] = sb[d]; p[j] = pp[d]; }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = dd + j * dstep;
                if (d < D) {
                    double mu = a[j] / b[j];
                    double delta = mu - (double)x[d];
                    s += (delta * delta) * p[j];
                }
            }
        }
    } else {                    // gaussian_components_diag.py:237-259
        const double cnt = (double)f.counts[k];
        const double k_N = f.k_0 + cnt, v_N = f.v_0 + cnt;
        for (int dd = d0; dd < D; dd += 8 * dstep) {
            double a[8], p[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = dd + j * dstep;
                a[j] = p[j] = 0.0;
                if (d < D) { a[j] = sa[d]; p[j] = pp[d]; }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int d = dd + j * dstep;
                if (d < D) {
                    double m = a[j] / k_N;
                    double delta = m - (double)x[d];
                    s += log(1. + (delta * delta) * p[j] * (1. / v_N));
                }
            }
        }
    }
    return s;
}

__device__ __forceinline__ double fb_pred_finish(const segk_fbgmm &f, int k, double s)
{
    if (f.cov_type == 0) return f.kconst[k] + 0.5 * f.log_prod[k] - 0.5 * s;
    const double v_N = f.v_0 + (double)f.counts[k];
    return f.kconst[k] - 0.5 * f.log_prod[k] - (v_N + 1.) / 2. * s;
}

template <typename XT>
__device__ double fb_log_post_pred_k(const segk_fbgmm &f, int D, int k, const XT *x)
{
    return fb_pred_finish(f, k, fb_pred_sum<XT>(f, D, k, x, 0, 1));
}

// prior predictive (an empty component): x-dependent sum over d0, d0+dstep, ... and the finish;
// the x-independent part is kconst[K_max] (k_fbgmm_init_stats)
template <typename XT>
__device__ __forceinline__ double fb_prior_sum(const segk_fbgmm &f, int D, const XT *x, int d0, int dstep)
{
    double s = 0.0;
    if (f.cov_type == 0) {      // gaussian_components_fixedvar.py:224-231 (precision_0 as the predictive precision)
        for (int d = d0; d < D; d += dstep) {
            double delta = (double)x[d] - f.prior_b[d];
            s += delta * delta * f.prior_c[d];
        }
    } else {                    // gaussian_components_diag.py:215-222
        for (int d = d0; d < D; d += dstep) {
            double var = (f.k_0 + 1.) / (f.k_0 * f.v_0) * f.prior_a[d];
            double delta = (double)x[d] - f.prior_b[d];
            s += log(1. + 1. / f.v_0 * (delta * delta) * (1. / var));
        }
    }
    return s;
}

__device__ __forceinline__ double fb_prior_finish(const segk_fbgmm &f, double s)
{
    if (f.cov_type == 0) return f.kconst[f.K_max] - 0.5 * s;
    return f.kconst[f.K_max] - (f.v_0 + 1.) / 2. * s;
}

template <typename XT>
__device__ double fb_log_prior(const segk_fbgmm &f, int D, const XT *x)
{
    return fb_prior_finish(f, fb_prior_sum<XT>(f, D, x, 0, 1));
}

// logits z[k], k < K_max, of embedding `e` into LDS.  Assignment prior by `mode`:
//   0  FBGMM.log_marg_i:   lms*(log(alpha/K_max + counts) - log(sum counts + alpha))   fbgmm.py:268-272
//   1  gibbs_sample_inside_loop_i: lms*log(alpha/K_max + counts)                        fbgmm.py:436-440
//   2  map_assign_i:       log(alpha/K_max + counts)                                    fbgmm.py:475-479
//   3  LM unigram:         lms*lm.log_prob_vec_i()             bigram_lms.py:64-69 (scoring and first segment)
//   4  LM bigram:          lms*log(lm.prob_vec_given_j(j_prev))                          bigram_lms.py:84-91
// Work split: a group of G lanes (G | 64, K_max*G <= blockDim where possible) shares one
// component and strides over the dimensions; partial sums are combined by a butterfly.
// pred_out (LDS, [K_max]): receives the predictive term of every active component (what is added to the assignment prior).
// pred_in / K_in / modf: those terms as an EARLIER call for the same row computed them (with K_in active components); they
// stand for components below K_in whose flag in modf is clear -- the caller vouches that their statistics have not changed
// since --, the others are computed here, by the same lanes in the same order: the logits are the same bits either way.
template <typename XT>
__device__ void fb_logits(const segk_corpus &c, const segk_fbgmm &f, int64_t e, int mode, int j_prev, XT *xrow,
                          double *z, double *red, const double *lprior_tab = nullptr, double *pred_out = nullptr,
                          const double *pred_in = nullptr, int K_in = 0, const uint8_t *modf = nullptr)
{
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const XT *X = (const XT *)c.X;
    const int K = *f.K;
    const int KM = f.K_max;
    __syncthreads();
    for (int d = tid; d < D; d += nt) xrow[d] = X[e * c.ldx + d];
    double csum = 0.0;
    const int64_t *cnts = (mode >= 3) ? f.lm_unigram : f.counts;
    double total = 0.0;
    if (mode == 1 || mode == 2) {
        __syncthreads();                                // (the assignment priors of these modes do not use the total) publishes xrow
    } else {
        for (int k = tid; k < KM; k += nt) csum += (double)cnts[k];
        total = block_sum(csum, red);                   // exact: integer-valued (also publishes xrow)
    }
    // (lprior_tab: this very expression evaluated once per row by k_fb_prior_tab with the same number of threads -- it depends
    // on the row and the prior only; the persistent chain passes it)
    double lprior = 0.0;
    if (K < KM) lprior = lprior_tab ? lprior_tab[e] : fb_prior_finish(f, block_sum(fb_prior_sum<XT>(f, D, xrow, tid, nt), red));
    int G = 1;
    while (G < 64 && KM * (G * 2) <= nt) G *= 2;
    const int g = tid & (G - 1), kk0 = tid / G, kstep = nt / G;
    for (int kb = 0; kb < KM; kb += kstep) {
        const int k = kb + kk0;
        double s = 0.0;
        const bool fresh = k < K && (!pred_in || k >= K_in || modf[k]);      // (the same for the G lanes of a component)
        if (fresh) s = fb_pred_sum<XT>(f, D, k, xrow, g, G);
        // (the same xor butterfly, o = G/2 ... 1; the steps inside a quad by DPP instead of two ds_bpermute round trips)
        for (int o = G >> 1; o > 2; o >>= 1) s += __shfl_xor(s, o);
        if (G >= 4) s += fb_dpp_f64<0x4E>(s);            // xor 2
        if (G >= 2) s += fb_dpp_f64<0xB1>(s);            // xor 1
        if (g == 0 && k < KM) {
            double v;
            if (mode == 0) v = f.lms * (log(f.alpha / (double)KM + (double)f.counts[k]) - log(total + f.alpha));
            else if (mode == 1) v = f.lms * log(f.alpha / (double)KM + (double)f.counts[k]);
            else if (mode == 2) v = log(f.alpha / (double)KM + (double)f.counts[k]);
            else if (mode == 3) v = (log((double)f.lm_unigram[k] + f.lm_a / (double)KM) - log(total + f.lm_a)) * f.lms;
            else {
                const double pi = ((double)f.lm_unigram[k] + f.lm_a / (double)KM) / (total + f.lm_a);
                const double pij = (1. - f.lm_lambda) * ((double)f.lm_bigram[(int64_t)j_prev * KM + k] + f.lm_b / (double)KM)
                                   / ((double)f.lm_unigram[j_prev] + f.lm_b);
                v = log(f.lm_lambda * pi + pij) * f.lms;
            }
            double pk = lprior;
            if (k < K) {
                pk = fresh ? fb_pred_finish(f, k, s) : pred_in[k];
                if (pred_out) pred_out[k] = pk;
            }
            z[k] = v + pk;
        }
    }
    __syncthreads();
}

// A4: out[ids[r]] = log_marg_i(ids[r]) for r < n (entries -1 skipped).  One workgroup per row.
template <typename XT>
__global__ __launch_bounds__(512) void k_fbgmm_score(segk_corpus c, segk_fbgmm f, const int32_t *ids, int64_t row0, double *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *z = (double *)smem;                      // [K_max]
    double *red = z + f.K_max;                       // [nt]
    XT *xrow = (XT *)(red + blockDim.x);             // [D]
    const int64_t e = ids ? (int64_t)ids[blockIdx.x] : row0 + blockIdx.x;
    if (e < 0) return;
    fb_logits<XT>(c, f, e, f.lm_unigram ? 3 : 0, -1, xrow, z, red);
    double mx = NEG_INF_D;
    for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) mx = z[k] > mx ? z[k] : mx;
    mx = block_max(mx, red);
    double s = 0.0;
    for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) s += exp(z[k] - mx);
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[e] = log(s) + mx;
}

// A2/A3 vector API: out[k] = log_post_pred_k(row) for k < K, 0 for K <= k < K_max,
// out[K_max] = log_prior(row)
template <typename XT>
__global__ void k_fbgmm_pred_vector(segk_corpus c, segk_fbgmm f, int64_t row, double *out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const XT *x = (const XT *)c.X + row * c.ldx;
    const int K = *f.K;
    if (k < K) out[k] = fb_log_post_pred_k<XT>(f, c.D, k, x);
    else if (k < f.K_max) out[k] = 0.0;
    else if (k == f.K_max) out[k] = fb_log_prior<XT>(f, c.D, x);
}

// ---------------------------------------------------------------------------------------
// A5 + A6/A7 for one utterance: vec from the per-embedding scores, DP, new boundaries.
// Single thread does the DP (N <= N_max landmarks, fp64, reference order); uniforms are taken
// from ustream[*ucursor ...] and the cursor advanced (one per backward-sampling step).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_unigram_segment(segk_corpus c, int utt, int viterbi, int n_min, int n_max, double wip,
                                  double time_power_term, double log_p_continue, double anneal_temp,
                                  const double *score, const double *ustream, int64_t *ucursor, int64_t ucap,
                                  uint8_t *boundaries, int32_t *new_tok, int32_t *n_new, double *out_logprob,
                                  int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)n_min;
    const int N = c.lengths[utt];
    const int tri = N * (N + 1) / 2;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const FbSpanTab tab = fb_span_tab(c, utt, N, viterbi == 2 ? 0 : n_max);
    const int32_t *vid = tab.vid;
    double *vec = (double *)smem;           // [tri]
    double *a = vec + triMax;               // [N]
    double *w = a + c.N_max;                // [N+1]
    double *pr = w + c.N_max + 1;           // [N+1]
    fb_fill_vec(tab, N, tri, [&](int id) { return score[id]; }, time_power_term, wip, vec, threadIdx.x, blockDim.x);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    // wave 0 runs the DP: control flow and values are wave-uniform, the exponentials of each
    // logsumexp / normalisation are spread over the lanes, sums and draws keep the reference order
    const int lane = threadIdx.x;
    uint8_t *bnd = boundaries + (int64_t)utt * c.N_max;
    if (viterbi == 2) {      // assignments_only (bigram_acoustic_wordseg.py:386-387,548-549): keep the boundaries
        if (lane == 0) {
            out_logprob[utt] = 0.0;
            n_new[utt] = fb_collect_tokens(vid, bnd, N, new_tok + (int64_t)utt * c.N_max);
        }
        return;
    }
    StreamUniforms usrc = {ustream, *ucursor, ucap, status};
    const double total = fb_dp_sample(vec, a, w, pr, N, tri, n_max, viterbi, log_p_continue, anneal_temp, bnd, lane, usrc);
    const int64_t cur = usrc.cur;
    const int nn = N <= 64 ? fb_collect_tokens_wave(vid, bnd, N, new_tok + (int64_t)utt * c.N_max, lane) : -1;
    if (lane != 0) return;
    if (!viterbi && total == NEG_INF_D) atomicOr(status, 16);      // unigram_acoustic_wordseg.py:753
    *ucursor = cur;
    out_logprob[utt] = total;
    n_new[utt] = nn >= 0 ? nn : fb_collect_tokens(vid, bnd, N, new_tok + (int64_t)utt * c.N_max);
}

// softmax of the logits in z (scipy logsumexp order: max-shift, sum, log), optional annealing
// (fbgmm.py:446-449), then utils.draw in forward order with one uniform of the stream -- or the
// first maximum when map_assign -- and the `k > K -> K` clamp (:459-460).  Result in *sh_k.
__device__ void fb_draw_component(const segk_fbgmm &f, double *z, double *red, int map_assign, double anneal_temp,
                                  const double *ustream, int64_t *ucursor, int64_t ucap, int32_t *status, int shK,
                                  int *sh_k_out, int64_t ubase = 0)      // ustream[0] is value `ubase` of the stream
{
    // scipy logsumexp: max-shift, sum, log
    double mx = NEG_INF_D;
    for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) mx = z[k] > mx ? z[k] : mx;
    mx = block_max(mx, red);
    double s = 0.0;
    for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) s += exp(z[k] - mx);
    s = block_sum(s, red);
    double lse = log(s) + mx;
    if (!map_assign && anneal_temp != 1.0) {     // :446-449
        for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) z[k] = (1. / anneal_temp) * (z[k] - lse);
        __syncthreads();
        double mx2 = NEG_INF_D;
        for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) mx2 = z[k] > mx2 ? z[k] : mx2;
        mx2 = block_max(mx2, red);
        double s2 = 0.0;
        for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) s2 += exp(z[k] - mx2);
        s2 = block_sum(s2, red);
        lse = log(s2) + mx2;
    }
    for (int k = threadIdx.x; k < f.K_max; k += blockDim.x) z[k] = exp(z[k] - lse);      // prob_z
    __syncthreads();
    if (threadIdx.x < 64) {
        int k;
        if (map_assign) {                         // np.argmax(prob_z): first maximum
            double bm = z[0];
            k = 0;
            for (int q = threadIdx.x; q < f.K_max; q += 64)
                if (z[q] > bm) { bm = z[q]; k = q; }
            for (int o = 32; o > 0; o >>= 1) {
                const double om = __shfl_xor(bm, o);
                const int ok = __shfl_xor(k, o);
                if (om > bm || (om == bm && ok < k)) { bm = om; k = ok; }
            }
        } else {                                  // utils.draw (utils.py:10-21), forward order
            const int64_t cur = *ucursor;
            const double uu = (cur < ucap) ? ustream[cur - ubase] : 0.5;
            k = fb_draw_seq(z, f.K_max, uu);
            if (threadIdx.x == 0) {
                if (cur >= ucap) atomicOr(status, 8);
                *ucursor = cur + 1;
            }
        }
        if (k > shK) k = shK;                     // :459-460
        if (threadIdx.x == 0) *sh_k_out = k;
    }
}

// ---------------------------------------------------------------------------------------
// A10 for the new segments of one utterance, in order (fbgmm.py:422-494).  One workgroup.
// ---------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(512) void k_fbgmm_assign(segk_corpus c, segk_fbgmm f, int utt, int map_assign, int j_prev0, double anneal_temp,
                               const int32_t *new_tok, const int32_t *n_new, const double *ustream,
                               int64_t *ucursor, int64_t ucap, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *z = (double *)smem;                      // [K_max]
    double *red = z + f.K_max;                       // [nt]
    XT *xrow = (XT *)(red + blockDim.x);             // [D]
    __shared__ int shK, sh_i, sh_k, sh_jprev;
    if (threadIdx.x == 0) { shK = *f.K; sh_jprev = j_prev0; }
    __syncthreads();
    const int nn = n_new[utt];
    for (int t = 0; t < nn; t++) {
        const int64_t e = new_tok[(int64_t)utt * c.N_max + t];
        if (threadIdx.x == 0) *f.K = shK;            // fb_logits reads K from memory
        __syncthreads();
        {
            int mode = map_assign ? 2 : 1;
            if (f.lm_unigram) mode = (sh_jprev < 0) ? 3 : 4;
            fb_logits<XT>(c, f, e, mode, sh_jprev, xrow, z, red);
        }
        fb_draw_component(f, z, red, map_assign, anneal_temp, ustream, ucursor, ucap, status, shK, &sh_k);
        __syncthreads();
        fb_add_item<XT>(c, f, e, sh_k, &shK, &sh_i, red);
        if (threadIdx.x == 0) sh_jprev = sh_k;      // bigram_acoustic_wordseg.py:482-494
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) *f.K = shK;
}


// ---------------------------------------------------------------------------------------
// The reference's serial Gibbs chain (gibbs_sample_i for one utterance after the other,
// unigram_acoustic_wordseg.py:252-360, 437-457) as ONE persistent kernel (round 4).
//
// The four launches per utterance (remove, score, sample boundaries, assign) cost 212 us per utterance at
// configs[1] (D = 39, K = 100): four kernel boundaries and, in the assignment kernel, a dozen dependent round trips
// through statistics in global memory per token, every one of them paid for every utterance because utterance
// i + 1 needs the statistics utterance i leaves behind.  Here G workgroups stay resident for a whole sweep:
//
//   * EVERY workgroup holds the whole model in LDS (statistics, derived terms, counts, K: 3 K_max D + 3 K_max
//     doubles; the kernel applies where that fits) and applies every update itself -- replicated arithmetic on
//     identical inputs, the same device functions as the launches, so nothing about the model ever crosses
//     workgroups and the results are the launches' bits;
//   * the one thing that is shared out is the span scores (log_marg_i of every candidate span: K_max x D terms
//     each, 105 spans per utterance at 20 landmarks): workgroup g scores the spans g, g + G, ... and publishes them
//     with atomics; ONE grid barrier per utterance; then every workgroup reads all of them and runs the same
//     forward filtering / backward sampling on the same uniforms (the stream's cursor is replicated too) and the
//     same sequence of assignments;
//   * workgroup 0 alone writes what leaves the kernel (labels, boundaries, token lists, totals; at the end the model).
//
// Labels: the utterance's rows are contiguous in X; their labels are staged in LDS (f.assignments addresses the
// staging array as if it began at row 0) and written back by workgroup 0 after the utterance.  A component that
// empties moves the last component into its slot (gaussian_components_*.py del_component): the utterance's own
// rows are relabelled in LDS at once, all workgroups leave after the utterance, the launcher relabels the other
// rows (k_fb_relabel) and starts the kernel again at the next utterance.
// ---------------------------------------------------------------------------------------
struct FbChainArgs {
    segk_corpus c;
    segk_fbgmm f;
    const int32_t *order;           // [dev] utterances of the sweep
    int q0, q1;                     // this launch walks order[q0 .. q1)
    const int32_t *row_start;       // [dev] [n_utt + 1] first row of every utterance
    int max_rows;                   // rows of an utterance at most
    int viterbi, map_assign, n_max;
    double wip, time_power_term, log_p_continue, anneal_fb, anneal_am;
    unsigned long long *score;      // [dev] [n_emb] span scores as bit patterns (the exchange between the workgroups)
    const double *ustream;
    int64_t *ucursor;
    int64_t ucap;
    uint8_t *boundaries;
    int32_t *new_tok, *n_new;
    double *out_logprob;
    int32_t *status;
    int32_t *ctl;                   // [0] barrier counter, [2] utterances completed, [3] error, [4] relabels logged, [5] utterance of
                                    // the relabels, [8 ..] the pairs, [32 (1 + f)] release words
    unsigned long long *stamp;      // development (make DEV=1, SEGK_CHAIN_STAMP=1): wall_clock64 of workgroup 0 at the phase boundaries
    const double *ktab;             // fb_diag_const by count (diagonal components), [ktab_n], or NULL
    int64_t ktab_n;
    const double *lprior_tab;       // [n_emb] log prior predictive of every row (k_fb_prior_tab), or NULL
    int64_t *lm_rep;                // language model: [gridDim.x][K_max^2] every workgroup's own copy of the bigram counts
    unsigned long long *ptab;       // [N_max (N_max + 1) / 2][K_max] the spans' predictive terms as bit patterns (exchange), or NULL
};

// lm.remove_counts_from_utterance / lm.counts_from_utterance (bigram_lms.py:98-114) over the transcript the boundaries define
// (sgn -1 / +1), by wave 0 of the workgroup on ITS copy of the counts: the components of the segments in order into `lmk`, then
// one lane per token.  As k_fbgmm_update op 5 / 6: a segment without embedding names row -1 -- python's last row, `last_asg` when
// that row is not the utterance's --, an unassigned row component -1 -- python's last count.  The bigram counts live in global
// memory and take atomic adds (a pair may occur twice in a transcript); the caller fences before anybody reads them.
static __device__ void fb_chain_lm_count(const int32_t *vid_l, const uint8_t *bnd_l, int N, const int32_t *asg_l, int64_t row0, int nrows,
                                         int64_t n_emb, int last_asg, int KM, int32_t *lmk, int64_t *lmu, int64_t *rep, long long sgn, int lane)
{
    const unsigned long long mask = __ballot(lane < N && bnd_l[lane] != 0);
    const bool bit = lane < N && ((mask >> lane) & 1ull);
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    if (bit) {
        const int jp = below ? 64 - __clzll((long long)below) : 0;
        int64_t id = vid_l[(lane + 1) * lane / 2 + jp];
        if (id < 0) id += n_emb;
        int k = (id >= row0 && id < row0 + nrows) ? asg_l[id - row0] : last_asg;
        if (k < 0) k += KM;
        lmk[__popcll(below)] = k;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = __popcll(mask);
    if (lane < n) {
        const int k = lmk[lane];
        atomicAdd((unsigned long long *)&lmu[k], (unsigned long long)sgn);
        if (lane > 0) atomicAdd((unsigned long long *)&rep[(int64_t)lmk[lane - 1] * KM + k], (unsigned long long)sgn);
    }
}
// development: accumulated ticks between sub-stamps of the assignment loop (row 256 of the stamp buffer: [p] += t_p - t_(p-1), [7] tokens)
#define FBC_SUB(p)                                                                                                         \
    do {                                                                                                                   \
        if (A.stamp && blockIdx.x == 0 && tid == 0) {                                                                      \
            const unsigned long long now_ = wall_clock64();                                                                \
            if ((p) > 0) A.stamp[256 * 8 + (p)] += now_ - fbc_sub_t;                                                       \
            else A.stamp[256 * 8 + 7] += 1;                                                                                \
            fbc_sub_t = now_;                                                                                              \
        }                                                                                                                  \
    } while (0)
#define FBC_STAMP(slot)                                                                                                   \
    do {                                                                                                                   \
        if (A.stamp && blockIdx.x == 0 && tid == 0 && q - A.q0 < 256) A.stamp[(q - A.q0) * 8 + (slot)] = wall_clock64();  \
    } while (0)

// COV, LM: the covariance type and the presence of a language model as compile-time constants (they are written into the
// kernel's private copy of the model below, and the device functions it is handed to are inlined): one kernel for every case
// was 18 400 instructions with every branch on f.cov_type / f.lm_unigram inside the per-segment loops.
template <typename XT, int COV, bool LM>
__global__ __launch_bounds__(512) void k_fb_chain(FbChainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fbc_lds[];
    const segk_corpus &c = A.c;
    const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;      // (nt = fb_nt(f): the launches' number of threads -- the
                                                                        // order of every block-wide sum depends on it)
    const int D = c.D, KM = A.f.K_max, NM = c.N_max;
    const int64_t KD = (int64_t)KM * D, triMax = (int64_t)NM * (NM + 1) / 2;
    // ---- LDS: the model, then the per-utterance buffers
    double *sa = reinterpret_cast<double *>(fbc_lds), *sb = sa + KD, *pp = sb + KD;
    double *lp = pp + KD, *kc = lp + KM;                                  // [K_max], [K_max + 1]
    int64_t *cn = reinterpret_cast<int64_t *>(kc + KM + 1);               // [K_max]
    double *z = reinterpret_cast<double *>(cn + KM);                      // [K_max]
    double *red = z + KM;                                                 // [nt]
    double *vec = red + nt;                                               // [triMax]
    double *al = vec + triMax, *ww = al + NM, *pr = ww + NM + 1;          // [N_max], [N_max + 1], [N_max + 1]
    XT *xrow = reinterpret_cast<XT *>(pr + NM + 1);                       // [D] (rounded to 8 bytes)
    int32_t *vid_l = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(xrow) + (((size_t)D * sizeof(XT) + 7) & ~(size_t)7));   // [triMax]
    int32_t *spans = vid_l + triMax;                                      // [triMax] the valid entries of vid_l, in order
    int32_t *asg_l = spans + triMax;                                      // [max_rows]
    int32_t *tok_l = asg_l + A.max_rows;                                  // [N_max]
    uint8_t *bnd_l = reinterpret_cast<uint8_t *>(tok_l + NM);             // [N_max] (rounded to 16 bytes)
    double *pri_l = reinterpret_cast<double *>(bnd_l + ((NM + 15) & ~15));    // [3][D] the prior's vectors
    XT *xs_l = reinterpret_cast<XT *>(pri_l + 3 * D);                     // [max_rows][D] the utterance's rows of X
    // language model (bigram_acoustic_wordseg.py:386-551): the unigram counts in LDS like the model, the bigram counts
    // (K_max^2: 80 KB at K = 100) in a copy of the workgroup's own in global memory -- replicated updates again, nothing shared
    constexpr bool lm = LM;
    // (aligned on 8 bytes whatever the buffers in front add up to: the counts take 64-bit LDS atomics, which -- unlike plain
    // loads and stores -- fault on a misaligned address)
    int64_t *lmu = reinterpret_cast<int64_t *>((reinterpret_cast<uintptr_t>(xs_l) + (size_t)A.max_rows * D * sizeof(XT) + 7) & ~(uintptr_t)7);   // [K_max]
    int32_t *lmk = reinterpret_cast<int32_t *>(lmu + KM);                 // [N_max]
    int64_t *rep = lm ? A.lm_rep + (int64_t)blockIdx.x * KM * KM : nullptr;
    // The predictive terms of the span scores, kept for the assignment (A.ptab != NULL).  Scoring a span evaluates, for every
    // component, the term the assignment of a segment made of that span needs again -- K_max x D logarithms (diagonal
    // covariances) that the workgroups REPLICATE per new segment: 7 of the 10 us a segment's assignment took.  The
    // workgroup that scores a span writes its row of terms; after the DP everybody reads the rows of the chosen spans
    // (pl), and fb_logits evaluates only the components that received a segment of this utterance since.
    const bool spec = A.ptab != nullptr;
    double *pz = reinterpret_cast<double *>(lm ? (reinterpret_cast<uintptr_t>(lmk + NM) + 7) & ~(uintptr_t)7 : reinterpret_cast<uintptr_t>(lmu));   // [K_max]
    double *pl = pz + KM;                                                 // [N_max][K_max] (with A.ptab; pz: always)
    int32_t *tokj_l = reinterpret_cast<int32_t *>(pl + (size_t)NM * KM);  // [N_max] triangular index of the new segments' spans
    uint8_t *modf = reinterpret_cast<uint8_t *>(tokj_l + NM);             // [K_max] component received a segment of this utterance
    // what the serial steps would otherwise fetch from global memory one dependent round trip (1.5-2 us) at a time: the
    // window of the uniform stream an utterance can consume (a draw per backward step and per new segment) and the log
    // prior predictive of its rows
    double *us_l = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(spec ? modf + KM : reinterpret_cast<uint8_t *>(pl)) + 7) & ~(uintptr_t)7);   // [2 N_max + 2]
    double *lpr_l = us_l + 2 * NM + 2;                                    // [max_rows]
    __shared__ int sh_last, sh_jprev;
    __shared__ int shK, ldsK, sh_i, sh_k, sh_flag, sh_nn, sh_nspan, n_relog;
    __shared__ int32_t relog[2 * FB_RELOG];
    __shared__ long long sh_cur;
    __shared__ double sh_total;
    for (int64_t i = tid; i < KD; i += nt) { sa[i] = A.f.stat_a[i]; sb[i] = A.f.stat_b[i]; pp[i] = A.f.pred[i]; }
    for (int i = tid; i < KM; i += nt) { lp[i] = A.f.log_prod[i]; kc[i] = A.f.kconst[i]; cn[i] = A.f.counts[i]; }
    if (tid == 0) { kc[KM] = A.f.kconst[KM]; shK = *A.f.K; ldsK = shK; sh_cur = *A.ucursor; }
    for (int d = tid; d < D; d += nt) { pri_l[d] = A.f.prior_a[d]; pri_l[D + d] = A.f.prior_b[d]; pri_l[2 * D + d] = A.f.prior_c[d]; }
    segk_fbgmm fl = A.f;
    fl.cov_type = COV;
    if (!LM) { fl.lm_unigram = nullptr; fl.lm_bigram = nullptr; }
    fl.stat_a = sa; fl.stat_b = sb; fl.pred = pp; fl.log_prod = lp; fl.kconst = kc; fl.counts = cn; fl.K = &ldsK;
    fl.prior_a = pri_l; fl.prior_b = pri_l + D; fl.prior_c = pri_l + 2 * D;
    if (lm) {
        for (int i = tid; i < KM; i += nt) lmu[i] = A.f.lm_unigram[i];
        for (int64_t i = tid; i < (int64_t)KM * KM; i += nt) rep[i] = A.f.lm_bigram[i];
        if (tid == 0) sh_last = A.f.assignments[c.n_emb - 1];
        fl.lm_unigram = lmu;
        fl.lm_bigram = rep;
    }
    __syncthreads();
    int phase = 0;
    unsigned long long fbc_sub_t = 0;
    for (int q = A.q0; q < A.q1; q++) {
        const int u = A.order[q];
        const int N = c.lengths[u], tri = N * (N + 1) / 2;
        const int64_t row0 = A.row_start[u];
        const int nrows = A.row_start[u + 1] - (int)row0;
        const FbSpanTab tab = fb_span_tab(c, u, N, A.n_max);
        // ---- (A) stage the utterance: labels of its rows, span table, old boundaries
        FBC_STAMP(0);
        for (int i = tid; i < nrows; i += nt) asg_l[i] = A.f.assignments[row0 + i];
        // (the rows themselves: every add_item / del_item / logits call read its row from global memory, a dependent round
        // trip each, two dozen per utterance)
        for (int i = tid; i < nrows * D; i += nt) {
            const int r = i / D, d = i - r * D;
            xs_l[i] = ((const XT *)c.X)[(row0 + r) * c.ldx + d];
        }
        if (tab.band) {                                  // (the band is complete: no embedding outside it)
            for (int j = tid; j < tri; j += nt) vid_l[j] = -1;
            __syncthreads();
            const int W = tab.W;
            for (int i = tid; i < N * W; i += nt) {
                const int t = i / W + 1, s = t - 1 - (i - (t - 1) * W);
                if (s >= 0) vid_l[t * (t - 1) / 2 + s] = tab.bandi[i];
            }
        } else {
            for (int j = tid; j < tri; j += nt) vid_l[j] = tab.vid[j];
        }
        for (int j = tid; j < N; j += nt) bnd_l[j] = A.boundaries[(int64_t)u * NM + j];
        if (tid == 0) n_relog = 0;
        if (spec)
            for (int k = tid; k < KM; k += nt) modf[k] = 0;
        const int64_t ucur0 = (int64_t)sh_cur;
        for (int i = tid; i < 2 * NM + 2; i += nt) us_l[i] = ucur0 + i < A.ucap ? A.ustream[ucur0 + i] : 0.5;
        if (A.lprior_tab)
            for (int i = tid; i < nrows; i += nt) lpr_l[i] = A.lprior_tab[row0 + i];
        // (always an LDS address: a pointer that is either this or NULL makes every access through it a flat one)
        const double *lprior_l = lpr_l - row0;
        __syncthreads();
        fl.assignments = asg_l - row0;                   // (only the utterance's rows are ever named)
        segk_corpus cl = c;                              // X as the device functions index it, backed by the staged rows
        cl.X = xs_l - row0 * D;
        cl.ldx = D;
        if (!A.lprior_tab) {                             // (no table from the host: D > 512) fb_logits' own expression, row by row
            for (int i = 0; i < nrows; i++) {
                const double lp_i = fb_prior_finish(fl, block_sum(fb_prior_sum<XT>(fl, D, xs_l + (int64_t)i * D, tid, nt), red));
                if (tid == 0) lpr_l[i] = lp_i;
            }
            __syncthreads();
        }
        const FbLocal loc{asg_l, row0, nrows, relog, &n_relog, A.ktab, A.ktab_n};
        if (tid < 64) {                                  // the valid spans, in table order (one wave: ballot + prefix count)
            int n = 0;
            for (int j0 = 0; j0 < tri; j0 += 64) {
                const int j = j0 + lane;
                const bool ok = j < tri && vid_l[j] >= 0;
                const unsigned long long m = __ballot(ok);
                if (ok) spans[n + __popcll(m & ((1ull << lane) - 1ull))] = j;
                n += __popcll(m);
            }
            if (lane == 0) sh_nspan = n;
        }
        // ---- (B) remove the utterance's old segments (unigram_acoustic_wordseg.py:270-273), replicated; with a language model
        // the counts of its transcript first (bigram_acoustic_wordseg.py:403-404)
        FBC_STAMP(1);
        if (lm) {
            __syncthreads();
            if (tid < 64) fb_chain_lm_count(vid_l, bnd_l, N, asg_l, row0, nrows, c.n_emb, sh_last, KM, lmk, lmu, rep, -1, lane);
            __threadfence();
            __syncthreads();
        }
        {
            int jp = 0;
            for (int j = 0; j < N; j++) {
                if (bnd_l[j]) {
                    const int id = vid_l[(j + 1) * j / 2 + jp];
                    jp = j + 1;
                    if (id >= 0) fb_del_item<XT>(cl, fl, id, &shK, &sh_i, red, &loc);
                }
            }
        }
        __syncthreads();
        if (tid == 0) ldsK = shK;
        __syncthreads();
        // ---- (C) this workgroup's share of the span scores: log_marg_i (fbgmm.py:256-285)
        FBC_STAMP(2);
        const int nspan = sh_nspan;
        const int K_C = ldsK;                            // active components while the spans are scored
        for (int s = blockIdx.x; s < nspan; s += gridDim.x) {
            const int64_t e = vid_l[spans[s]];
            fb_logits<XT>(cl, fl, e, lm ? 3 : 0, -1, xrow, z, red, lprior_l, pz);
            if (spec)
                for (int k = tid; k < K_C; k += nt)
                    __hip_atomic_store(&A.ptab[(int64_t)spans[s] * KM + k], (unsigned long long)__double_as_longlong(pz[k]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double mx = NEG_INF_D;
            for (int k = tid; k < KM; k += nt) mx = z[k] > mx ? z[k] : mx;
            mx = block_max(mx, red);
            double sm = 0.0;
            for (int k = tid; k < KM; k += nt) sm += exp(z[k] - mx);
            sm = block_sum(sm, red);
            if (tid == 0) atomicExch(&A.score[e], (unsigned long long)__double_as_longlong(log(sm) + mx));
        }
        FBC_STAMP(3);
        if (!chain_barrier(A.ctl, ++phase, &sh_flag)) return;
        FBC_STAMP(4);
        // ---- (D) vec (unigram_acoustic_wordseg.py:474-511) from everybody's scores, the DP by wave 0 (replicated)
        fb_fill_vec(tab, N, tri, [&](int id) {
            return __longlong_as_double((long long)__hip_atomic_load(&A.score[id], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }, A.time_power_term, A.wip, vec, tid, nt);
        __syncthreads();
        if (tid < 64) {
            StreamUniforms usrc = {us_l, (int64_t)sh_cur, A.ucap, A.status, ucur0};
            const double total = fb_dp_sample(vec, al, ww, pr, N, tri, A.n_max, A.viterbi, A.log_p_continue, A.anneal_fb, bnd_l, lane, usrc);
            const int nn = fb_collect_tokens_wave(vid_l, bnd_l, N, tok_l, lane, spec ? tokj_l : nullptr);
            if (lane == 0) {
                if (!A.viterbi && total == NEG_INF_D) atomicOr(A.status, 16);      // unigram_acoustic_wordseg.py:753
                sh_cur = usrc.cur;
                sh_total = total;
                sh_nn = nn;
            }
        }
        __syncthreads();
        // ---- (E) the new segments, in order (fbgmm.py:422-494), replicated
        FBC_STAMP(5);
        const int nn = sh_nn;
        if (tid == 0) sh_jprev = -1;
        if (spec)                                        // the chosen spans' rows of predictive terms, one round trip
            for (int i = tid; i < nn * K_C; i += nt) {
                const int t = i / K_C, k = i - t * K_C;
                pl[t * KM + k] = __longlong_as_double((long long)__hip_atomic_load(&A.ptab[(int64_t)tokj_l[t] * KM + k], __ATOMIC_RELAXED,
                                                                                    __HIP_MEMORY_SCOPE_AGENT));
            }
        for (int t = 0; t < nn; t++) {
            const int64_t e = tok_l[t];
            if (tid == 0) ldsK = shK;
            __syncthreads();
            FBC_SUB(0);
            const int j_prev = sh_jprev;
            // (with a language model: lm.log_prob_vec_i for the first segment, then given the one before, :482-494)
            fb_logits<XT>(cl, fl, e, lm ? (j_prev < 0 ? 3 : 4) : (A.map_assign ? 2 : 1), j_prev, xrow, z, red, lprior_l, nullptr,
                          pl + t * KM, spec ? K_C : 0, modf);      // (K_in = 0: every component evaluated, pl / modf not read)
            FBC_SUB(1);
            fb_draw_component(fl, z, red, A.map_assign, A.anneal_am, us_l, (int64_t *)&sh_cur, A.ucap, A.status, shK, &sh_k, ucur0);
            __syncthreads();
            FBC_SUB(2);
            fb_add_item<XT>(cl, fl, e, sh_k, &shK, &sh_i, red, &loc);
            if (tid == 0) {
                sh_jprev = sh_k;
                if (spec) modf[sh_k] = 1;
            }
            __syncthreads();
            FBC_SUB(3);
        }
        if (lm) {                                    // the counts of the new transcript (:546-547)
            // (the row python's -1 names, as the launches see it at this point: relabelled when a component moved)
            if (tid == 0) {
                int v = sh_last;
                for (int i = 0; i < n_relog && i < FB_RELOG; i++)
                    if (v == relog[2 * i]) v = relog[2 * i + 1];
                sh_last = v;
            }
            __syncthreads();
            if (tid < 64) fb_chain_lm_count(vid_l, bnd_l, N, asg_l, row0, nrows, c.n_emb, sh_last, KM, lmk, lmu, rep, 1, lane);
            __threadfence();
            __syncthreads();
            if (tid == 0 && c.n_emb - 1 >= row0 && c.n_emb - 1 < row0 + nrows) sh_last = asg_l[c.n_emb - 1 - row0];
        }
        // ---- (F) what leaves the kernel, by workgroup 0
        FBC_STAMP(6);
        bool stop = false;
        for (int i = 0; i < n_relog && i < FB_RELOG; i++) stop = stop || relog[2 * i] != relog[2 * i + 1];
        stop = stop || n_relog > FB_RELOG;
        if (blockIdx.x == 0) {
            for (int i = tid; i < nrows; i += nt) A.f.assignments[row0 + i] = asg_l[i];
            for (int j = tid; j < N; j += nt) A.boundaries[(int64_t)u * NM + j] = bnd_l[j];
            for (int t = tid; t < nn; t += nt) A.new_tok[(int64_t)u * NM + t] = tok_l[t];
            if (tid == 0) {
                A.n_new[u] = nn;
                A.out_logprob[u] = sh_total;
                A.ctl[2] = q + 1;
                if (stop) {
                    A.ctl[4] = n_relog;
                    A.ctl[5] = u;
                    for (int i = 0; i < 2 * FB_RELOG; i++) A.ctl[8 + i] = relog[i];
                }
            }
        }
        __syncthreads();
        if (stop) break;
    }
    // ---- the model and the stream's cursor back to memory
    if (blockIdx.x == 0) {
        for (int64_t i = tid; i < KD; i += nt) { A.f.stat_a[i] = sa[i]; A.f.stat_b[i] = sb[i]; A.f.pred[i] = pp[i]; }
        for (int i = tid; i < KM; i += nt) { A.f.log_prod[i] = lp[i]; A.f.kconst[i] = kc[i]; A.f.counts[i] = cn[i]; }
        if (tid == 0) { *A.f.K = shK; *A.ucursor = (int64_t)sh_cur; }
        if (lm) {
            for (int i = tid; i < KM; i += nt) A.f.lm_unigram[i] = lmu[i];
            for (int64_t i = tid; i < (int64_t)KM * KM; i += nt) A.f.lm_bigram[i] = rep[i];
        }
    }
}

// Fingerprint of everything the table of log prior predictives is computed from: the rows of X (as stored, every 32-bit word
// mixed with its index) and the prior's three vectors.  A sum, so the order of the threads does not matter.  The table used
// to be keyed by the ADDRESSES of X and of the prior: a second model built after the first was freed gets the same addresses
// from a caching allocator -- another corpus of the same shape then ran on the first one's table, silently.
__global__ __launch_bounds__(256) void k_fb_fingerprint(segk_corpus c, segk_fbgmm f, unsigned long long *out)
{
    const int64_t wpr = (int64_t)c.D * (c.x_dtype == SEGK_F32 ? 1 : 2);            // 32-bit words per row
    const int64_t ldw = (int64_t)c.ldx * (c.x_dtype == SEGK_F32 ? 1 : 2);
    const int64_t n = c.n_emb * wpr;
    const uint32_t *X = (const uint32_t *)c.X;
    unsigned long long h = 0ull;
    auto mix = [](unsigned long long v, unsigned long long i) -> unsigned long long {
        unsigned long long z = (v + 0x9E3779B97F4A7C15ull) ^ (i * 0xBF58476D1CE4E5B9ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / wpr, w = i - r * wpr;
        h += mix(X[r * ldw + w], (unsigned long long)i);
    }
    if (blockIdx.x == 0)
        for (int d = threadIdx.x; d < c.D; d += blockDim.x) {
            h += mix((unsigned long long)__double_as_longlong(f.prior_a[d]), 0x1000000000ull + d);
            h += mix((unsigned long long)__double_as_longlong(f.prior_b[d]), 0x2000000000ull + d);
            if (f.cov_type == 0) h += mix((unsigned long long)__double_as_longlong(f.prior_c[d]), 0x3000000000ull + d);
        }
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

// out[e] = log prior predictive of row e, by the expression (and the number of threads) fb_logits uses: one workgroup per row
template <typename XT>
__global__ __launch_bounds__(512) void k_fb_prior_tab(segk_corpus c, segk_fbgmm f, double *out)
{
    __shared__ double red[16];
    __shared__ XT xrow[512];
    const int64_t e = blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    for (int d = tid; d < D; d += nt) xrow[d] = ((const XT *)c.X)[e * c.ldx + d];
    __syncthreads();
    const double v = fb_prior_finish(f, block_sum(fb_prior_sum<XT>(f, D, xrow, tid, nt), red));
    if (tid == 0) out[e] = v;
}

__global__ void k_fb_kconst_tab(segk_fbgmm f, int D, int64_t n, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fb_diag_const(f, D, (double)i);
}

// the rows of every utterance but `skip_lo .. skip_hi` relabelled from -> to (a component moved into an emptied slot)
__global__ void k_fb_relabel(int32_t *assignments, int64_t n, int64_t skip_lo, int64_t skip_hi, int from, int to)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && (e < skip_lo || e >= skip_hi) && assignments[e] == from) assignments[e] = to;
}

// ---------------------------------------------------------------------------------------
// FBGMM.gibbs_sample inner loop (fbgmm.py:352-405) over the items ids[0..n) (NULL: rows 0..n) in
// order, one workgroup: cache the old component's statistics, del_item, logits, draw, then either
// restore the cached statistics (same component, no component deleted) or add_item.
// ---------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(512) void k_fbgmm_gibbs_items(segk_corpus c, segk_fbgmm f, const int32_t *ids, int64_t n,
                                    int consider_unassigned, double anneal_temp, const double *ustream,
                                    int64_t *ucursor, int64_t ucap, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *z = (double *)smem;                      // [K_max]
    double *red = z + f.K_max;                       // [nt]
    double *cache = red + blockDim.x;                // [3 D]
    XT *xrow = (XT *)(cache + 3 * c.D);              // [D]
    __shared__ int shK, sh_i, sh_k, sh_kold;
    __shared__ double sh_lp, sh_kc;
    __shared__ long long sh_cnt;
    const int D = c.D, tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) shK = *f.K;
    __syncthreads();
    for (int64_t t = 0; t < n; t++) {
        const int64_t e = ids ? (int64_t)ids[t] : t;
        if (tid == 0) sh_kold = f.assignments[e];
        __syncthreads();
        const int k_old = sh_kold;
        __syncthreads();                                         // sh_kold is rewritten next iteration
        if (!consider_unassigned && k_old == -1) continue;
        const int K_old = shK;
        const int kc = k_old < 0 ? k_old + f.K_max : k_old;      // python row -1 for an unassigned item
        for (int d = tid; d < D; d += nt) {                      // cache_component_stats
            cache[d] = f.stat_a[(int64_t)kc * D + d];
            cache[D + d] = f.stat_b[(int64_t)kc * D + d];
            cache[2 * D + d] = f.pred[(int64_t)kc * D + d];
        }
        if (tid == 0) { sh_lp = f.log_prod[kc]; sh_kc = f.kconst[kc]; sh_cnt = f.counts[kc]; }
        __syncthreads();
        fb_del_item<XT>(c, f, e, &shK, &sh_i, red);
        __syncthreads();
        if (tid == 0) *f.K = shK;
        __syncthreads();
        fb_logits<XT>(c, f, e, 1, -1, xrow, z, red);
        fb_draw_component(f, z, red, 0, anneal_temp, ustream, ucursor, ucap, status, shK, &sh_k);
        __syncthreads();
        if (sh_k == k_old && shK == K_old) {                     // restore_component_from_stats (:397-400)
            for (int d = tid; d < D; d += nt) {
                f.stat_a[(int64_t)kc * D + d] = cache[d];
                f.stat_b[(int64_t)kc * D + d] = cache[D + d];
                f.pred[(int64_t)kc * D + d] = cache[2 * D + d];
            }
            if (tid == 0) {
                f.log_prod[kc] = sh_lp;
                f.kconst[kc] = sh_kc;
                f.counts[kc] = sh_cnt;
                f.assignments[e] = k_old;
            }
            __syncthreads();
        } else {
            fb_add_item<XT>(c, f, e, sh_k, &shK, &sh_i, red);
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) *f.K = shK;
}

// ---------------------------------------------------------------------------------------
// Components __init__ (fixedvar:110-120 / diag:114-120): add_item(i, k) for k ascending and i
// ascending within k.  One wave per component scans `assignments`; statistics are accumulated
// in exactly that order (including the repeated `+= precision`), derived values once at the end.
// ---------------------------------------------------------------------------------------
template <typename XT>
__global__ void k_fbgmm_init_stats(segk_corpus c, segk_fbgmm f, const int32_t *blk_lo, int n_blocks, const int32_t *sorted,
                                   const int32_t *koff)
{
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (k > f.K_max) return;
    const int D = c.D;
    if (k == f.K_max) {         // x-independent part of the prior predictive (fixedvar:224-231, diag:215-222)
        double part = 0.0;
        for (int d = lane; d < D; d += 64)
            part += f.cov_type == 0 ? log(f.prior_c[d]) : log((f.k_0 + 1.) / (f.k_0 * f.v_0) * f.prior_a[d]);
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0)
            f.kconst[k] = f.cov_type == 0 ? -0.5 * (double)D * LOG_2PI + 0.5 * part : fb_diag_const(f, D, 0.0) - 0.5 * part;
        return;
    }
    const XT *X = (const XT *)c.X;
    int64_t cnt = 0;
    double lp_part = 0.0;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        double a = 0.0, b = 0.0;
        if (d < D) {
            if (f.cov_type == 0) { a = f.prior_c[d] * f.prior_b[d]; b = f.prior_c[d]; }
            else { a = f.k_0 * f.prior_b[d]; b = f.prior_a[d] + f.k_0 * (f.prior_b[d] * f.prior_b[d]); }
        }
        int64_t n = 0;
        // the component's rows in ascending order, from the bucketed row lists (segk_rows_by_label: before, every
        // component scanned the whole assignment vector -- K_max x n_emb ballots, 6 ms at a million rows); eight rows
        // are fetched together, the additions stay one row after the other
        for (int bb = 0; bb < n_blocks; bb++) {
            const int32_t *ko = koff + (int64_t)bb * (f.K_max + 1);
            const int64_t p0 = blk_lo[bb];
            const int q0 = ko[k], q1 = ko[k + 1];
            for (int qb = q0; qb < q1; qb += 8) {
                XT xv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int64_t ee = p0 + sorted[p0 + (qb + q < q1 ? qb + q : q0)];       // clamped: always valid
                    xv[q] = X[ee * c.ldx + (d < D ? d : 0)];
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (qb + q < q1) {
                        n++;
                        if (d < D) {
                            const double x = (double)xv[q];
                            if (f.cov_type == 0) { a += f.prior_a[d] * x; b += f.prior_a[d]; }
                            else { a += x; b += x_sq<XT>(xv[q]); }
                        }
                    }
            }
        }
        cnt = n;
        if (d < D) {
            double pr = 0.0;
            if (n > 0) {
                if (f.cov_type == 0) {
                    pr = b * f.prior_a[d] / (b + f.prior_a[d]);
                    lp_part += log(pr);
                } else {
                    const double k_N = f.k_0 + (double)n, v_N = f.v_0 + (double)n;
                    double m_N = a / k_N;
                    double var = (k_N + 1.) / (k_N * v_N) * (b - k_N * (m_N * m_N));
                    pr = 1. / var;
                    lp_part += log(var);
                }
            } else { a = 0.0; b = 0.0; }
            f.stat_a[(int64_t)k * D + d] = a;
            f.stat_b[(int64_t)k * D + d] = b;
            f.pred[(int64_t)k * D + d] = pr;
        }
    }
    for (int o = 32; o > 0; o >>= 1) lp_part += __shfl_xor(lp_part, o);
    if (lane == 0) {
        f.counts[k] = cnt;
        f.log_prod[k] = cnt ? lp_part : 0.0;
        f.kconst[k] = !cnt ? 0.0 : (f.cov_type == 0 ? -0.5 * (double)D * LOG_2PI : fb_diag_const(f, D, (double)cnt));
        if (cnt) atomicMax(f.K, k + 1);
    }
}

// ======================================================================================
// C ABI
// ======================================================================================
#define DISPATCH_XT(c, ...)                         \
    do {                                            \
        if ((c)->x_dtype == SEGK_F32) {             \
            typedef float XT;                       \
            __VA_ARGS__                             \
        } else {                                    \
            typedef double XT;                      \
            __VA_ARGS__                             \
        }                                           \
    } while (0)

static int check_fb(const segk_corpus *c, const segk_fbgmm *f)
{
    SEGK_REQUIRE(c && f, "NULL corpus / fbgmm");
    SEGK_REQUIRE(f->cov_type == 0 || f->cov_type == 1, "cov_type must be 0 (fixed) or 1 (diag)");
    SEGK_REQUIRE(f->K_max > 0 && c->D > 0, "sizes");
    SEGK_REQUIRE(f->kconst != NULL, "kconst buffer missing");
    return SEGK_OK;
}

// workgroup width of the logits kernels: small component banks get several lanes per component
static int fb_nt(const segk_fbgmm *f) { return f->K_max >= 256 ? 256 : 512; }

static size_t fb_lds(const segk_corpus *c, const segk_fbgmm *f, int nt)
{
    return (size_t)(f->K_max + nt) * sizeof(double) + (size_t)((c->D + 1) & ~1) * (c->x_dtype == SEGK_F32 ? 4 : 8);
}

extern "C" {

int32_t segk_fbgmm_update(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, int32_t op, int32_t utt,
                          int64_t item, int32_t k, const uint8_t *boundaries, void *stream)
{
    (void)ctx;
    int rc = check_fb(c, f);
    if (rc) return rc;
    SEGK_REQUIRE(op == 0 || op == 1 || op == 2 || op == 4 || op == 5 || op == 6, "op");
    DISPATCH_XT(c, hipLaunchKernelGGL(k_fbgmm_update<XT>, dim3(1), dim3(fb_nt(f)), 0, (hipStream_t)stream, *c, *f, op, utt,
                                       item, k, boundaries););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbgmm_init_stats(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = check_fb(c, f);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(f->K, 0, sizeof(int32_t), st));
    const int32_t *blk_lo = nullptr, *sorted = nullptr, *koff = nullptr;
    int n_blocks = 0;
    rc = segk_rows_by_label(ctx, f->assignments, c->n_emb, f->K_max, &blk_lo, &n_blocks, &sorted, &koff, stream);
    if (rc) return rc;
    int64_t grid = ((int64_t)f->K_max + 1 + 3) / 4;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_fbgmm_init_stats<XT>, dim3((unsigned)grid), dim3(256), 0, st, *c, *f, blk_lo, n_blocks,
                                       sorted, koff););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbgmm_score(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const int32_t *ids,
                         int64_t row0, int64_t n, double *out, void *stream)
{
    (void)ctx;
    int rc = check_fb(c, f);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    const int nt = fb_nt(f);
    size_t lds = fb_lds(c, f, nt);
    SEGK_REQUIRE(lds <= 160 * 1024, "K_max too large for the LDS logits buffer");
    DISPATCH_XT(c, {
        if (lds > 48 * 1024)
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbgmm_score<XT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fbgmm_score<XT>, dim3((unsigned)n), dim3(nt), lds, (hipStream_t)stream, *c, *f, ids, row0,
                           out);
    });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbgmm_pred_vector(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, int64_t row,
                               double *out, void *stream)
{
    (void)ctx;
    int rc = check_fb(c, f);
    if (rc) return rc;
    SEGK_REQUIRE(row >= 0 && row < c->n_emb, "row out of range");
    DISPATCH_XT(c, hipLaunchKernelGGL(k_fbgmm_pred_vector<XT>, dim3((f->K_max + 1 + 127) / 128), dim3(128), 0,
                                       (hipStream_t)stream, *c, *f, row, out););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_unigram_segment(segk_ctx *ctx, const segk_corpus *c, int32_t utt, int32_t viterbi,
                             int32_t n_slices_min, int32_t n_slices_max, double wip, double time_power_term,
                             double log_p_continue, double anneal_temp, const double *score,
                             const double *ustream, int64_t *ucursor, int64_t ucap, uint8_t *boundaries,
                             int32_t *new_tok, int32_t *n_new, double *out_logprob, int32_t *status, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(c && utt >= 0 && utt < c->n_utt, "utterance");
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1, "n_slices_min must be 0 or 1");
    const int64_t triMax = (int64_t)c->N_max * (c->N_max + 1) / 2;
    size_t lds = (size_t)(triMax + 3 * c->N_max + 2) * sizeof(double);
    SEGK_REQUIRE(lds <= 160 * 1024, "N_max too large for the LDS score vector");
    if (lds > 48 * 1024)
        SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_unigram_segment, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds));
    hipLaunchKernelGGL(k_unigram_segment, dim3(1), dim3(128), lds, (hipStream_t)stream, *c, utt, viterbi, n_slices_min,
                       n_slices_max, wip, time_power_term, log_p_continue, anneal_temp, score, ustream, ucursor, ucap,
                       boundaries, new_tok, n_new, out_logprob, status);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbgmm_assign(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, int32_t utt, int32_t map_assign,
                          int32_t j_prev, double anneal_temp, const int32_t *new_tok, const int32_t *n_new,
                          const double *ustream, int64_t *ucursor, int64_t ucap, int32_t *status, void *stream)
{
    (void)ctx;
    int rc = check_fb(c, f);
    if (rc) return rc;
    const int nt = fb_nt(f);
    size_t lds = fb_lds(c, f, nt);
    SEGK_REQUIRE(lds <= 160 * 1024, "K_max too large for the LDS logits buffer");
    DISPATCH_XT(c, {
        if (lds > 48 * 1024)
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbgmm_assign<XT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fbgmm_assign<XT>, dim3(1), dim3(nt), lds, (hipStream_t)stream, *c, *f, utt, map_assign,
                           j_prev, anneal_temp, new_tok, n_new, ustream, ucursor, ucap, status);
    });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// gibbs_sample_i (unigram_acoustic_wordseg.py:252-360) for every utterance of `order` in turn by the persistent kernel
// k_fb_chain.  SEGK_ERR_UNSUPPORTED (nothing enqueued) where it does not apply: a language model, a model that does
// not fit a workgroup's LDS, more than 64 landmarks, an utterance listed twice, assignments_only.  The call SYNCHRONISES the
// stream after every launch (it reads how far the kernel got: a launch ends early when a component empties).
int32_t segk_fbgmm_sequential_sweep(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, const int32_t *order, int32_t n_order,
                                    const int32_t *row_start, int32_t viterbi, int32_t map_assign, int32_t n_slices_min,
                                    int32_t n_slices_max, double wip, double time_power_term, double log_p_continue,
                                    double anneal_temp_fb, double anneal_temp_am, double *score, const double *ustream,
                                    int64_t *ucursor, int64_t ucap, uint8_t *boundaries, int32_t *new_tok, int32_t *n_new,
                                    double *out_logprob, int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = check_fb(c, f);
    if (rc) return rc;
    SEGK_REQUIRE(order && n_order >= 0 && row_start && score && ustream && ucursor && boundaries && new_tok && n_new && out_logprob && status,
                 "sequential sweep operands");
    SEGK_REQUIRE(c->vec_ids && c->durations && c->lengths && c->n_utt > 0, "corpus without utterances");
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1, "n_slices_min must be 0 or 1");
    SEGK_REQUIRE(viterbi == 0 || viterbi == 1, "viterbi");
    if (n_order == 0) return SEGK_OK;
    const char *env = getenv("SEGK_FB_CHAIN");
    if (env && atoi(env) == 0) { segk_set_error("segk_fbgmm_sequential_sweep: disabled (SEGK_FB_CHAIN=0)"); return SEGK_ERR_UNSUPPORTED; }
    if (c->N_max > 64 || ctx->capturing) {
        segk_set_error("segk_fbgmm_sequential_sweep: at most 64 landmarks, not under stream capture");
        return SEGK_ERR_UNSUPPORTED;
    }
    // rows of an utterance (host copy of row_start: the caller passes the device array; the extent comes from the order's
    // utterances, read back once per corpus would do -- it is 4 (n_utt + 1) bytes)
    std::vector<int32_t> rs((size_t)c->n_utt + 1);
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemcpyAsync(rs.data(), row_start, rs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // (under the same synchronisation) the fingerprint of the rows and the prior, for the table of log prior predictives below
    unsigned long long fp = 0ull;
    if (c->D <= 512) {
        if (!ctx->fb_fp_dev) SEGK_CHECK_HIP(hipMalloc((void **)&ctx->fb_fp_dev, sizeof(unsigned long long)));
        SEGK_CHECK_HIP(hipMemsetAsync(ctx->fb_fp_dev, 0, sizeof(unsigned long long), st));
        const int64_t words = c->n_emb * c->D * (c->x_dtype == SEGK_F32 ? 1 : 2);
        int64_t grid = (words + 256 * 16 - 1) / (256 * 16);
        if (grid > 4 * ctx->n_cu) grid = 4 * ctx->n_cu;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(k_fb_fingerprint, dim3((unsigned)grid), dim3(256), 0, st, *c, *f, ctx->fb_fp_dev);
        SEGK_CHECK_HIP(hipMemcpyAsync(&fp, ctx->fb_fp_dev, sizeof(fp), hipMemcpyDeviceToHost, st));
    }
    SEGK_CHECK_HIP(hipStreamSynchronize(st));
    int max_rows = 1;
    {
        std::vector<uint8_t> seen((size_t)c->n_utt, 0);
        for (int32_t q = 0; q < n_order; q++) {
            SEGK_REQUIRE(order[q] >= 0 && order[q] < c->n_utt, "utterance index out of range");
            if (seen[order[q]]) { segk_set_error("segk_fbgmm_sequential_sweep: an utterance is listed twice"); return SEGK_ERR_UNSUPPORTED; }
            seen[order[q]] = 1;
            const int nr = rs[order[q] + 1] - rs[order[q]];
            SEGK_REQUIRE(nr >= 0 && rs[order[q] + 1] <= c->n_emb, "row_start");
            if (nr > max_rows) max_rows = nr;
        }
    }
    const int KM = f->K_max, D = c->D, NM = c->N_max;
    const int64_t KD = (int64_t)KM * D, triMax = (int64_t)NM * (NM + 1) / 2;
    const size_t xb = ((size_t)D * (c->x_dtype == SEGK_F32 ? 4 : 8) + 7) & ~(size_t)7;
    const int nt = fb_nt(f);
    const size_t lds = (size_t)(3 * KD + 3 * KM + 1 + KM + nt + triMax + 3 * NM + 2 + 3 * D) * sizeof(double) + xb +
                       (size_t)(2 * triMax + max_rows + NM) * sizeof(int32_t) + (size_t)((NM + 15) & ~15) +
                       (size_t)max_rows * D * (c->x_dtype == SEGK_F32 ? 4 : 8) + 16 +
                       (f->lm_unigram ? (size_t)KM * sizeof(int64_t) + (size_t)NM * sizeof(int32_t) + 16 : 0) +
                       (size_t)(2 * NM + 2 + max_rows + KM) * sizeof(double) + 16;  // the staged uniforms and log prior predictives, pz
    if (lds > 150 * 1024) {
        segk_set_error("segk_fbgmm_sequential_sweep: the model (%d components x %d dimensions) does not fit a workgroup's LDS", KM, D);
        return SEGK_ERR_UNSUPPORTED;
    }
    // the spans' predictive terms kept for the assignment (k_fb_chain, `spec`): [K_max] + [N_max][K_max] doubles, the segments'
    // span indices and the components' flags -- when they fit beside the model (SEGK_FB_CHAIN_TERMS=0: evaluate every
    // component per segment as before; the same results)
    const size_t lds_terms = (size_t)((size_t)NM * KM) * sizeof(double) + (size_t)NM * sizeof(int32_t) + (((size_t)KM + 15) & ~(size_t)15) + 16;
    const char *et = getenv("SEGK_FB_CHAIN_TERMS");
    const bool terms = lds + lds_terms <= 150 * 1024 && !(et && atoi(et) == 0);
    // control words + the order, owned by the context
    const size_t ctl_bytes = 32 * (1 + CH_FLAGS) * sizeof(int32_t), need = ctl_bytes + (size_t)n_order * sizeof(int32_t);
    if (ctx->fbchain_bytes < need) {
        if (ctx->fbchain_buf) (void)hipFree(ctx->fbchain_buf);
        ctx->fbchain_buf = nullptr;
        ctx->fbchain_bytes = 0;
        SEGK_CHECK_HIP(hipMalloc(&ctx->fbchain_buf, need));
        ctx->fbchain_bytes = need;
    }
    unsigned char *buf = (unsigned char *)ctx->fbchain_buf;
    FbChainArgs A{};
    A.c = *c; A.f = *f;
    A.ctl = (int32_t *)buf;
    A.order = (const int32_t *)(buf + ctl_bytes);
    A.row_start = row_start; A.max_rows = max_rows;
    A.viterbi = viterbi; A.map_assign = map_assign; A.n_max = n_slices_max;
    A.wip = wip; A.time_power_term = time_power_term; A.log_p_continue = log_p_continue;
    A.anneal_fb = anneal_temp_fb; A.anneal_am = anneal_temp_am;
    A.score = (unsigned long long *)score;
    A.ustream = ustream; A.ucursor = ucursor; A.ucap = ucap;
    A.boundaries = boundaries; A.new_tok = new_tok; A.n_new = n_new; A.out_logprob = out_logprob; A.status = status;
    SEGK_CHECK_HIP(hipMemcpyAsync((void *)A.order, order, (size_t)n_order * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const bool stamping = segk_dev_env("SEGK_CHAIN_STAMP") != 0;      // make DEV=1 builds only
    static unsigned long long *stamp_dev = nullptr;
    if (stamping && !stamp_dev) SEGK_CHECK_HIP(hipMalloc((void **)&stamp_dev, 257 * 8 * sizeof(unsigned long long)));
    A.stamp = stamping ? stamp_dev : nullptr;
    // diagonal components: fb_diag_const by count, once per (prior, corpus size)
    A.ktab = nullptr;
    A.ktab_n = 0;
    if (f->cov_type == 1) {
        const int64_t n = c->n_emb + 2;
        if (!ctx->fb_ktab || ctx->fb_ktab_n < n || ctx->fb_ktab_v0 != f->v_0 || ctx->fb_ktab_D != D) {
            if (ctx->fb_ktab) (void)hipFree(ctx->fb_ktab);
            ctx->fb_ktab = nullptr;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->fb_ktab, (size_t)n * sizeof(double)));
            hipLaunchKernelGGL(k_fb_kconst_tab, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *f, D, n, ctx->fb_ktab);
            ctx->fb_ktab_n = n;
            ctx->fb_ktab_v0 = f->v_0;
            ctx->fb_ktab_D = D;
        }
        A.ktab = ctx->fb_ktab;
        A.ktab_n = ctx->fb_ktab_n;
    }
    // the rows' log prior predictive, once per (corpus, prior, number of threads)
    A.lprior_tab = nullptr;
    if (D <= 512) {
        if (!ctx->fb_ptab || ctx->fb_ptab_n < c->n_emb || ctx->fb_ptab_rows != c->n_emb || ctx->fb_ptab_fp != fp || ctx->fb_ptab_D != c->D ||
            ctx->fb_ptab_nt != nt || ctx->fb_ptab_cov != f->cov_type || ctx->fb_ptab_k0 != f->k_0 || ctx->fb_ptab_v0 != f->v_0) {
            if (ctx->fb_ptab_n < c->n_emb) {
                if (ctx->fb_ptab) (void)hipFree(ctx->fb_ptab);
                ctx->fb_ptab = nullptr;
                ctx->fb_ptab_n = 0;
                SEGK_CHECK_HIP(hipMalloc((void **)&ctx->fb_ptab, (size_t)c->n_emb * sizeof(double)));
                ctx->fb_ptab_n = c->n_emb;
            }
            DISPATCH_XT(c, hipLaunchKernelGGL(k_fb_prior_tab<XT>, dim3((unsigned)c->n_emb), dim3(nt), 0, st, *c, *f, ctx->fb_ptab););
            SEGK_LAUNCH_CHECK();
            ctx->fb_ptab_fp = fp; ctx->fb_ptab_rows = c->n_emb; ctx->fb_ptab_D = c->D; ctx->fb_ptab_nt = nt; ctx->fb_ptab_cov = f->cov_type;
            ctx->fb_ptab_k0 = f->k_0; ctx->fb_ptab_v0 = f->v_0;
        }
        A.lprior_tab = ctx->fb_ptab;
    }
    // one workgroup per span of an utterance at most (105 at 20 landmarks and a window of six), never more than fit together
    int G = ctx->n_cu < 128 ? ctx->n_cu : 128;
    {
        const int64_t spans = (int64_t)max_rows < triMax ? max_rows : triMax;
        if (spans < G) G = (int)(spans > 0 ? spans : 1);
    }
    A.lm_rep = nullptr;
    if (f->lm_unigram) {                               // every workgroup's copy of the bigram counts
        const size_t nb = (size_t)G * KM * KM * sizeof(int64_t);
        if (ctx->fbchain_lm_bytes < nb) {
            if (ctx->fbchain_lm) (void)hipFree(ctx->fbchain_lm);
            ctx->fbchain_lm = nullptr;
            ctx->fbchain_lm_bytes = 0;
            SEGK_CHECK_HIP(hipMalloc(&ctx->fbchain_lm, nb));
            ctx->fbchain_lm_bytes = nb;
        }
        A.lm_rep = (int64_t *)ctx->fbchain_lm;
    }
    A.ptab = nullptr;
    if (terms) {
        const size_t nb = (size_t)triMax * KM * sizeof(unsigned long long);
        if (ctx->fbchain_terms_bytes < nb) {
            if (ctx->fbchain_terms) (void)hipFree(ctx->fbchain_terms);
            ctx->fbchain_terms = nullptr;
            ctx->fbchain_terms_bytes = 0;
            SEGK_CHECK_HIP(hipMalloc(&ctx->fbchain_terms, nb));
            ctx->fbchain_terms_bytes = nb;
        }
        A.ptab = (unsigned long long *)ctx->fbchain_terms;
    }
    const size_t lds_all = lds + (terms ? lds_terms : 0);
    int q = 0;
    while (q < n_order) {
        SEGK_CHECK_HIP(hipMemsetAsync(buf, 0, ctl_bytes, st));
        if (stamping) SEGK_CHECK_HIP(hipMemsetAsync(stamp_dev + 256 * 8, 0, 8 * sizeof(unsigned long long), st));
        A.q0 = q; A.q1 = n_order;
#define FB_CHAIN_LAUNCH(COV, LM)                                                                      \
    DISPATCH_XT(c, {                                                                                  \
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_fb_chain<XT, COV, LM>, lds_all));                 \
        hipLaunchKernelGGL((k_fb_chain<XT, COV, LM>), dim3(G), dim3(nt), lds_all, st, A);             \
    })
        if (f->cov_type == 0) {
            if (f->lm_unigram) FB_CHAIN_LAUNCH(0, true); else FB_CHAIN_LAUNCH(0, false);
        } else {
            if (f->lm_unigram) FB_CHAIN_LAUNCH(1, true); else FB_CHAIN_LAUNCH(1, false);
        }
#undef FB_CHAIN_LAUNCH
        SEGK_LAUNCH_CHECK();
        int32_t ctl[8 + 2 * FB_RELOG];
        SEGK_CHECK_HIP(hipMemcpyAsync(ctl, A.ctl, sizeof(ctl), hipMemcpyDeviceToHost, st));
        SEGK_CHECK_HIP(hipStreamSynchronize(st));
        if (ctl[3] != 0) {
            segk_set_error("FBGMM chain: grid barrier timed out (%d workgroups were not resident together?)", G);
            return SEGK_ERR_HIP;
        }
        if (ctl[2] <= q) {
            segk_set_error("FBGMM chain: no progress at utterance %d", q);
            return SEGK_ERR_HIP;
        }
        if (stamping && ctl[2] - q >= 120) {       // a long launch: mean time of the phases, in 10 ns ticks of the 100 MHz clock
            static unsigned long long hs[257 * 8];
            SEGK_CHECK_HIP(hipMemcpy(hs, stamp_dev, sizeof(hs), hipMemcpyDeviceToHost));
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int n0 = 10, n1 = 110;
            for (int i = n0; i < n1; i++) {
                for (int j = 0; j < 6; j++) acc[j] += (double)(hs[i * 8 + j + 1] - hs[i * 8 + j]);
                acc[6] += (double)(hs[(i + 1) * 8] - hs[i * 8 + 6]);
                acc[7] += (double)(hs[(i + 1) * 8] - hs[i * 8]);
            }
            const double dn = 100.0 * (n1 - n0);
            fprintf(stderr, "fb chain stamps (us): stage %.2f  remove %.2f  scores %.2f  barrier %.2f  vec+dp %.2f  assign %.2f  write %.2f | per utterance %.2f\n",
                    acc[0] / dn, acc[1] / dn, acc[2] / dn, acc[3] / dn, acc[4] / dn, acc[5] / dn, acc[6] / dn, acc[7] / dn);
            const double ntok = (double)hs[256 * 8 + 7] > 0 ? (double)hs[256 * 8 + 7] : 1.0;
            fprintf(stderr, "  per new segment (us, %.0f segments over the launch): logits %.2f  draw %.2f  add_item %.2f\n", ntok,
                    (double)hs[256 * 8 + 1] / 100.0 / ntok, (double)hs[256 * 8 + 2] / 100.0 / ntok, (double)hs[256 * 8 + 3] / 100.0 / ntok);
        }
        q = ctl[2];
        if (ctl[4] > 0) {                 // components emptied during utterance ctl[5]: the other utterances' rows follow
            SEGK_REQUIRE(ctl[4] <= FB_RELOG, "FBGMM chain: more components emptied within one utterance than the log holds");
            const int u = ctl[5];
            for (int i = 0; i < ctl[4]; i++) {
                const int from = ctl[8 + 2 * i], to = ctl[8 + 2 * i + 1];
                if (from != to)
                    hipLaunchKernelGGL(k_fb_relabel, dim3((unsigned)((c->n_emb + 255) / 256)), dim3(256), 0, st, f->assignments,
                                       c->n_emb, (int64_t)rs[u], (int64_t)rs[u + 1], from, to);
            }
            SEGK_LAUNCH_CHECK();
        }
    }
    return SEGK_OK;
}

int32_t segk_fbgmm_gibbs_items(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, const int32_t *ids, int64_t n,
                               int32_t consider_unassigned, double anneal_temp, const double *ustream,
                               int64_t *ucursor, int64_t ucap, int32_t *status, void *stream)
{
    (void)ctx;
    int rc = check_fb(c, f);
    if (rc) return rc;
    SEGK_REQUIRE(f->lm_unigram == NULL, "FBGMM.gibbs_sample has no language-model variant in the reference");
    SEGK_REQUIRE(ids != NULL || n <= c->n_emb, "n exceeds the number of rows");
    if (n <= 0) return SEGK_OK;
    const int nt = fb_nt(f);
    size_t lds = fb_lds(c, f, nt) + (size_t)3 * c->D * sizeof(double);
    SEGK_REQUIRE(lds <= 160 * 1024, "K_max too large for the LDS logits buffer");
    DISPATCH_XT(c, {
        if (lds > 48 * 1024)
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbgmm_gibbs_items<XT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fbgmm_gibbs_items<XT>, dim3(1), dim3(nt), lds, (hipStream_t)stream, *c, *f, ids, n,
                           consider_unassigned, anneal_temp, ustream, ucursor, ucap, status);
    });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

}  // extern "C"
