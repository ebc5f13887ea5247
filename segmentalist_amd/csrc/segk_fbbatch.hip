// segk_fbbatch.hip -- batch-synchronous ("blocked parallel Gibbs") sweep of the FBGMM / bigram
// word-segmentation samplers on gfx950.  The reference has no parallel mode; the sampler is
// specified in oracle/np_fbgmm_batch.py and built from the reference's pieces: log_marg_i
// (fbgmm.py:256-285), the predictive densities (gaussian_components_fixedvar.py:224-253,
// gaussian_components_diag.py:215-259), forward filtering / backward sampling
// (unigram_acoustic_wordseg.py:653-756), utils.draw (utils.py:10-21), the bigram LM prior
// (bigram_lms.py:64-91).
//
// One Gibbs step b of a sweep (host: device.py FbgmmBatchSweeper):
//   k_fbb_prepare   statistics of all tokens outside block b from the per-(block, slice) partial
//                   sums, in the fixed order of the specification, and the per-slot predictive
//                   parameters derived from them (transposed [D, K_max] for coalesced slot lanes)
//   k_fbb_score     log_marg_i of every candidate span of the block: R rows x all slots per
//                   workgroup, online logsumexp per row                         (fp64 VALU)
//   k_fbb_segment   one workgroup per utterance: score vector, DP, backward sampling with the
//                   counter-based uniforms, new token list
//   k_fbb_assign    one workgroup per utterance: logits -> softmax -> draw for each new token
//                   (with a language model: the bigram prior of the previous token's slot)
//   k_fbb_partials  the block's partial sums from its new tokens (token order per slot)
// Everything is fp64; the 1e-4 contract of the path would allow fp32 matrix arithmetic for the
// fixed-variance score -- left for a later round.
#include <stdlib.h>

#include <type_traits>
#include "segk_fb_common.h"

#define FBB_R 8            // rows per workgroup of the score kernel
#define FBA_R 16           // token rows of the assignment kernel's LDS image (chunks of 16 tokens with four row groups, else <= FBB_R)
#define FBB_MAXCH 4        // dimension chunks of 64 lanes held in registers (D <= 256)

static __device__ __forceinline__ int64_t fbb_rec(const segk_fbgmm &f, int D) { return (int64_t)f.K_max * (2 * D + 1); }

template <typename XT>
static __device__ __forceinline__ double fbb_sq(XT x)      // np.square(X) in the dtype of X (diag:125)
{
    XT q = x * x;
    return (double)q;
}

// which (slice, local index) a workgroup works on: `off` is the prefix of the per-slice counts
struct FbbMap {
    int n;
    int lo[16];
    int off[17];
};

// Column maps of the packed fp16x2 image (written by k_fbb_rows16), behind the constants: [K_max] column of slot k (-1: empty), [K_max] the
// pseudo-component's (= the number of occupied slots), [K_max + 1] tiles in use, then [K_max + 1] the slot of column c.  The
// token-likelihood matrix has the image's columns.
static __host__ __device__ __forceinline__ int32_t *fbb_cmap(const segk_fbatch *bt, int KM)
{
    return reinterpret_cast<int32_t *>(bt->consts16 + KM + 2);
}

static __device__ __forceinline__ bool fbb_locate(const FbbMap &m, int wg, int *slice, int *idx)
{
    for (int s = 0; s < m.n; s++)
        if (wg < m.off[s + 1]) {
            *slice = s;
            *idx = wg - m.off[s];
            return true;
        }
    return false;
}

// ---------------------------------------------------------------------------------------
// partial sums of (slice s, block b) from the current token lists: one wave per (s, slot)
// ---------------------------------------------------------------------------------------
template <typename XT>
__global__ void k_fbb_partials(segk_corpus c, segk_fbgmm f, segk_fbatch bt, int s_lo, int s_n, int b,
                               const int32_t *new_tok, const int32_t *n_new)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    // (the step's totals -- read by its score and assignment kernels, rebuilt with atomics by the next k_fbb_prepare -- are
    // cleared here: a memset of 16 bytes in front of every prepare was a 4.5 us launch)
    if (wave == 0 && lane < 2) bt.scal[lane] = 0.0;
    if (wave >= s_n * f.K_max) return;
    const int s = s_lo + wave / f.K_max, k = wave % f.K_max;
    const int D = c.D;
    const XT *X = (const XT *)c.X;
    const int u0 = bt.utt_range[(s * bt.n_blocks + b) * 2], u1 = bt.utt_range[(s * bt.n_blocks + b) * 2 + 1];
    double ax[FBB_MAXCH], axx[FBB_MAXCH];
#pragma unroll
    for (int q = 0; q < FBB_MAXCH; q++) { ax[q] = 0.0; axx[q] = 0.0; }
    double n = 0.0;
    // four utterances per trip, stage by stage (token counts, tokens, slots: three round trips for four utterances instead of
    // three per utterance); the matches are then taken in utterance and token order as before
    for (int ub = u0; ub < u1; ub += 4) {
        int nn4[4], id4[4], match4[4];
#pragma unroll
        for (int v = 0; v < 4; v++) nn4[v] = ub + v < u1 ? n_new[ub + v] : 0;
#pragma unroll
        for (int v = 0; v < 4; v++) id4[v] = lane < nn4[v] ? new_tok[(int64_t)(ub + v) * c.N_max + lane] : -1;
#pragma unroll
        for (int v = 0; v < 4; v++) match4[v] = id4[v] >= 0 ? (bt.slot[id4[v]] == k) : 0;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            unsigned long long bal = __ballot(match4[v]);
            while (bal) {                                   // token order
                const int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                const int64_t e = __shfl(id4[v], src);
                n += 1.0;
#pragma unroll
                for (int q = 0; q < FBB_MAXCH; q++) {
                    const int d = q * 64 + lane;
                    if (d < D) {
                        const XT x = X[e * c.ldx + d];
                        ax[q] += (double)x;
                        axx[q] += fbb_sq<XT>(x);
                    }
                }
            }
        }
    }
    double *rec = bt.partials + ((int64_t)b * bt.n_slices + s) * fbb_rec(f, D);
    if (lane == 0) rec[k] = n;
#pragma unroll
    for (int q = 0; q < FBB_MAXCH; q++) {
        const int d = q * 64 + lane;
        if (d < D) {
            rec[f.K_max + (int64_t)k * D + d] = ax[q];
            rec[f.K_max + (int64_t)f.K_max * D + (int64_t)k * D + d] = axx[q];
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same in two steps for banks of many slots.  Above, every (slice, slot) wave walks ALL utterances of the block --
// two dependent loads per utterance, 1 250 utterances, 1 000 waves: 115 us per Gibbs step of bigram_c5.  Here one
// workgroup per slice first buckets the block's tokens by slot -- a stable counting sort: per-wave histograms in LDS, the
// waves own contiguous runs of token positions, the rank of a token among the equal keys of its 64 comes from ballots --
// and the (slice, slot) waves then sum their own lists, in token order as before: the same additions in the same order.
// ---------------------------------------------------------------------------------------
#define FBS_THREADS 1024
#define FBS_WAVES 16
__global__ __launch_bounds__(FBS_THREADS) void k_fbb_sort(segk_corpus c, segk_fbgmm f, segk_fbatch bt, int s_lo, int b,
                                                        const int32_t *new_tok, const int32_t *n_new, int32_t *sorted,
                                                        int64_t sorted_stride, int32_t *koff)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KM = f.K_max, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t *cntw = (int32_t *)smem;                 // [FBS_WAVES][KM]
    int32_t *wsum = cntw + FBS_WAVES * KM;           // [FBS_WAVES + 1]
    const int s = s_lo + blockIdx.x;
    const int u0 = bt.utt_range[(s * bt.n_blocks + b) * 2], u1 = bt.utt_range[(s * bt.n_blocks + b) * 2 + 1];
    const int S = (u1 - u0) * c.N_max;               // token positions (utterance-major), most of them unused
    const int per = ((S + FBS_WAVES - 1) / FBS_WAVES + 63) & ~63;
    int32_t *out = sorted + (int64_t)blockIdx.x * sorted_stride;
    int32_t *ko = koff + (int64_t)blockIdx.x * (KM + 1);
    int nbits = 1;
    while ((1 << nbits) < KM) nbits++;
    for (int i = tid; i < FBS_WAVES * KM; i += FBS_THREADS) cntw[i] = 0;
    __syncthreads();
    auto key_of = [&](int p, int32_t *id_out) -> int {
        int key = -1;
        *id_out = -1;
        if (p < S) {
            const int u = u0 + p / c.N_max, j = p % c.N_max;
            if (j < n_new[u]) {
                const int32_t id = new_tok[(int64_t)u * c.N_max + j];
                *id_out = id;
                key = bt.slot[id];
            }
        }
        return key;
    };
    const int p_lo = wv * per, p_hi = p_lo + per < S ? p_lo + per : S;
    // (keys of four chunks of 64 positions at a time, stage by stage -- token count, token, slot: three round trips for 256
    // positions; one chunk after the other was three DEPENDENT round trips per chunk, twelve for a wave's run, in both passes)
    auto keys4 = [&](int p0, int key[4], int32_t id[4]) {
        int u_[4], j_[4], nn_[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = p0 + 64 * q + lane < p_hi ? p0 + 64 * q + lane : S;
            u_[q] = p < S ? u0 + p / c.N_max : -1;
            j_[q] = p < S ? p % c.N_max : 0;
            nn_[q] = u_[q] >= 0 ? n_new[u_[q]] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) id[q] = j_[q] < nn_[q] ? new_tok[(int64_t)u_[q] * c.N_max + j_[q]] : -1;
#pragma unroll
        for (int q = 0; q < 4; q++) key[q] = id[q] >= 0 ? bt.slot[id[q]] : -1;
    };
    (void)key_of;
    for (int p0 = p_lo; p0 < p_hi; p0 += 256) {
        int key[4];
        int32_t id[4];
        keys4(p0, key, id);
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (key[q] >= 0) atomicAdd(&cntw[wv * KM + key[q]], 1);
    }
    __syncthreads();
    // per slot: the waves' counts -> running offsets inside the slot; the slots' totals -> an exclusive scan over the slots
    int tot = 0;
    if (tid < KM)
        for (int w = 0; w < FBS_WAVES; w++) {
            const int v = cntw[w * KM + tid];
            cntw[w * KM + tid] = tot;
            tot += v;
        }
    int incl = tot;                                 // inclusive scan over the threads (slots tid = 0 .. KM - 1; KM <= 1024)
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < FBS_WAVES; w++) { const int v = wsum[w]; wsum[w] = run; run += v; }
        wsum[FBS_WAVES] = run;
    }
    __syncthreads();
    const int base = wsum[wv] + incl - tot;         // first position of slot tid
    if (tid < KM) {
        ko[tid] = base;
        for (int w = 0; w < FBS_WAVES; w++) cntw[w * KM + tid] += base;
    }
    if (tid == 0) ko[KM] = wsum[FBS_WAVES];
    __syncthreads();
    // placement: every wave walks its run again, 64 positions at a time in order
    for (int p0 = p_lo; p0 < p_hi; p0 += 256) {
        int key4[4];
        int32_t id4[4];
        keys4(p0, key4, id4);
#pragma unroll
        for (int q = 0; q < 4; q++) {               // the chunks of 64 in order
            const int key = key4[q];
            const int32_t id = id4[q];
            const bool ok = key >= 0;
            unsigned long long mask = __ballot(ok);
            if (mask == 0ull) continue;
            for (int bit = 0; bit < nbits; bit++) {
                const unsigned long long bal = __ballot((key >> bit) & 1);
                mask &= ((key >> bit) & 1) ? bal : ~bal;
            }
            if (ok) {
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                const int o = cntw[wv * KM + key];
                out[o + rank] = id;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (ok && (mask & ((1ull << lane) - 1ull)) == 0ull) cntw[wv * KM + key] += __popcll(mask);      // the first lane of every key
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename XT>
__global__ void k_fbb_partials_sorted(segk_corpus c, segk_fbgmm f, segk_fbatch bt, int s_lo, int s_n, int b,
                                      const int32_t *sorted, int64_t sorted_stride, const int32_t *koff)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    // (the step's totals -- read by its score and assignment kernels, rebuilt with atomics by the next k_fbb_prepare -- are
    // cleared here: a memset of 16 bytes in front of every prepare was a 4.5 us launch)
    if (wave == 0 && lane < 2) bt.scal[lane] = 0.0;
    if (wave >= s_n * f.K_max) return;
    const int si = wave / f.K_max, s = s_lo + si, k = wave % f.K_max;
    const int D = c.D;
    const XT *X = (const XT *)c.X;
    const int32_t *list = sorted + (int64_t)si * sorted_stride;
    const int q0 = koff[(int64_t)si * (f.K_max + 1) + k], q1 = koff[(int64_t)si * (f.K_max + 1) + k + 1];
    double ax[FBB_MAXCH], axx[FBB_MAXCH];
#pragma unroll
    for (int q = 0; q < FBB_MAXCH; q++) { ax[q] = 0.0; axx[q] = 0.0; }
    // token order.  The list entries of up to 64 tokens by one load, the rows of four tokens in flight together (an entry and
    // then its row per token were two dependent round trips each: 30 us for three tokens per slot on average); the additions
    // stay sequential
    for (int t0 = q0; t0 < q1; t0 += 64) {
        const int cnt = q1 - t0 < 64 ? q1 - t0 : 64;
        const int my_e = lane < cnt ? list[t0 + lane] : 0;
        for (int u0 = 0; u0 < cnt; u0 += 4) {
            XT xv[4][FBB_MAXCH];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t e = __builtin_amdgcn_readlane(my_e, u0 + u < cnt ? u0 + u : cnt - 1);
#pragma unroll
                for (int q = 0; q < FBB_MAXCH; q++) {
                    const int d = q * 64 + lane;
                    xv[u][q] = d < D ? X[e * c.ldx + d] : (XT)0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (u0 + u < cnt) {
#pragma unroll
                    for (int q = 0; q < FBB_MAXCH; q++) {
                        const int d = q * 64 + lane;
                        if (d < D) {
                            ax[q] += (double)xv[u][q];
                            axx[q] += fbb_sq<XT>(xv[u][q]);
                        }
                    }
                }
        }
    }
    double *rec = bt.partials + ((int64_t)b * bt.n_slices + s) * fbb_rec(f, D);
    if (lane == 0) rec[k] = (double)(q1 - q0);
#pragma unroll
    for (int q = 0; q < FBB_MAXCH; q++) {
        const int d = q * 64 + lane;
        if (d < D) {
            rec[f.K_max + (int64_t)k * D + d] = ax[q];
            rec[f.K_max + (int64_t)f.K_max * D + (int64_t)k * D + d] = axx[q];
        }
    }
}

// the pairing of oracle tree_sum: [(0+1), (2+3), ...], odd element carried
static __device__ __forceinline__ double fbb_tree(double *p, int n, int stride)
{
    while (n > 1) {
        const int h = n >> 1;
        for (int j = 0; j < h; j++) p[j * stride] = p[2 * j * stride] + p[(2 * j + 1) * stride];
        if (n & 1) p[h * stride] = p[(n - 1) * stride];
        n = (n + 1) >> 1;
    }
    return p[0];
}

// ---------------------------------------------------------------------------------------
// statistics without block b (b = -1: all) and the per-slot predictive parameters.
// One workgroup of three waves per slot: wave 0 sums the counts, wave 1 the sums, wave 2 the sums of squares (their loads
// are one memory round trip side by side; as one wave per slot they were three in a row -- the kernel is K_max workgroups of
// pure latency: 36 us with a load per loop iteration, 28 with eight in flight, 21 with a quantity's 64 in flight), lanes over
// the dimensions; wave 0 then forms the parameters.  Shared scratch: [waves][16 slices][64 lanes].
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void k_fbb_prepare(segk_fbgmm f, segk_fbatch bt, int D, int b, double prior_alpha)
{
    __shared__ double scr[3][16][64];
    __shared__ double res[3][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x;
    const int S = bt.n_slices, B = bt.n_blocks, KM = f.K_max;
    const int64_t rec = fbb_rec(f, D);
    if (k == 0 && threadIdx.x == 0 && bt.consts16) bt.consts16[KM + 1] = 0.0;      // max |row|^2 of the step's fp16x2 image (k_fbb_rows16)
    double (*my)[64] = scr[w];
    // my[s][lane] = sum over the blocks bp != b of partials[(bp * S + s) * rec + off], in block order, for every slice s: the
    // loads of eight slices x eight blocks are in flight together
    auto sum_slices = [&](int64_t off) {
        for (int s0 = 0; s0 < S; s0 += 8) {
            double a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = 0.0;
            for (int bp0 = 0; bp0 < B; bp0 += 8) {
                double v[8][8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int sidx = s0 + i < S ? s0 + i : S - 1;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int bp = bp0 + j < B ? bp0 + j : B - 1;
                        v[i][j] = bt.partials[((int64_t)bp * S + sidx) * rec + off];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++)
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (bp0 + j < B && bp0 + j != b) a[i] += v[i][j];
            }
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (s0 + i < S) my[s0 + i][lane] = a[i];
        }
    };
    double lsum = 0.0, n = 0.0, k_N = 0.0, v_N = 0.0;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        if (w == 0) {
            if (d0 == 0) {
                sum_slices(k);
                res[0][lane] = fbb_tree(&my[0][lane], S, 64);
            }
        } else if (d < D && (w == 1 || f.cov_type == 1)) {
            sum_slices(w == 1 ? KM + (int64_t)k * D + d : KM + (int64_t)KM * D + (int64_t)k * D + d);
            res[w][lane] = fbb_tree(&my[0][lane], S, 64);
        }
        __syncthreads();
        if (w == 0) {
            if (d0 == 0) {
                n = res[0][lane];
                k_N = f.k_0 + n;
                v_N = f.v_0 + n;
            }
            if (d < D) {
                const double sx = res[1][lane], sxx = f.cov_type == 1 ? res[2][lane] : 0.0;
                double mean, q, lt;
                if (f.cov_type == 0) {          // fixedvar:153-170, 317-325
                    const double pN = f.prior_c[d] + n * f.prior_a[d];
                    mean = (f.prior_c[d] * f.prior_b[d] + f.prior_a[d] * sx) / pN;
                    q = pN * f.prior_a[d] / (pN + f.prior_a[d]);
                    lt = log(q);
                } else {                        // diag:162-177, 332-345
                    mean = (f.k_0 * f.prior_b[d] + sx) / k_N;
                    const double var = (k_N + 1.) / (k_N * v_N)
                                       * (f.prior_a[d] + f.k_0 * (f.prior_b[d] * f.prior_b[d]) + sxx - k_N * (mean * mean));
                    q = 1. / var * (1. / v_N);
                    lt = log(var);
                }
                bt.mean_t[(int64_t)d * KM + k] = mean;
                bt.q_t[(int64_t)d * KM + k] = q;
                lsum += lt;
            }
        }
        if (d0 + 64 < D) __syncthreads();           // (res is written again)
    }
    if (w != 0) return;
    lsum = fb_wave_sum(lsum);
    // the slot's scalars: the two log-gammas side by side on lanes 0 and 1, the two logarithms likewise
    const double lg = f.cov_type == 0 ? 0.0 : lgamma(lane == 0 ? (v_N + 1.) / 2. : v_N / 2.);
    const double lv = log(lane == 0 ? v_N : prior_alpha / (double)KM + n);
    const double lg1 = __shfl(lg, 1), lv1 = __shfl(lv, 1);
    if (lane == 0) {
        double lconst;
        if (f.cov_type == 0) lconst = -0.5 * (double)D * 1.8378770664093453 + 0.5 * lsum;
        else lconst = (double)D * (lg - lg1 - 0.5 * lv - 0.5 * 1.1447298858494002) - 0.5 * lsum;
        bt.cnt[k] = n;
        bt.lconst[k] = lconst;
        bt.zconst[k] = f.lms * lv1 + (n > 0.0 ? lconst : 0.0);
        bt.half[k] = f.cov_type == 0 ? 0.5 : (v_N + 1.) / 2.;
        atomicAdd(&bt.scal[0], n);                       // integer valued: exact in any order
        if (n > 0.0) atomicAdd(&bt.scal[1], 1.0);
    }
}

// acc[r] += sum_d term(mean[d, k], q[d, k], xs[r][d]) for the FBB_R rows in LDS; the parameters of
// four dimensions are fetched ahead of their use (the loads are independent, the compiler keeps
// them in flight together), dimensions in increasing order.
template <int COV>
static __device__ __forceinline__ void fbb_accumulate(const segk_fbatch &bt, int KM, int k, int D, const double *xs,
                                                      double *acc)
{
    int d = 0;
    for (; d + 4 <= D; d += 4) {
        double m[4], q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m[j] = bt.mean_t[(int64_t)(d + j) * KM + k];
            q[j] = bt.q_t[(int64_t)(d + j) * KM + k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < FBB_R; r++) {
                const double delta = m[j] - xs[r * D + d + j];
                if (COV == 0) acc[r] += (delta * delta) * q[j];
                else acc[r] += log(1. + (delta * delta) * q[j]);
            }
        }
    }
    for (; d < D; d++) {
        const double m = bt.mean_t[(int64_t)d * KM + k], q = bt.q_t[(int64_t)d * KM + k];
#pragma unroll
        for (int r = 0; r < FBB_R; r++) {
            const double delta = m - xs[r * D + d];
            if (COV == 0) acc[r] += (delta * delta) * q;
            else acc[r] += log(1. + (delta * delta) * q);
        }
    }
}

// the same for NR rows starting at xs (a row group of k_fbb_assign): per (row, slot) the identical sequence of operations
template <int COV, int NR>
static __device__ __forceinline__ void fbb_accumulate_rows(const segk_fbatch &bt, int KM, int k, int D, const double *xs, double *acc)
{
    int d = 0;
    for (; d + 4 <= D; d += 4) {
        double m[4], q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m[j] = bt.mean_t[(int64_t)(d + j) * KM + k];
            q[j] = bt.q_t[(int64_t)(d + j) * KM + k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const double delta = m[j] - xs[r * D + d + j];
                if (COV == 0) acc[r] += (delta * delta) * q[j];
                else acc[r] += log(1. + (delta * delta) * q[j]);
            }
        }
    }
    for (; d < D; d++) {
        const double m = bt.mean_t[(int64_t)d * KM + k], q = bt.q_t[(int64_t)d * KM + k];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const double delta = m - xs[r * D + d];
            if (COV == 0) acc[r] += (delta * delta) * q;
            else acc[r] += log(1. + (delta * delta) * q);
        }
    }
}

// the diagonal (Student-t) terms of NR rows in float32 with the hardware logarithm (log2; the caller multiplies by ln 2):
// the arithmetic of k_fbb_score_diag32, for the token likelihoods of the assignment step (`score_precision="f32"`)
template <int NR>
static __device__ __forceinline__ void fbb_accumulate_rows32(const segk_fbatch &bt, int KM, int k, int D, const double *xs, float *acc)
{
    int d = 0;
    // sixteen dimensions' parameters in flight (four at a time were ten dependent round trips for D = 39); the terms in the
    // same order
    for (; d + 16 <= D; d += 16) {
        float m[16], q[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            m[j] = (float)bt.mean_t[(int64_t)(d + j) * KM + k];
            q[j] = (float)bt.q_t[(int64_t)(d + j) * KM + k];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float delta = m[j] - (float)xs[r * D + d + j];
                acc[r] += __builtin_amdgcn_logf(1.f + (delta * delta) * q[j]);
            }
        }
    }
    for (; d + 4 <= D; d += 4) {
        float m[4], q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m[j] = (float)bt.mean_t[(int64_t)(d + j) * KM + k];
            q[j] = (float)bt.q_t[(int64_t)(d + j) * KM + k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float delta = m[j] - (float)xs[r * D + d + j];
                acc[r] += __builtin_amdgcn_logf(1.f + (delta * delta) * q[j]);
            }
        }
    }
    for (; d < D; d++) {
        const float m = (float)bt.mean_t[(int64_t)d * KM + k], q = (float)bt.q_t[(int64_t)d * KM + k];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const float delta = m - (float)xs[r * D + d];
            acc[r] += __builtin_amdgcn_logf(1.f + (delta * delta) * q);
        }
    }
}

// x-dependent part of the prior predictive of one row, by one wave (lanes over d); result in all lanes
template <typename XT>
static __device__ double fbb_prior_row(const segk_fbgmm &f, int D, const double *x, int lane)
{
    double s = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double delta = x[d] - f.prior_b[d];
        if (f.cov_type == 0) s += delta * delta * f.prior_c[d];
        else {
            const double var = (f.k_0 + 1.) / (f.k_0 * f.v_0) * f.prior_a[d];
            s += log(1. + 1. / f.v_0 * (delta * delta) * (1. / var));
        }
    }
    s = fb_wave_sum(s);
    return f.cov_type == 0 ? f.kconst[f.K_max] - 0.5 * s : f.kconst[f.K_max] - (f.v_0 + 1.) / 2. * s;
}

// the same for a row in memory in the dtype of X (k_fbb_prior_rows, D > 256)
template <typename XT>
static __device__ double fbb_prior_row_mem(const segk_fbgmm &f, int D, const XT *x, int lane)
{
    double s = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double delta = (double)x[d] - f.prior_b[d];
        if (f.cov_type == 0) s += delta * delta * f.prior_c[d];
        else {
            const double var = (f.k_0 + 1.) / (f.k_0 * f.v_0) * f.prior_a[d];
            s += log(1. + 1. / f.v_0 * (delta * delta) * (1. / var));
        }
    }
    s = fb_wave_sum(s);
    return f.cov_type == 0 ? f.kconst[f.K_max] - 0.5 * s : f.kconst[f.K_max] - (f.v_0 + 1.) / 2. * s;
}

// ---------------------------------------------------------------------------------------
// the prior predictive of every embedding row, one wave per row (segk_fbb_prior_rows): the value the kernels below compute
// for themselves when segk_fbatch.prior_rows is NULL -- same function, same lanes, same order
template <typename XT>
__global__ __launch_bounds__(256) void k_fbb_prior_rows(segk_corpus c, segk_fbgmm f, double *out)
{
    __shared__ double xs[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, D = c.D;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= c.n_emb) return;
    const XT *X = (const XT *)c.X;
    // (rows of more than 256 dimensions: straight from memory -- fbb_prior_row reads x[d] for d = lane, lane + 64, ...)
    double v;
    if (D <= 256) {
        for (int d = lane; d < D; d += 64) xs[w][d] = (double)X[row * c.ldx + d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        v = fbb_prior_row<XT>(f, D, xs[w], lane);
    } else {
        v = fbb_prior_row_mem<XT>(f, D, X + row * c.ldx, lane);
    }
    if (lane == 0) out[row] = v;
}

// score of FBB_R rows against all slots per workgroup; online logsumexp per row and thread,
// merged across the workgroup at the end.
// ---------------------------------------------------------------------------------------
template <typename XT, int COV>
__global__ __launch_bounds__(256) void k_fbb_score(segk_corpus c, segk_fbgmm f, segk_fbatch bt, FbbMap map, int b, double prior_alpha,
                            double *score)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = c.D, KM = f.K_max, tid = threadIdx.x, nt = blockDim.x;
    double *xs = (double *)smem;                 // [R][D]
    double *lpr = xs + FBB_R * D;                // [R]
    double *red = lpr + FBB_R;                   // [16]
    int s, idx;
    if (!fbb_locate(map, blockIdx.x, &s, &idx)) return;
    const int slice = map.lo[s];
    const int64_t r_lo = bt.row_range[(slice * bt.n_blocks + b) * 2], r_hi = bt.row_range[(slice * bt.n_blocks + b) * 2 + 1];
    const int64_t row0 = r_lo + (int64_t)idx * FBB_R;
    const int nr = (int)((r_hi - row0) < FBB_R ? (r_hi - row0) : FBB_R);
    const XT *X = (const XT *)c.X;
    for (int j = tid; j < FBB_R * D; j += nt) {
        const int r = j / D, d = j - r * D;
        xs[j] = r < nr ? (double)X[(row0 + r) * c.ldx + d] : 0.0;
    }
    __syncthreads();
    {
        const int w = tid >> 6, lane = tid & 63, nw = nt >> 6;
        for (int r = w; r < FBB_R; r += nw) {
            const double v = (bt.prior_rows && r < nr) ? bt.prior_rows[row0 + r] : fbb_prior_row<XT>(f, D, xs + r * D, lane);
            if (lane == 0) lpr[r] = v;
        }
    }
    __syncthreads();
    const double zc_empty = f.lms * log(prior_alpha / (double)KM);
    double mx[FBB_R], sm[FBB_R];
#pragma unroll
    for (int r = 0; r < FBB_R; r++) { mx[r] = NEG_INF_D; sm[r] = 0.0; }
    for (int k = tid; k < KM; k += nt) {
        double z[FBB_R];
        if (bt.cnt[k] > 0.0) {
            double acc[FBB_R];
#pragma unroll
            for (int r = 0; r < FBB_R; r++) acc[r] = 0.0;
            fbb_accumulate<COV>(bt, KM, k, D, xs, acc);
            const double zc = bt.zconst[k], h = bt.half[k];
#pragma unroll
            for (int r = 0; r < FBB_R; r++) z[r] = zc - h * acc[r];
        } else {
#pragma unroll
            for (int r = 0; r < FBB_R; r++) z[r] = zc_empty + lpr[r];
        }
#pragma unroll
        for (int r = 0; r < FBB_R; r++) {
            if (z[r] > mx[r]) {
                sm[r] = sm[r] * exp(mx[r] - z[r]) + 1.0;       // exp(-inf) = 0 on the first value
                mx[r] = z[r];
            } else {
                sm[r] += exp(z[r] - mx[r]);
            }
        }
    }
    const double norm = f.lms * log(bt.scal[0] + prior_alpha);
#pragma unroll
    for (int r = 0; r < FBB_R; r++) {
        const double M = block_max(mx[r], red);
        const double part = mx[r] == NEG_INF_D ? 0.0 : sm[r] * exp(mx[r] - M);
        const double S = block_sum(part, red);
        if (tid == 0 && r < nr) score[row0 + r] = log(S) + M - norm;
    }
}

// ---------------------------------------------------------------------------------------
// The diagonal (Student-t) span score in float32 (`score_precision="f32"`, opt-in): the kernel above spends its life in
// the fp64 software logarithm of every (row, slot, dimension) term -- one log per term, K_max * D terms per row.  Here
// the term log(1 + delta^2 q) is four float32 instructions, the logarithm the hardware's v_log_f32 (base 2; ln 2 is
// folded into the slot's factor), the terms of a (row, slot) accumulate in float32, z = zc - h ln2 acc is formed in
// fp64 and the online log-sum-exp over the slots runs in float32 in base 2 (v_exp_f32) relative to the running
// maximum.  Error budget against the fp64 kernel (tests/test_gpu_fbgmm_batch.py measures it): |log2(1 + t)| carries
// ~2^-23 relative, D terms, times h ln2 <= (v_0 + N + 1)/2 -- observed <= 2e-6 relative to max(|log_marg_i|, 1),
// contract 1e-4.  FBB_R32 rows per workgroup: the slot tables are read once for 16 rows.
// ---------------------------------------------------------------------------------------
#define FBB_R32 16
// RPG rows per thread (16: one thread group over the slots; 8: two groups when K_max <= 128 would leave most of the 256
// threads idle).  TLDS: the slot tables (mean, q) as float32 in LDS, staged once per workgroup with coalesced loads -- read
// from memory inside the term loop (TLDS = false, tables beyond the LDS budget) every four dimensions cost a dependent
// round trip, which is where the first version of this kernel spent its time (9 % of what the ALUs sustain on the term).
template <typename XT, int RPG, bool TLDS>
__global__ __launch_bounds__(256) void k_fbb_score_diag32(segk_corpus c, segk_fbgmm f, segk_fbatch bt, FbbMap map, int b,
                                                          double prior_alpha, double *score, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NG = FBB_R32 / RPG;            // thread groups
    const int D = c.D, KM = f.K_max, tid = threadIdx.x, nt = blockDim.x;
    double *xs64 = (double *)smem;               // [8][D] scratch of the prior predictive (fp64, 8 rows at a time)
    double *lpr = xs64 + 8 * D;                  // [R32]
    float *wm = (float *)(lpr + FBB_R32);        // [4 waves][R32] the waves' maxima
    float *ws = wm + 4 * FBB_R32;                // [4 waves][R32] and sums
    float *xs = ws + 4 * FBB_R32;                // [D][R32]: the rows of a thread at one dimension are consecutive
    float *tm = xs + FBB_R32 * D;                // [D][KM] (TLDS)
    float *tq = tm + (TLDS ? D * KM : 0);        // [D][KM]
    int s, idx;
    if (!fbb_locate(map, blockIdx.x, &s, &idx)) return;
    const int slice = map.lo[s];
    if (dbg & 32) return;
    // (the workgroup is a chain of dependent round trips -- 15 of its 30 us with the table fill and the terms switched off: what
    // does not depend on another load is requested here, together with the first one)
    const int gsz = nt / NG, kk = tid % gsz, r0 = (tid / gsz) * RPG;       // this thread's slots kk, kk + gsz, ...; rows r0 .. r0 + RPG
    const int k_first = kk < KM ? kk : 0;
    const double cnt_first = bt.cnt[k_first], zc_first = bt.zconst[k_first], half_first = bt.half[k_first], total = bt.scal[0];
    const int64_t r_lo = bt.row_range[(slice * bt.n_blocks + b) * 2], r_hi = bt.row_range[(slice * bt.n_blocks + b) * 2 + 1];
    const int64_t row0 = r_lo + (int64_t)idx * FBB_R32;
    const int nr = (int)((r_hi - row0) < FBB_R32 ? (r_hi - row0) : FBB_R32);
    const XT *X = (const XT *)c.X;
    for (int j = tid; j < FBB_R32 * D; j += nt) {
        const int r = j / D, d = j - r * D;
        xs[d * FBB_R32 + r] = r < nr ? (float)X[(row0 + r) * c.ldx + d] : 0.f;
    }
    if (TLDS && !(dbg & 1))                       // (dbg, make DEV=1 only: 1 no table fill, 2 no terms, 4 no reductions -- timing)
        for (int j = tid; j < D * KM; j += nt) {
            tm[j] = (float)bt.mean_t[j];
            tq[j] = (float)bt.q_t[j];
        }
    // the prior predictive of every row (an empty slot's likelihood): fp64 as in the fp64 kernel, once per row -- from
    // segk_fbb_prior_rows' table when the caller keeps one (the values are constants of the corpus: 16 x D software
    // logarithms and four barriers per workgroup and Gibbs step otherwise)
    if (bt.prior_rows) {
        for (int r = tid; r < FBB_R32; r += nt) lpr[r] = r < nr ? bt.prior_rows[row0 + r] : 0.0;
    } else
    for (int r8 = 0; r8 < FBB_R32; r8 += 8) {
        __syncthreads();
        for (int j = tid; j < 8 * D; j += nt) {
            const int r = r8 + j / D, d = j % D;
            xs64[j] = r < nr ? (double)X[(row0 + r) * c.ldx + d] : 0.0;
        }
        __syncthreads();
        const int w = tid >> 6, lane = tid & 63, nw = nt >> 6;
        for (int r = w; r < 8; r += nw) {
            const double v = fbb_prior_row<XT>(f, D, xs64 + r * D, lane);
            if (lane == 0) lpr[r8 + r] = v;
        }
    }
    __syncthreads();
    if (dbg & 16) return;
    // (v_log_f32 for the two constants as for the terms: ~1e-7 relative, two fp64 library logarithms were 2 us of every workgroup)
    const double zc_empty = f.lms * fb_log_fast(prior_alpha / (double)KM);
    const double norm = f.lms * fb_log_fast(total + prior_alpha);
    // The term log2(1 + (m - x)^2 q) on PAIRS of rows: subtract, multiply, multiply-add and the accumulation as packed float32
    // operations (v_pk_*_f32: two rows per instruction), the logarithm per row -- three vector instructions per term instead of
    // five, and one 16-byte LDS read for four rows
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    static_assert(RPG % 4 == 0, "rows per thread in fours");
    const float *xg = xs + r0;
    float mx[RPG], sm[RPG];                      // running maximum and sum of 2^(z log2 e - mx)
#pragma unroll
    for (int r = 0; r < RPG; r++) { mx[r] = -3.0e38f; sm[r] = 0.f; }
    for (int k = kk; k < KM; k += gsz) {
        float z2[RPG];
        if ((k == kk ? cnt_first : bt.cnt[k]) > 0.0) {
            f32x2_t acc2[RPG / 2];
#pragma unroll
            for (int r = 0; r < RPG / 2; r++) acc2[r] = (f32x2_t){0.f, 0.f};
            const f32x2_t one2 = {1.f, 1.f};
            auto term = [&](int d, float m, float q) {
                const f32x2_t m2 = {m, m}, q2 = {q, q};
#pragma unroll
                for (int r4 = 0; r4 < RPG / 4; r4++) {
                    const f32x4_t xv = *reinterpret_cast<const f32x4_t *>(xg + d * FBB_R32 + 4 * r4);
                    const f32x2_t d0 = m2 - xv.xy, d1 = m2 - xv.zw;
                    const f32x2_t u0 = __builtin_elementwise_fma(d0 * d0, q2, one2), u1 = __builtin_elementwise_fma(d1 * d1, q2, one2);
                    acc2[2 * r4] += (f32x2_t){__builtin_amdgcn_logf(u0.x), __builtin_amdgcn_logf(u0.y)};       // v_log_f32: log2
                    acc2[2 * r4 + 1] += (f32x2_t){__builtin_amdgcn_logf(u1.x), __builtin_amdgcn_logf(u1.y)};
                }
            };
            int d = (dbg & 2) ? D : 0;
            for (; d + 4 <= D; d += 4) {
                float m[4], q[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    m[j] = TLDS ? tm[(d + j) * KM + k] : (float)bt.mean_t[(int64_t)(d + j) * KM + k];
                    q[j] = TLDS ? tq[(d + j) * KM + k] : (float)bt.q_t[(int64_t)(d + j) * KM + k];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) term(d + j, m[j], q[j]);
            }
            for (; d < D; d++)
                term(d, TLDS ? tm[d * KM + k] : (float)bt.mean_t[(int64_t)d * KM + k], TLDS ? tq[d * KM + k] : (float)bt.q_t[(int64_t)d * KM + k]);
            const double zc = k == kk ? zc_first : bt.zconst[k], hl = (k == kk ? half_first : bt.half[k]) * 0.6931471805599453;
#pragma unroll
            for (int r = 0; r < RPG; r++) z2[r] = (float)((zc - hl * (double)acc2[r >> 1][r & 1]) * 1.4426950408889634);
        } else {
#pragma unroll
            for (int r = 0; r < RPG; r++) z2[r] = (float)((zc_empty + lpr[r0 + r]) * 1.4426950408889634);
        }
#pragma unroll
        for (int r = 0; r < RPG; r++) {
            const float nm = fmaxf(mx[r], z2[r]);
            sm[r] = sm[r] * __builtin_amdgcn_exp2f(mx[r] - nm) + __builtin_amdgcn_exp2f(z2[r] - nm);
            mx[r] = nm;
        }
    }
    // per row: the waves' (maximum, sum) by shuffles, then the four waves' by the row's thread -- one barrier for all rows
    // (a block-wide maximum and a block-wide sum per row were 32 double barriers)
    if (dbg & 8) return;
    if (!(dbg & 4)) {
        const int w = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int r = 0; r < RPG; r++) {
            const float M = fb_wave_max_f32(mx[r]);
            const float S = fb_wave_sum_f32(sm[r] == 0.f ? 0.f : sm[r] * __builtin_amdgcn_exp2f(mx[r] - M));
            // (with two thread groups a wave lies inside one group: 128 threads each)
            if (lane == 0) { wm[w * FBB_R32 + r0 + r] = M; ws[w * FBB_R32 + r0 + r] = S; }
        }
        if (NG > 1) {                              // the rows of the other groups: neutral
            for (int r = lane; r < FBB_R32; r += 64)
                if (r < r0 || r >= r0 + RPG) { wm[w * FBB_R32 + r] = -3.0e38f; ws[w * FBB_R32 + r] = 0.f; }
        }
    }
    __syncthreads();
    if (tid < nr) {
        float M = wm[tid];
        for (int w = 1; w < 4; w++) M = fmaxf(M, wm[w * FBB_R32 + tid]);
        // (hardware exp2 / log2 like the terms above: the fp64 library calls were a dependent chain of ~2.5 us on sixteen lanes)
        float S = 0.f;
        for (int w = 0; w < 4; w++) S += ws[w * FBB_R32 + tid] == 0.f ? 0.f : ws[w * FBB_R32 + tid] * __builtin_amdgcn_exp2f(wm[w * FBB_R32 + tid] - M);
        score[row0 + tid] = ((double)__builtin_amdgcn_logf(S) + (double)M) * 0.6931471805599453 - norm;
    }
}

// ---------------------------------------------------------------------------------------
// One Gibbs step of the diagonal (Student-t) sampler in float32 terms as ONE kernel (round 4): span scores, boundaries and
// slots of an utterance by the workgroup that owns it -- k_fbb_score_diag32, k_fbb_segment and k_fbb_assign back to back
// were three launches of ~25 us each for 125 utterances (6.6 us of each the empty launch, the rest one workgroup's chain
// of round trips), and the assignment kernel evaluated the token likelihoods a second time.  Here:
//   (1) the utterance's spans that have an embedding are listed (band or triangle), their rows and the slot tables staged
//       in LDS (float32, rows transposed as in k_fbb_score_diag32);
//   (2) L[i][k] = (zconst_k - half_k ln 2 sum_d log2(1 + (m - x)^2 q)) log2 e for every (span, slot) -- the arithmetic of
//       k_fbb_score_diag32 term for term (packed float32 on row pairs), kept in LDS: it is the span score's summand AND
//       the assignment logit (zconst holds the prior's term already), empty slots carry zc_empty + the row's prior predictive;
//   (3) the span score = log-sum-exp over the slots, per wave of 64 slots by DPP and the waves' pairs in order: the bits
//       of k_fbb_score_diag32 (also written to `score`);
//   (4) vec, forward filtering / backward sampling by wave 0 (fb_dp_sample, as k_fbb_segment), old tokens' slots cleared;
//   (5) every new token's slot: one wave per token, softmax over L's row in the log2 domain (v_exp_f32 / v_log_f32), the
//       draw walking the slots in order on the token's uniform (fb_draw_chunked, as k_fbb_assign).
// Applies where the tables and L fit in LDS (K_max <= 256; the host checks), no language model; probes as the kernels
// it replaces (segk_fbb_set_probe).  Token likelihoods differ from k_fbb_assign's float32 form in the last bits (one
// multiply-add instead of multiply and add): both within the 1e-4 contract, the draws are checked against their
// slots' intervals (tests/test_gpu_tolerance_modes.py).
// ---------------------------------------------------------------------------------------
struct FbbStepArgs {
    uint64_t sweep;
    int b, n_max, r_cap;               // r_cap: spans with an embedding per utterance at most (a multiple of 8)
    double wip, time_power_term, anneal_fb, anneal_am, prior_alpha;
    double *score;
    uint8_t *boundaries;
    int32_t *new_tok, *n_new;
    double *out_logprob;
    int32_t *status;
    double *probe_alpha, *probe_ll;
    int64_t probe_ld;
    int dbg;                           // development (make DEV=1, SEGK_STEP_DBG; results wrong): 1 no terms, 2 no span-score reductions,
                                       // 4 no DP, 8 no draws, 16 return behind the staging
};

template <typename XT>
__global__ __launch_bounds__(512) void k_fbb_step_diag32(segk_corpus c, segk_fbgmm f, segk_fbatch bt, FbbMap map, FbbStepArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int D = c.D, KM = f.K_max, NM = c.N_max, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int RC = A.r_cap, nch = (KM + 63) >> 6;
    const int64_t triMax = (int64_t)NM * (NM + 1) / 2;
    // ---- LDS
    double *vec = (double *)smem;                       // [triMax]
    double *al = vec + triMax;                          // [N_max]
    double *ww = al + NM;                               // [N_max + 1]
    double *pr = ww + NM + 1;                           // [N_max + 1]
    double *dur_l = pr + NM + 1;                        // [triMax]
    double *sc_l = dur_l + triMax;                      // [r_cap] span scores
    double *lpr = sc_l + RC;                            // [r_cap] prior predictive of the spans' rows
    double *zp = lpr + RC;                              // [waves][K_max] a token's probabilities
    double *prk = zp + (int64_t)nw * KM;                // [K_max] the prior's term of every slot (probe only)
    float *tm = (float *)(prk + KM);                    // [D][K_max]
    float *tq = tm + D * KM;                            // [D][K_max]
    float *xs = (float *)(((uintptr_t)(tq + D * KM) + 15) & ~(uintptr_t)15);    // [D][r_cap], 16-byte aligned (read as float4)
    float *Lm = xs + D * RC;                            // [r_cap][K_max]
    float *wm = Lm + (int64_t)RC * KM;                  // [r_cap][4] the chunks' maxima
    float *ws = wm + RC * 4;                            // [r_cap][4] and sums
    int32_t *vid_l = (int32_t *)(ws + RC * 4);          // [triMax]
    int32_t *ent_j = vid_l + triMax;                    // [r_cap] triangular index of the i-th span with an embedding
    int32_t *ent_id = ent_j + RC;                       // [r_cap] its embedding row
    int32_t *old_l = ent_id + RC;                       // [N_max]
    int32_t *tok_l = old_l + NM;                        // [N_max]
    int32_t *tokj_l = tok_l + NM;                       // [N_max]
    short *ent_of = (short *)(tokj_l + NM);             // [triMax] span -> its place in the list (-1)
    uint8_t *bnd_l = (uint8_t *)(ent_of + ((triMax + 3) & ~(int64_t)3));    // [N_max]
    short *occ_l = (short *)(bnd_l + ((NM + 15) & ~15));                     // [K_max] the occupied slots, then the empty ones
    __shared__ int sh_nent, sh_nn, sh_nocc;
    int s, idx;
    if (!fbb_locate(map, blockIdx.x, &s, &idx)) return;
    const int slice = map.lo[s];
    const int utt = bt.utt_range[(slice * bt.n_blocks + A.b) * 2] + idx;
    const double total_cnt = bt.scal[0];
    const int N = c.lengths[utt];
    const int tri = N * (N + 1) / 2;
    const FbSpanTab tab = fb_span_tab(c, utt, N, A.n_max);
    uint8_t *bnd_g = A.boundaries + (int64_t)utt * NM;
    const XT *X = (const XT *)c.X;
    // ---- (1) tables, span table, flags
    for (int j = tid; j < D * KM; j += nt) {
        tm[j] = (float)bt.mean_t[j];
        tq[j] = (float)bt.q_t[j];
    }
    for (int j = tid; j < N; j += nt) bnd_l[j] = bnd_g[j];
    for (int j = tid; j < tri; j += nt) { vid_l[j] = -1; dur_l[j] = 0.0; ent_of[j] = -1; }
    __syncthreads();
    if (tab.band) {
        const int W = tab.W;
        for (int i = tid; i < N * W; i += nt) {
            const int t = i / W + 1, s2 = t - 1 - (i - (t - 1) * W);
            if (s2 >= 0) {
                vid_l[t * (t - 1) / 2 + s2] = tab.bandi[i];
                dur_l[t * (t - 1) / 2 + s2] = tab.bandd[i];
            }
        }
    } else {
        for (int j = tid; j < tri; j += nt) { vid_l[j] = tab.vid[j]; dur_l[j] = tab.dur[j]; }
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) {                       // the occupied slots first, the empty ones behind them (a wave of 64
        int no = 0;                                     // consecutive slots, half of them empty, paid the terms for all 64)
        for (int k0 = 0; k0 < KM; k0 += 64) {
            const int k = k0 + lane;
            const bool occ = k < KM && bt.cnt[k] > 0.0;
            const unsigned long long m = __ballot(occ);
            if (occ) occ_l[no + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
            no += __popcll(m);
        }
        int ne = 0;
        for (int k0 = 0; k0 < KM; k0 += 64) {
            const int k = k0 + lane;
            const bool emp = k < KM && !(bt.cnt[k] > 0.0);
            const unsigned long long m = __ballot(emp);
            if (emp) occ_l[no + ne + __popcll(m & ((1ull << lane) - 1ull))] = (short)k;
            ne += __popcll(m);
        }
        if (lane == 0) sh_nocc = no;
    }
    if (tid < 64) {                                     // the spans with an embedding, in table order (ballot + prefix count)
        int n = 0;
        for (int j0 = 0; j0 < tri; j0 += 64) {
            const int j = j0 + lane;
            const bool ok = j < tri && vid_l[j] >= 0;
            const unsigned long long m = __ballot(ok);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && pos < RC) { ent_j[pos] = j; ent_id[pos] = vid_l[j]; ent_of[j] = (short)pos; }
            n += __popcll(m);
        }
        if (lane == 0) sh_nent = n;
    }
    __syncthreads();
    const int n_ent = sh_nent;
    if (n_ent > RC) {                                   // (the host sized r_cap for the span tables: cannot happen)
        if (tid == 0) atomicOr(A.status, 32);
        return;
    }
    const int n_pad = (n_ent + 7) & ~7;
    for (int j = tid; j < n_pad * D; j += nt) {
        const int i = j / D, d = j - i * D;
        xs[d * RC + i] = i < n_ent ? (float)X[(int64_t)ent_id[i] * c.ldx + d] : 0.f;
    }
    for (int i = tid; i < n_pad; i += nt) lpr[i] = i < n_ent ? bt.prior_rows[ent_id[i]] : 0.0;
    const double zc_empty = f.lms * fb_log_fast(A.prior_alpha / (double)KM);
    const double norm = f.lms * fb_log_fast(total_cnt + A.prior_alpha);
    if (A.probe_ll)
        for (int k = tid; k < KM; k += nt) prk[k] = bt.cnt[k] > 0.0 ? bt.zconst[k] - bt.lconst[k] : zc_empty;
    __syncthreads();
    if (A.dbg & 16) return;
    // ---- (2) L: work item = (group of eight spans, occupied slot); the empty slots' logits need no terms
    {
        const int n_occ = (A.dbg & 1) ? 0 : sh_nocc;
        for (int item = tid; item < n_pad * (KM - n_occ); item += nt) {
            const int i = item / (KM - n_occ), k = occ_l[n_occ + item - i * (KM - n_occ)];
            Lm[(int64_t)i * KM + k] = (float)((zc_empty + lpr[i]) * 1.4426950408889634);
        }
        const int n_items = (n_pad >> 3) * n_occ;
        for (int item = tid; item < n_items; item += nt) {
            const int g = item / n_occ, k = occ_l[item - g * n_occ], r0 = 8 * g;
            const float *xg = xs + r0;
            float z2[8];
            {
                f32x2_t acc2[4];
#pragma unroll
                for (int r = 0; r < 4; r++) acc2[r] = (f32x2_t){0.f, 0.f};
                const f32x2_t one2 = {1.f, 1.f};
                for (int d = 0; d < D; d++) {
                    const float m = tm[d * KM + k], q = tq[d * KM + k];
                    const f32x2_t m2 = {m, m}, q2 = {q, q};
#pragma unroll
                    for (int r4 = 0; r4 < 2; r4++) {
                        const f32x4_t xv = *reinterpret_cast<const f32x4_t *>(xg + d * RC + 4 * r4);
                        const f32x2_t d0 = m2 - xv.xy, d1 = m2 - xv.zw;
                        const f32x2_t u0 = __builtin_elementwise_fma(d0 * d0, q2, one2), u1 = __builtin_elementwise_fma(d1 * d1, q2, one2);
                        acc2[2 * r4] += (f32x2_t){__builtin_amdgcn_logf(u0.x), __builtin_amdgcn_logf(u0.y)};
                        acc2[2 * r4 + 1] += (f32x2_t){__builtin_amdgcn_logf(u1.x), __builtin_amdgcn_logf(u1.y)};
                    }
                }
                const double zc = bt.zconst[k], hl = bt.half[k] * 0.6931471805599453;
#pragma unroll
                for (int r = 0; r < 8; r++) z2[r] = (float)((zc - hl * (double)acc2[r >> 1][r & 1]) * 1.4426950408889634);
            }
#pragma unroll
            for (int r = 0; r < 8; r++) Lm[(int64_t)(r0 + r) * KM + k] = z2[r];
        }
    }
    __syncthreads();
    // ---- (3) span scores: (maximum, sum) per chunk of 64 slots by one wave, the chunks in order by the span's thread
    // One THREAD per (span, chunk): the maximum of the chunk's 64 logits, their exponentials and the sum in the order the wave
    // reduction of k_fbb_score_diag32 adds them (fb_wave_sum_f32: a balanced tree over each sixteen lanes, then (r0 + r1) +
    // (r2 + r3); float addition commutes, so the tree alone fixes the bits) -- a wave per pair was 240 dependent DPP chains,
    // 7 us of the step.
    for (int pair = tid; pair < ((A.dbg & 2) ? 0 : n_ent * nch); pair += nt) {
        const int i = pair / nch, ch = pair - i * nch;
        const float *Lr = Lm + (int64_t)i * KM + 64 * ch;
        const int nv = KM - 64 * ch < 64 ? KM - 64 * ch : 64;
        float M = -3.0e38f;
        for (int q = 0; q < nv; q++) M = fmaxf(M, Lr[q]);
        float r16[4];
#pragma unroll
        for (int b16 = 0; b16 < 4; b16++) {
            float e[16];
#pragma unroll
            for (int q = 0; q < 16; q++) e[q] = 16 * b16 + q < nv ? __builtin_amdgcn_exp2f(Lr[16 * b16 + q] - M) : 0.f;
#pragma unroll
            for (int q = 0; q < 16; q += 2) e[q] += e[q + 1];
#pragma unroll
            for (int q = 0; q < 16; q += 4) e[q] += e[q + 2];
#pragma unroll
            for (int q = 0; q < 16; q += 8) e[q] += e[q + 4];
            r16[b16] = e[0] + e[8];
        }
        wm[i * 4 + ch] = M;
        ws[i * 4 + ch] = (r16[0] + r16[1]) + (r16[2] + r16[3]);
    }
    __syncthreads();
    for (int i = tid; i < n_ent; i += nt) {
        float M = wm[i * 4];
        for (int ch = 1; ch < nch; ch++) M = fmaxf(M, wm[i * 4 + ch]);
        float S = 0.f;
        for (int ch = 0; ch < nch; ch++) S += ws[i * 4 + ch] == 0.f ? 0.f : ws[i * 4 + ch] * __builtin_amdgcn_exp2f(wm[i * 4 + ch] - M);
        const double scv = ((double)__builtin_amdgcn_logf(S) + (double)M) * 0.6931471805599453 - norm;
        sc_l[i] = scv;
        A.score[ent_id[i]] = scv;
    }
    __syncthreads();
    // ---- (4) vec (unigram_acoustic_wordseg.py:474-511), the DP by wave 0
    for (int j = tid; j < tri; j += nt) {
        const int i = ent_of[j];
        double v = NEG_INF_D;
        if (i >= 0) {
            const double dd = dur_l[j];
            v = isnan(dd) ? NEG_INF_D : sc_l[i] * (A.time_power_term == 1.0 ? dd : pow(dd, A.time_power_term));
        }
        vec[j] = v + A.wip;
    }
    __syncthreads();
    if (tid < 64) {
        const int n_old = N <= 64 ? fb_collect_tokens_wave(vid_l, bnd_l, N, old_l, lane)
                                  : __shfl(lane == 0 ? fb_collect_tokens(vid_l, bnd_l, N, old_l) : 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        CounterUniforms usrc = {bt.seed, A.sweep, (uint64_t)utt, 0};
        const double total = (A.dbg & 4) ? 0.0 : fb_dp_sample(vec, al, ww, pr, N, tri, A.n_max, 0, 0.0, A.anneal_fb, bnd_l, lane, usrc, bt.fast_dp);
        if (A.probe_alpha)
            for (int j = lane; j < N; j += 64) A.probe_alpha[(int64_t)utt * NM + j] = al[j];
        for (int j = lane; j < n_old; j += 64) bt.slot[old_l[j]] = -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the new segments: embedding row and span, in order
        int nn = 0;
        if (N <= 64) {
            const unsigned long long mask = __ballot(lane < N && bnd_l[lane < N ? lane : 0] != 0);
            const bool bit = lane < N && ((mask >> lane) & 1ull);
            const unsigned long long below = mask & ((1ull << lane) - 1ull);
            const int jp = below ? 64 - __clzll((long long)below) : 0;
            const int j = bit ? (lane + 1) * lane / 2 + jp : 0;
            const int id = bit ? vid_l[j] : -1;
            const unsigned long long keep = __ballot(bit && id >= 0);
            if (bit && id >= 0) {
                const int pos = __popcll(keep & ((1ull << lane) - 1ull));
                tok_l[pos] = id;
                tokj_l[pos] = j;
            }
            nn = __popcll(keep);
        } else if (lane == 0) {
            int jp = 0;
            for (int j2 = 0; j2 < N; j2++)
                if (bnd_l[j2]) {
                    const int j = (j2 + 1) * j2 / 2 + jp;
                    const int id = vid_l[j];
                    if (id >= 0) { tok_l[nn] = id; tokj_l[nn] = j; nn++; }
                    jp = j2 + 1;
                }
        }
        if (N > 64) nn = __shfl(nn, 0);
        for (int j = lane; j < N; j += 64) bnd_g[j] = bnd_l[j];
        for (int t = lane; t < nn; t += 64) A.new_tok[(int64_t)utt * NM + t] = tok_l[t];
        if (lane == 0) {
            if (total == NEG_INF_D) atomicOr(A.status, 16);
            A.out_logprob[utt] = total;
            A.n_new[utt] = nn;
            sh_nn = nn;
        }
    }
    __syncthreads();
    // ---- (5) the new segments' slots: one wave per token (fbgmm.py:422-463; no language model: the draws are independent)
    const int nn = (A.dbg & 8) ? 0 : sh_nn;
    const float inv_T = (float)(1. / A.anneal_am);
    for (int t = wv; t < nn; t += nw) {
        const int64_t e = tok_l[t];
        const float *zr = Lm + (int64_t)ent_of[tokj_l[t]] * KM;
        double *zq = zp + (int64_t)wv * KM;
        float mx = -3.0e38f;
        for (int k = lane; k < KM; k += 64) mx = fmaxf(mx, zr[k]);
        mx = fb_wave_max_f32(mx);
        float sm = 0.f;
        for (int k = lane; k < KM; k += 64) sm += __builtin_amdgcn_exp2f(zr[k] - mx);
        float lse2 = mx + __builtin_amdgcn_logf(fb_wave_sum_f32(sm));
        if (A.anneal_am != 1.0) {                            // fbgmm.py:446-449
            float mx2 = -3.0e38f;
            for (int k = lane; k < KM; k += 64) mx2 = fmaxf(mx2, inv_T * (zr[k] - lse2));
            mx2 = fb_wave_max_f32(mx2);
            float sm2 = 0.f;
            for (int k = lane; k < KM; k += 64) sm2 += __builtin_amdgcn_exp2f(inv_T * (zr[k] - lse2) - mx2);
            const float lse3 = mx2 + __builtin_amdgcn_logf(fb_wave_sum_f32(sm2));
            for (int k = lane; k < KM; k += 64) zq[k] = (double)__builtin_amdgcn_exp2f(inv_T * (zr[k] - lse2) - lse3);
        } else {
            for (int k = lane; k < KM; k += 64) zq[k] = (double)__builtin_amdgcn_exp2f(zr[k] - lse2);
        }
        if (A.probe_ll)
            for (int k = lane; k < KM; k += 64)
                A.probe_ll[((int64_t)utt * NM + t) * A.probe_ld + k] = (double)zr[k] * 0.6931471805599453 - prk[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int kd = fb_draw_chunked(zq, KM, segk_u01(bt.seed, A.sweep, (uint64_t)utt, (uint64_t)(NM + t)), lane);
        if (lane == 0) bt.slot[e] = kd;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// calibration of the roofline of k_fbb_score_diag32: the same four float32 instructions per term (subtract, multiply,
// multiply-add, v_log_f32) and the accumulate, operands in registers, eight independent chains per thread -- what the
// vector ALUs deliver on this chip for the kernel's inner term with nothing else in the way
__global__ __launch_bounds__(256) void k_vlog_calibrate(int iters, float seed, float *out)
{
    float acc[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { acc[j] = 0.f; m[j] = seed + 0.01f * (float)(threadIdx.x + j); }
    float x = seed * 0.5f, q = 1.0f + seed;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float delta = m[j] - x;
            acc[j] += __builtin_amdgcn_logf(1.f + (delta * delta) * q);
        }
        x += 1e-6f;
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) t += acc[j];
    if (t == 12345.678f) out[0] = t;              // never true: keeps the loop alive
}

// ---------------------------------------------------------------------------------------
// boundaries of one utterance per workgroup (128 threads; wave 0 runs the DP)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_fbb_segment(segk_corpus c, segk_fbatch bt, FbbMap map, int b, uint64_t sweep, int n_max, double wip,
                              double time_power_term, double anneal_temp, const double *score, uint8_t *boundaries,
                              int32_t *new_tok, int32_t *n_new, double *out_logprob, int32_t *status, double *probe_alpha)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int s, idx;
    if (!fbb_locate(map, blockIdx.x, &s, &idx)) return;
    const int slice = map.lo[s];
    const int utt = bt.utt_range[(slice * bt.n_blocks + b) * 2] + idx;
    const int N = c.lengths[utt];
    const int tri = N * (N + 1) / 2;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const FbSpanTab tab = fb_span_tab(c, utt, N, n_max);
    const int32_t *vid = tab.vid;
    double *vec = (double *)smem;           // [tri]
    double *a = vec + triMax;               // [N]
    double *w = a + c.N_max;                // [N+1]
    double *pr = w + c.N_max + 1;           // [N+1]
    int32_t *old = (int32_t *)(pr + c.N_max + 1);      // [N_max]
    // The span ids (triangular) and the boundary flags staged in LDS: the token lists before and after the DP are then taken from
    // LDS -- from memory each was two dependent round trips, the second one behind the DP's own stores of the flags -- and the
    // flags go back to memory once, at the end (the workgroup is a chain of round trips: 19 of its 25 us)
    int32_t *vid_l = old + c.N_max;                    // [triMax]
    uint8_t *bnd_l = (uint8_t *)(vid_l + triMax);      // [N_max]
    uint8_t *bnd_g = boundaries + (int64_t)utt * c.N_max;
    for (int j = threadIdx.x; j < N; j += blockDim.x) bnd_l[j] = bnd_g[j];
    if (tab.band) {
        for (int j = threadIdx.x; j < tri; j += blockDim.x) vid_l[j] = -1;
        __syncthreads();
        const int W = tab.W;
        for (int i = threadIdx.x; i < N * W; i += blockDim.x) {
            const int t = i / W + 1, s2 = t - 1 - (i - (t - 1) * W);
            if (s2 >= 0) vid_l[t * (t - 1) / 2 + s2] = tab.bandi[i];
        }
    } else {
        for (int j = threadIdx.x; j < tri; j += blockDim.x) vid_l[j] = vid[j];
    }
    fb_fill_vec(tab, N, tri, [&](int id) { return score[id]; }, time_power_term, wip, vec, threadIdx.x, blockDim.x);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint8_t *bnd = bnd_l;
    vid = vid_l;
    const int n_old = N <= 64 ? fb_collect_tokens_wave(vid, bnd, N, old, lane)
                              : __shfl(lane == 0 ? fb_collect_tokens(vid, bnd, N, old) : 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    CounterUniforms usrc = {bt.seed, sweep, (uint64_t)utt, 0};
    const double total = fb_dp_sample(vec, a, w, pr, N, tri, n_max, 0, 0.0, anneal_temp, bnd, lane, usrc, bt.fast_dp);
    if (probe_alpha)            // segk_fbb_set_probe: the forward filter's values (the backward pass only reads them)
        for (int j = lane; j < N; j += 64) probe_alpha[(int64_t)utt * c.N_max + j] = a[j];
    for (int j = lane; j < n_old; j += 64) bt.slot[old[j]] = -1;
    // (the boundary flags were written by lane 0 of this wave)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nn = N <= 64 ? fb_collect_tokens_wave(vid, bnd, N, new_tok + (int64_t)utt * c.N_max, lane) : -1;
    for (int j = lane; j < N; j += 64) bnd_g[j] = bnd_l[j];
    if (lane != 0) return;
    if (total == NEG_INF_D) atomicOr(status, 16);
    out_logprob[utt] = total;
    n_new[utt] = nn >= 0 ? nn : fb_collect_tokens(vid, bnd, N, new_tok + (int64_t)utt * c.N_max);
}

// ---------------------------------------------------------------------------------------
// slots of the new tokens of one utterance per workgroup
// ---------------------------------------------------------------------------------------
template <typename XT, int COV, int F32 = 0>
__global__ __launch_bounds__(512) void k_fbb_assign(segk_corpus c, segk_fbgmm f, segk_fbatch bt, FbbMap map, int b, uint64_t sweep,
                             double prior_alpha, double anneal_temp, const int32_t *new_tok, const int32_t *n_new,
                             int rcap, int dbg, const float *llmat, int64_t ll_ld, double *probe_ll, int64_t probe_ld)
{
    // The likelihood part of the logits does not depend on the previous segment's slot, so it is
    // evaluated for up to `rcap` (<= FBB_R) tokens of the utterance at once -- the component
    // parameters are streamed once per chunk -- and kept in LDS; the draws then run in token order
    // (with a language model each needs the slot of the one before).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = c.D, KM = f.K_max, tid = threadIdx.x, nt = blockDim.x;
    double *z = (double *)smem;                  // [K_max]
    double *ll = z + KM;                         // [rcap][K_max]
    double *xs = ll + (int64_t)rcap * KM;        // [FBA_R][D]
    double *lpr = xs + FBA_R * D;                // [FBA_R]
    double *red = lpr + FBA_R;                   // [16]
    __shared__ int sh_k;
    int s, idx;
    if (!fbb_locate(map, blockIdx.x, &s, &idx)) return;
    const int slice = map.lo[s];
    const int utt = bt.utt_range[(slice * bt.n_blocks + b) * 2] + idx;
    const XT *X = (const XT *)c.X;
    const int nn = n_new[utt];
    const double zc_empty = f.lms * log(prior_alpha / (double)KM);
    const double tot = bt.scal[0];
    int j_prev = -1;
    for (int t0 = 0; t0 < nn; t0 += rcap) {
        const int nr = nn - t0 < rcap ? nn - t0 : rcap;
        __syncthreads();
        if (llmat) {
            // token likelihoods from the matrix-core contraction (segk_fbb_token_scores): row blockIdx.x*N_max + t
            // holds acc_k = (zconst_k - s_k/2 - norm + <.,.>) log2 e per slot and the empty-slot row at K_max
            const double LN2 = 0.6931471805599453;
            const double norm = f.lms * log(tot + prior_alpha);
            const double n_empty = (double)KM - bt.scal[1];
            const int32_t *cm = fbb_cmap(&bt, KM);
            const int pc = cm[KM];
            for (int r = 0; r < nr; r++) {
                const float *mrow = llmat + ((int64_t)blockIdx.x * c.N_max + t0 + r) * ll_ld;
                for (int k = tid; k < KM; k += nt) {
                    const int ck = cm[k];
                    ll[(int64_t)r * KM + k] = bt.cnt[k] > 0.0 && ck >= 0 ? (double)mrow[ck] * LN2 - (bt.zconst[k] - bt.lconst[k]) + norm
                                                                         : (double)mrow[pc] * LN2 - zc_empty - log(n_empty) + norm;
                }
            }
        } else {
        for (int j = tid; j < (rcap > FBB_R ? rcap : FBB_R) * D; j += nt) {
            const int r = j / D, d = j - r * D;
            xs[j] = r < nr ? (double)X[(int64_t)new_tok[(int64_t)utt * c.N_max + t0 + r] * c.ldx + d] : 0.0;
        }
        __syncthreads();
        {
            const int w = tid >> 6, lane = tid & 63, nw = nt >> 6;
            for (int r = w; r < nr; r += nw) {
                const double v = bt.prior_rows ? bt.prior_rows[new_tok[(int64_t)utt * c.N_max + t0 + r]]
                                               : fbb_prior_row<XT>(f, D, xs + r * D, lane);
                if (lane == 0) lpr[r] = v;
            }
        }
        __syncthreads();
        if (nt >= 4 * 128 && KM <= 128) {
            // few slots: four groups of threads take two rows each (the per-(row, slot) arithmetic is unchanged; with one
            // thread per slot 100 of 256 threads each walked D x 8 software logarithms) -- four rows each when the chunk
            // holds sixteen tokens (an utterance with nine tokens paid the whole chunk twice)
            auto rows_of_group = [&](auto NRC) {
                constexpr int NR = decltype(NRC)::value;
                const int k = tid & 127, r0 = NR * (tid >> 7);
                if (k < KM && r0 < nr) {
                    if (bt.cnt[k] > 0.0) {
                        double acc[NR];
#pragma unroll
                        for (int r = 0; r < NR; r++) acc[r] = 0.0;
                        if (F32) {
                            float a32[NR];
#pragma unroll
                            for (int r = 0; r < NR; r++) a32[r] = 0.f;
                            fbb_accumulate_rows32<NR>(bt, KM, k, (dbg & 1) ? 1 : D, xs + r0 * D, a32);
#pragma unroll
                            for (int r = 0; r < NR; r++) acc[r] = 0.6931471805599453 * (double)a32[r];
                        } else
                        fbb_accumulate_rows<COV, NR>(bt, KM, k, (dbg & 1) ? 1 : D, xs + r0 * D, acc);
                        const double lc = bt.lconst[k], h = bt.half[k];
#pragma unroll
                        for (int r = 0; r < NR; r++)
                            if (r0 + r < nr) ll[(int64_t)(r0 + r) * KM + k] = lc - h * acc[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < NR; r++)
                            if (r0 + r < nr) ll[(int64_t)(r0 + r) * KM + k] = lpr[r0 + r];
                    }
                }
            };
            if (rcap > FBB_R) rows_of_group(std::integral_constant<int, 4>());
            else rows_of_group(std::integral_constant<int, 2>());
        } else
        for (int k = tid; k < KM; k += nt) {
            if (bt.cnt[k] > 0.0) {
                double acc[FBB_R];
#pragma unroll
                for (int r = 0; r < FBB_R; r++) acc[r] = 0.0;
                if (F32) {
                    float a32[FBB_R];
#pragma unroll
                    for (int r = 0; r < FBB_R; r++) a32[r] = 0.f;
                    fbb_accumulate_rows32<FBB_R>(bt, KM, k, (dbg & 1) ? 1 : D, xs, a32);
#pragma unroll
                    for (int r = 0; r < FBB_R; r++) acc[r] = 0.6931471805599453 * (double)a32[r];
                } else
                fbb_accumulate<COV>(bt, KM, k, (dbg & 1) ? 1 : D, xs, acc);
                const double lc = bt.lconst[k], h = bt.half[k];
#pragma unroll
                for (int r = 0; r < FBB_R; r++)
                    if (r < nr) ll[(int64_t)r * KM + k] = lc - h * acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < FBB_R; r++)
                    if (r < nr) ll[(int64_t)r * KM + k] = lpr[r];
            }
        }
        }
        __syncthreads();
        if (probe_ll) {         // segk_fbb_set_probe: the likelihood part of the logits, as the draws below use it
            for (int j = tid; j < nr * KM; j += nt) {
                const int r = j / KM, k = j - r * KM;
                probe_ll[((int64_t)utt * c.N_max + t0 + r) * probe_ld + k] = ll[(int64_t)r * KM + k];
            }
            __syncthreads();
        }
        if (!f.lm_unigram) {
            // Without a language model a token's draw does not depend on the token before it: one WAVE per token, maxima
            // and sums by shuffles.  The sums keep the association of the block-wide form below at 256 threads (per wave
            // of 64 consecutive slots a butterfly, the waves' results added in order), so the probabilities -- and the
            // draws -- are the same bits.
            const int w = tid >> 6, lane = tid & 63, nw = nt >> 6;
            for (int r = w; r < ((dbg & 2) ? 1 : nr); r += nw) {
                const int64_t e = new_tok[(int64_t)utt * c.N_max + t0 + r];
                double *zr = ll + (int64_t)r * KM;
                double mx = NEG_INF_D;
                for (int k = lane; k < KM; k += 64) {
                    const double n = bt.cnt[k];
                    const double v = (n > 0.0 ? f.lms * log(prior_alpha / (double)KM + n) : zc_empty) + zr[k];      // fbgmm.py:436-440
                    zr[k] = v;
                    mx = v > mx ? v : mx;
                }
                mx = fb_wave_max(mx, false);
                auto sum_exp = [&](double shift) -> double {       // sum_k exp(zr[k] - shift) in the order of block_sum at 256 threads
                    double tot = 0.0;
                    for (int cw = 0; cw < 4 && cw * 64 < KM; cw++) {
                        double sv = 0.0;
                        for (int k = cw * 64 + lane; k < KM; k += 256) sv += exp(zr[k] - shift);
                        sv = fb_wave_sum(sv);
                        tot = cw == 0 ? sv : tot + sv;
                    }
                    return tot;
                };
                double lse = log(sum_exp(mx)) + mx;
                if (anneal_temp != 1.0) {                           // fbgmm.py:446-449
                    double mx2 = NEG_INF_D;
                    for (int k = lane; k < KM; k += 64) {
                        const double v = (1. / anneal_temp) * (zr[k] - lse);
                        zr[k] = v;
                        mx2 = v > mx2 ? v : mx2;
                    }
                    mx2 = fb_wave_max(mx2, false);
                    lse = log(sum_exp(mx2)) + mx2;
                }
                for (int k = lane; k < KM; k += 64) zr[k] = exp(zr[k] - lse);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int k = (dbg & 4) ? 0 : fb_draw_chunked(zr, KM, segk_u01(bt.seed, sweep, (uint64_t)utt, (uint64_t)(c.N_max + t0 + r)), lane);
                if (lane == 0) bt.slot[e] = k;
            }
        } else
        for (int r = 0; r < ((dbg & 2) ? 1 : nr); r++) {
            const int64_t e = new_tok[(int64_t)utt * c.N_max + t0 + r];
            for (int k = tid; k < KM; k += nt) {
                const double n = bt.cnt[k];
                double pz;
                if (!f.lm_unigram) pz = n > 0.0 ? f.lms * log(prior_alpha / (double)KM + n) : zc_empty;   // fbgmm.py:436-440
                else if (j_prev < 0) pz = (log(n + f.lm_a / (double)KM) - log(tot + f.lm_a)) * f.lms;  // bigram_lms.py:64-69
                else {                                                                                // bigram_lms.py:84-91
                    const double pi = (n + f.lm_a / (double)KM) / (tot + f.lm_a);
                    const double pij = (1. - f.lm_lambda) * ((double)f.lm_bigram[(int64_t)j_prev * KM + k] + f.lm_b / (double)KM)
                                       / (bt.cnt[j_prev] + f.lm_b);
                    pz = log(f.lm_lambda * pi + pij) * f.lms;
                }
                z[k] = pz + ll[(int64_t)r * KM + k];
            }
            __syncthreads();
            // softmax (scipy logsumexp order), optional annealing (fbgmm.py:446-449), utils.draw
            double mx = NEG_INF_D;
            for (int k = tid; k < KM; k += nt) mx = z[k] > mx ? z[k] : mx;
            mx = block_max(mx, red);
            double sm = 0.0;
            for (int k = tid; k < KM; k += nt) sm += exp(z[k] - mx);
            sm = block_sum(sm, red);
            double lse = log(sm) + mx;
            if (anneal_temp != 1.0) {
                for (int k = tid; k < KM; k += nt) z[k] = (1. / anneal_temp) * (z[k] - lse);
                __syncthreads();
                double mx2 = NEG_INF_D;
                for (int k = tid; k < KM; k += nt) mx2 = z[k] > mx2 ? z[k] : mx2;
                mx2 = block_max(mx2, red);
                double s2 = 0.0;
                for (int k = tid; k < KM; k += nt) s2 += exp(z[k] - mx2);
                s2 = block_sum(s2, red);
                lse = log(s2) + mx2;
            }
            for (int k = tid; k < KM; k += nt) z[k] = exp(z[k] - lse);
            __syncthreads();
            if (tid < 64) {
                const int k = (dbg & 4) ? 0 : fb_draw_chunked(z, KM, segk_u01(bt.seed, sweep, (uint64_t)utt, (uint64_t)(c.N_max + t0 + r)), tid);
                if (tid == 0) {
                    bt.slot[e] = k;
                    sh_k = k;
                }
            }
            __syncthreads();
            if (f.lm_unigram) j_prev = sh_k;
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same with a language model and the token likelihoods from the matrix-core contraction: ONE WAVE per utterance.
// The draws of an utterance are a chain (each needs the slot of the token before), and the block-wide form above
// pays seven workgroup barriers per token for it; a wave needs none, and four times as many utterances are in flight.
// Per token the same expressions in the same order; the sums keep the association of the block-wide form at 256 threads
// (virtual thread v = 64 cw + lane sums the slots v, v + 256, ...; a butterfly per cw; the four results added in order),
// so the probabilities and the draws are the same bits.
// ---------------------------------------------------------------------------------------
// PROBE: the variant the library launches while segk_fbb_set_probe is in force -- the same statements plus the stores of
// the token likelihoods (a test of a null pointer per slot inside the token loop cost the production kernel ~9 %)
template <int F32, bool PROBE = false>
__global__ __launch_bounds__(256) void k_fbb_assign_lm_wave(segk_corpus c, segk_fbgmm f, segk_fbatch bt, FbbMap map, int b, uint64_t sweep,
                                                            double prior_alpha, double anneal_temp, const int32_t *new_tok,
                                                            const int32_t *n_new, const float *llmat, int64_t ll_ld, int n_items,
                                                            double *probe_ll, int64_t probe_ld)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KM = f.K_max, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + w;
    if (item >= n_items) return;
    double *z = (double *)smem + (int64_t)w * KM;      // [K_max], this wave's
    // column of the token-likelihood matrix per slot (an empty slot: the pseudo-component's).  The table is the workgroup's, every
    // wave writes all of it (the same values) and waits for its own stores only -- waves leave early, no workgroup barrier here
    // (the packed form below wants the other direction, slot of column c, in the same place)
    int *cm = (int *)((double *)smem + 4 * (int64_t)KM);   // [K_max + 1]
    constexpr int KPL = 16;
    const bool packed = F32 && KM <= 64 * KPL;
    {
        const int32_t *g = fbb_cmap(&bt, KM);
        const int pcol = g[KM];
        if (packed) {
            for (int q = lane; q <= pcol; q += 64) cm[q] = g[KM + 2 + q];
        } else {
            for (int k = lane; k <= KM; k += 64) {
                const int ck = g[k];
                cm[k] = ck >= 0 ? ck : pcol;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int s, idx;
    if (!fbb_locate(map, item, &s, &idx)) return;
    const int slice = map.lo[s];
    const int utt = bt.utt_range[(slice * bt.n_blocks + b) * 2] + idx;
    const int nn = n_new[utt];
    const double LN2 = 0.6931471805599453;
    const double zc_empty = f.lms * log(prior_alpha / (double)KM);
    const double tot = bt.scal[0];
    const double norm = f.lms * log(tot + prior_alpha);
    const double n_empty = (double)KM - bt.scal[1];
    auto sum_exp = [&](double shift) -> double {           // sum_k exp(z[k] - shift) in the order of block_sum at 256 threads
        double acc = 0.0;
        for (int cw = 0; cw < 4 && cw * 64 < KM; cw++) {
            double sv = 0.0;
            for (int k = cw * 64 + lane; k < KM; k += 256) sv += exp(z[k] - shift);
            sv = fb_wave_sum(sv);
            acc = cw == 0 ? sv : acc + sv;
        }
        return acc;
    };
    int j_prev = -1;
    if (F32) {
        // The matrix-core modes (`score_precision` f32 / f16: log-likelihoods within 1e-4 relative) take the token
        // likelihoods as float32 already; here the softmax follows: K_max hardware logarithms and 2 K_max hardware
        // exponentials per token (v_log_f32 / v_exp_f32 on differences that were formed in fp64) instead of 3 K_max fp64
        // software ones -- the step was bound by them (290 us per Gibbs step of bigram_c5).  Probabilities carry ~3e-6
        // relative error; a draw changes when the uniform falls that close to a cumulative boundary.
        const float LOG2E = 1.4426950408889634f;
        const double le = log(n_empty), ltot = log(tot + f.lm_a), inv_tot = 1. / (tot + f.lm_a), aK = f.lm_a / (double)KM, bK = f.lm_b / (double)KM;
        if (packed) {
            // The softmax over the OCCUPIED slots only (419 of 1 000 at configs[4]): the empty ones share one logit -- no
            // tokens, no bigram counts, the pseudo-component's likelihood -- which is evaluated once and enters the sum
            // n_empty times.  Lane l holds the columns l, l + 64, ... of the packed matrix (up to 16; what does not change
            // from token to token stays in registers for the utterance, and a token's loads are all issued before the
            // first is used).  The draw walks the slots in slot order as utils.draw does: the columns' probabilities go to
            // LDS, lane l sums the chunk [l per, (l + 1) per) of them plus the empty slots that lie between its first
            // column's slot and the next chunk's, a scan over the lanes finds the chunk and one walk over it the slot.
            const int *inv = cm;
            const int n_occ = fbb_cmap(&bt, KM)[KM];
            const int jmax = (n_occ + 63) >> 6, per = jmax > 0 ? jmax : 1;
            const bool has_e = n_empty > 0.0;
            // (per column: count + a/K for the prior's argument -- which goes through v_log_f32 and is formed in float32 --
            // and the constant of the likelihood; float32 like the matrix's values they are added to)
            float naK[KPL], c2[KPL];
#pragma unroll
            for (int j = 0; j < KPL; j++) {
                naK[j] = 1.f;
                c2[j] = 0.f;
                if (j < jmax) {
                    const int q = lane + 64 * j, k = q < n_occ ? inv[q] : inv[0];
                    naK[j] = (float)(bt.cnt[k] + aK);
                    c2[j] = (float)(norm - (bt.zconst[k] - bt.lconst[k]));
                }
            }
            const float lam_it = (float)(f.lm_lambda * inv_tot), bKf = (float)bK, lmsf = (float)f.lms, LN2f = 0.6931471805599453f;
            const int *bg32 = (const int *)f.lm_bigram;                         // (counts: the low words)
            const int lo = lane * per < n_occ ? lane * per : n_occ, hi = lo + per < n_occ ? lo + per : n_occ;
            const int b_lo = lane == 0 ? 0 : (lo < n_occ ? inv[lo] : KM);       // first slot of this lane's chunk
            const int b_hi = hi < n_occ ? inv[hi] : KM;                         // ... of the next one's
            const double n_gap = (double)((b_hi - b_lo) - (hi - lo));           // empty slots inside
            for (int t = 0; t < nn; t++) {
                const int64_t e = new_tok[(int64_t)utt * c.N_max + t];
                const float *mrow = llmat + ((int64_t)item * c.N_max + t) * ll_ld;
                const double inv_prev = j_prev >= 0 ? (1. - f.lm_lambda) / (bt.cnt[j_prev] + f.lm_b) : 0.0;
                const float mre = mrow[n_occ];
                const float inv_prevf = (float)inv_prev;
                float mr[KPL];
                int bg[KPL];
#pragma unroll
                for (int j = 0; j < KPL; j++) {
                    mr[j] = 0.f;
                    bg[j] = 0;
                    if (j < jmax) {
                        const int q = lane + 64 * j;
                        mr[j] = mrow[q < n_occ ? q : n_occ];
                        bg[j] = j_prev >= 0 ? bg32[2 * ((int64_t)j_prev * KM + inv[q < n_occ ? q : 0])] : 0;
                    }
                }
                const double empty_ll = (double)mre * LN2 - zc_empty - le + norm;
                double ze = NEG_INF_D;
                if (has_e) {
                    double pz;
                    if (j_prev < 0) pz = ((double)(__builtin_amdgcn_logf((float)aK) * 0.6931471805599453f) - ltot) * f.lms;
                    else pz = (double)(__builtin_amdgcn_logf((float)aK * lam_it + bKf * inv_prevf) * LN2f * lmsf);
                    ze = pz + empty_ll;
                }
                if constexpr (PROBE) {
                    const int32_t *g = fbb_cmap(&bt, KM);
                    for (int k = lane; k < KM; k += 64)
                        if (g[k] < 0) probe_ll[((int64_t)utt * c.N_max + t) * probe_ld + k] = empty_ll;
                }
                double zv[KPL], mx = ze;
#pragma unroll
                for (int j = 0; j < KPL; j++) {
                    zv[j] = NEG_INF_D;
                    if (j < jmax) {
                        const double llv = (double)mr[j] * LN2 + (double)c2[j];
                        const bool ok = lane + 64 * j < n_occ;
                        if constexpr (PROBE)
                            if (ok) probe_ll[((int64_t)utt * c.N_max + t) * probe_ld + inv[lane + 64 * j]] = llv;
                        double pz;
                        if (j_prev < 0) pz = ((double)(__builtin_amdgcn_logf(naK[j]) * LN2f) - ltot) * f.lms;
                        else pz = (double)(__builtin_amdgcn_logf(naK[j] * lam_it + ((float)bg[j] + bKf) * inv_prevf) * LN2f * lmsf);
                        zv[j] = ok ? pz + llv : NEG_INF_D;
                        mx = zv[j] > mx ? zv[j] : mx;
                    }
                }
                mx = fb_wave_max(mx, false);
                // (the sum of the exponentials in float32 like its terms, the logarithm by v_log_f32: the differences were formed in
                // fp64, what is summed lies in [0, 1])
                auto sum_reg = [&](double shift) -> double {
                    float sv = 0.f;
#pragma unroll
                    for (int j = 0; j < KPL; j++)
                        if (j < jmax) sv += __builtin_amdgcn_exp2f((float)(zv[j] - shift) * LOG2E);      // 2^-inf = 0 behind the last column
                    sv = fb_wave_sum_f32(sv);
                    if (has_e) sv += (float)n_empty * __builtin_amdgcn_exp2f((float)(ze - shift) * LOG2E);
                    return (double)sv;
                };
                double lse = fb_log_fast(sum_reg(mx)) + mx;
                if (anneal_temp != 1.0) {                               // fbgmm.py:446-449
                    double mx2 = NEG_INF_D;
                    if (has_e) {
                        ze = (1. / anneal_temp) * (ze - lse);
                        mx2 = ze;
                    }
#pragma unroll
                    for (int j = 0; j < KPL; j++)
                        if (j < jmax) {
                            zv[j] = (1. / anneal_temp) * (zv[j] - lse);
                            mx2 = zv[j] > mx2 ? zv[j] : mx2;
                        }
                    mx2 = fb_wave_max(mx2, false);
                    lse = fb_log_fast(sum_reg(mx2)) + mx2;
                }
#pragma unroll
                for (int j = 0; j < KPL; j++)
                    if (j < jmax && lane + 64 * j < n_occ) z[lane + 64 * j] = (double)__builtin_amdgcn_exp2f((float)(zv[j] - lse) * LOG2E);
                const double p_e = has_e ? (double)__builtin_amdgcn_exp2f((float)(ze - lse) * LOG2E) : 0.0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // ---- the draw
                double sl = 0.0;
                for (int q = lo; q < hi; q++) sl += z[q];
                sl += n_gap * p_e;
                const double incl = fb_wave_scan(sl);
                const double u = segk_u01(bt.seed, sweep, (uint64_t)utt, (uint64_t)(c.N_max + t));
                const unsigned long long below = __ballot(u < incl);
                int kd = KM - 1;
                if (below) {
                    const int run = __ffsll((long long)below) - 1;
                    double r = u - (fb_readlane(incl, run) - fb_readlane(sl, run));
                    const int rlo = run * per < n_occ ? run * per : n_occ, rhi = rlo + per < n_occ ? rlo + per : n_occ;
                    int pos = __builtin_amdgcn_readlane(b_lo, run);
                    const int end = __builtin_amdgcn_readlane(b_hi, run);
                    bool found = false;
                    for (int q = rlo; q < rhi && !found; q++) {
                        const int k = inv[q];
                        const int gap = k - pos;
                        if (gap > 0) {
                            const double m = (double)gap * p_e;
                            if (r < m) {
                                int o = (int)(r / p_e);
                                o = o < 0 ? 0 : (o > gap - 1 ? gap - 1 : o);
                                kd = pos + o;
                                found = true;
                                break;
                            }
                            r -= m;
                        }
                        const double pq = z[q];
                        if (r < pq) {
                            kd = k;
                            found = true;
                            break;
                        }
                        r -= pq;
                        pos = k + 1;
                    }
                    if (!found) {
                        const int gap = end - pos;
                        if (gap > 0 && p_e > 0.0) {
                            int o = (int)(r / p_e);
                            o = o < 0 ? 0 : (o > gap - 1 ? gap - 1 : o);
                            kd = pos + o;
                        } else {
                            kd = pos > 0 ? pos - 1 : 0;
                        }
                    }
                }
                if (lane == 0) bt.slot[e] = kd;
                j_prev = kd;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
        for (int t = 0; t < nn; t++) {
            const int64_t e = new_tok[(int64_t)utt * c.N_max + t];
            const float *mrow = llmat + ((int64_t)item * c.N_max + t) * ll_ld;
            const double inv_prev = j_prev >= 0 ? (1. - f.lm_lambda) / (bt.cnt[j_prev] + f.lm_b) : 0.0;
            const double empty_ll = (double)mrow[cm[KM]] * LN2 - zc_empty - le + norm;
            double mx = NEG_INF_D;
            for (int k = lane; k < KM; k += 64) {
                const double n = bt.cnt[k];
                const double llv = n > 0.0 ? (double)mrow[cm[k]] * LN2 - (bt.zconst[k] - bt.lconst[k]) + norm : empty_ll;
                if constexpr (PROBE) probe_ll[((int64_t)utt * c.N_max + t) * probe_ld + k] = llv;
                double pz;
                if (j_prev < 0) pz = ((double)(__builtin_amdgcn_logf((float)(n + aK)) * 0.6931471805599453f) - ltot) * f.lms;      // bigram_lms.py:64-69
                else {                                                                                                          // bigram_lms.py:84-91
                    const double pi = (n + aK) * inv_tot;
                    const double pij = ((double)f.lm_bigram[(int64_t)j_prev * KM + k] + bK) * inv_prev;
                    pz = (double)(__builtin_amdgcn_logf((float)(f.lm_lambda * pi + pij)) * 0.6931471805599453f) * f.lms;
                }
                const double v = pz + llv;
                z[k] = v;
                mx = v > mx ? v : mx;
            }
            mx = fb_wave_max(mx, false);
            auto sum_exp32 = [&](double shift) -> double {
                double sv = 0.0;
                for (int k = lane; k < KM; k += 64) sv += (double)__builtin_amdgcn_exp2f((float)(z[k] - shift) * LOG2E);
                sv = fb_wave_sum(sv);
                return sv;
            };
            double lse = log(sum_exp32(mx)) + mx;
            if (anneal_temp != 1.0) {                               // fbgmm.py:446-449
                double mx2 = NEG_INF_D;
                for (int k = lane; k < KM; k += 64) {
                    const double v = (1. / anneal_temp) * (z[k] - lse);
                    z[k] = v;
                    mx2 = v > mx2 ? v : mx2;
                }
                mx2 = fb_wave_max(mx2, false);
                lse = log(sum_exp32(mx2)) + mx2;
            }
            for (int k = lane; k < KM; k += 64) z[k] = (double)__builtin_amdgcn_exp2f((float)(z[k] - lse) * LOG2E);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int kd = fb_draw_chunked(z, KM, segk_u01(bt.seed, sweep, (uint64_t)utt, (uint64_t)(c.N_max + t)), lane);
            if (lane == 0) bt.slot[e] = kd;
            j_prev = __shfl(kd, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    for (int t = 0; t < nn; t++) {
        const int64_t e = new_tok[(int64_t)utt * c.N_max + t];
        const float *mrow = llmat + ((int64_t)item * c.N_max + t) * ll_ld;
        double mx = NEG_INF_D;
        for (int k = lane; k < KM; k += 64) {
            const double n = bt.cnt[k];
            const double llv = n > 0.0 ? (double)mrow[cm[k]] * LN2 - (bt.zconst[k] - bt.lconst[k]) + norm
                                       : (double)mrow[cm[KM]] * LN2 - zc_empty - log(n_empty) + norm;
            if constexpr (PROBE) probe_ll[((int64_t)utt * c.N_max + t) * probe_ld + k] = llv;
            double pz;
            if (j_prev < 0) pz = (log(n + f.lm_a / (double)KM) - log(tot + f.lm_a)) * f.lms;                       // bigram_lms.py:64-69
            else {                                                                                              // bigram_lms.py:84-91
                const double pi = (n + f.lm_a / (double)KM) / (tot + f.lm_a);
                const double pij = (1. - f.lm_lambda) * ((double)f.lm_bigram[(int64_t)j_prev * KM + k] + f.lm_b / (double)KM)
                                   / (bt.cnt[j_prev] + f.lm_b);
                pz = log(f.lm_lambda * pi + pij) * f.lms;
            }
            const double v = pz + llv;
            z[k] = v;
            mx = v > mx ? v : mx;
        }
        mx = fb_wave_max(mx, false);
        double lse = log(sum_exp(mx)) + mx;
        if (anneal_temp != 1.0) {                               // fbgmm.py:446-449
            double mx2 = NEG_INF_D;
            for (int k = lane; k < KM; k += 64) {
                const double v = (1. / anneal_temp) * (z[k] - lse);
                z[k] = v;
                mx2 = v > mx2 ? v : mx2;
            }
            mx2 = fb_wave_max(mx2, false);
            lse = log(sum_exp(mx2)) + mx2;
        }
        for (int k = lane; k < KM; k += 64) z[k] = exp(z[k] - lse);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int kd = fb_draw_chunked(z, KM, segk_u01(bt.seed, sweep, (uint64_t)utt, (uint64_t)(c.N_max + t)), lane);
        if (lane == 0) bt.slot[e] = kd;
        j_prev = __shfl(kd, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------
// language-model tables (integers): transcripts of block b, all slices, +/-; and the fill of the
// replicated transcript store from the local slices' new tokens.
//   lm_tok [B][S][U_max][N_max] int32 slots, -1 padded
// ---------------------------------------------------------------------------------------
__global__ void k_fbb_lm_apply(segk_fbgmm f, segk_fbatch bt, int N_max, int b, int sign)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_rows = (int64_t)bt.n_slices * bt.u_max;
    if (row >= n_rows) return;
    const int32_t *t = bt.lm_tok + ((int64_t)b * n_rows + row) * N_max;
    int prev = -1;
    // (eight slots per round trip: one load per token with the exit test in between was a chain of up to N_max of them)
    for (int j0 = 0; j0 < N_max; j0 += 8) {
        int kk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) kk[u] = t[j0 + u < N_max ? j0 + u : N_max - 1];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = j0 + u < N_max ? kk[u] : -1;
            if (k < 0) return;
            if (prev >= 0)
                atomicAdd((unsigned long long *)&f.lm_bigram[(int64_t)prev * f.K_max + k], (unsigned long long)(long long)sign);
            prev = k;
        }
    }
}

__global__ void k_fbb_lm_fill(segk_corpus c, segk_fbatch bt, FbbMap map, int b, const int32_t *new_tok,
                              const int32_t *n_new)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    int s, idx;
    if (!fbb_locate(map, g, &s, &idx)) return;
    const int slice = map.lo[s];
    const int utt = bt.utt_range[(slice * bt.n_blocks + b) * 2] + idx;
    int32_t *t = bt.lm_tok + (((int64_t)b * bt.n_slices + slice) * bt.u_max + idx) * c.N_max;
    const int nn = n_new[utt];
    // (eight tokens per pair of round trips: the stores to t[] kept the compiler from moving any load of the next token up)
    for (int j0 = 0; j0 < c.N_max; j0 += 8) {
        int e[8], k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = j0 + u < nn ? new_tok[(int64_t)utt * c.N_max + j0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; u++) k[u] = e[u] >= 0 ? bt.slot[e[u]] : -1;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (j0 + u < c.N_max) t[j0 + u] = j0 + u < nn ? k[u] : -1;
    }
}

// token lists of all utterances from the boundaries (entering batch mode)
__global__ void k_fbb_collect(segk_corpus c, const uint8_t *boundaries, int32_t *new_tok, int32_t *n_new)
{
    const int utt = blockIdx.x * blockDim.x + threadIdx.x;
    if (utt >= c.n_utt) return;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    n_new[utt] = fb_collect_tokens(c.vec_ids + (int64_t)utt * triMax, boundaries + (int64_t)utt * c.N_max,
                                   c.lengths[utt], new_tok + (int64_t)utt * c.N_max);
}

// the reference's view: occupied slots relabelled 0..K-1 in increasing slot order
__global__ void k_fbb_remap(segk_fbgmm f, segk_fbatch bt, int32_t *remap)
{
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < f.K_max; k0 += blockDim.x) {
        const int k = k0 + threadIdx.x;
        const int occ = (k < f.K_max) && (bt.cnt[k] > 0.0);
        // ranks inside this chunk of blockDim slots (blockDim = 64: one wave)
        const unsigned long long bal = __ballot(occ);
        const int before = __popcll(bal & ((1ull << threadIdx.x) - 1ull));
        if (k < f.K_max) remap[k] = occ ? base + before : -1;
        __syncthreads();
        if (threadIdx.x == 0) base += __popcll(bal);
        __syncthreads();
    }
    if (threadIdx.x == 0) *f.K = base;
}

__global__ void k_fbb_apply_remap(segk_fbgmm f, segk_fbatch bt, int64_t n_emb, const int32_t *remap)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_emb) return;
    const int sl = bt.slot[e];
    f.assignments[e] = sl >= 0 ? remap[sl] : -1;
}

// ---------------------------------------------------------------------------------------
// fp32 matrix-core span score of fixed-variance components (the MFMA kernel of segk_score_f32.hip in
// log-sum-exp mode): operands.
//   Y[row] = [x_0^2, x_0, x_1^2, x_1, ...]                                    (once per corpus)
//   tile row of slot k (count > 0):  [-pp_kd/2, pp_kd*mu_kd]_d * log2(e),
//       constant (zconst_k - sum_d pp_kd mu_kd^2 / 2) * log2(e)
//   pseudo-component K_max = all empty slots: [-p0_d/2, p0_d*mu0_d]_d * log2(e), constant
//       (lms*log(alpha/K_max) + log(#empty) + kconst[K_max] - sum_d p0_d mu0_d^2 / 2) * log2(e)
//   so that acc_k = z_k * log2(e) and sum_k 2^acc_k = sum over ALL K_max slots of exp(z).
// Tile image layout: segk_internal.h (32 slots per tile, dims 4g + 2(lane>>5) + {0,1}).
// ---------------------------------------------------------------------------------------
template <typename XT>
__global__ void k_fbb_make_y(segk_corpus c, float *y, int64_t ldy)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= c.n_emb * ldy) return;
    const int64_t row = idx / ldy;
    const int j = (int)(idx - row * ldy), d = j >> 1;
    float v = 0.f;
    if (d < c.D) {
        const float x = (float)((const XT *)c.X)[row * c.ldx + d];
        v = (j & 1) ? x : x * x;
    }
    y[idx] = v;
}

__global__ void k_fbb_tiles32(segk_fbgmm f, segk_fbatch bt, int D, double prior_alpha)
{
    const double LOG2E = 1.4426950408889634;
    const int tile = blockIdx.x, KM = f.K_max, D2 = 2 * D;
    const int G = segk_gmax(D2), stride = segk_tile_stride(D2);
    float *T = bt.tiles32 + (int64_t)tile * stride;
    __shared__ double cst[32];
    {
        const int ci = threadIdx.x >> 3, sub = threadIdx.x & 7;      // 256 threads = 32 slots x 8 lanes
        const int k = tile * 32 + ci;
        double s = 0.0;
        if (k < KM && bt.cnt[k] > 0.0)
            for (int d = sub; d < D; d += 8) {
                const double m = bt.mean_t[(int64_t)d * KM + k];
                s += bt.q_t[(int64_t)d * KM + k] * m * m;
            }
        else if (k == KM)
            for (int d = sub; d < D; d += 8) s += f.prior_c[d] * f.prior_b[d] * f.prior_b[d];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (sub == 0) {
            // the normaliser of log_marg_i, lms*log(total + alpha) (fbgmm.py:268-272), is folded into the constants
            const double norm = f.lms * log(bt.scal[0] + prior_alpha);
            double v = -3.0e38;
            if (k < KM && bt.cnt[k] > 0.0) v = (bt.zconst[k] - 0.5 * s - norm) * LOG2E;
            else if (k == KM) {
                const double n_empty = (double)KM - bt.scal[1];
                if (n_empty > 0.0)
                    v = (f.lms * log(prior_alpha / (double)KM) + log(n_empty) + f.kconst[KM] - 0.5 * s - norm) * LOG2E;
            }
            cst[ci] = v;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < stride; idx += blockDim.x) {
        float v = 0.f;
        if (idx < G * 128) {
            const int g = idx >> 7, rem = idx & 127, lane = rem >> 1, sel = rem & 1;
            const int k = tile * 32 + (lane & 31);
            const int j = 4 * g + 2 * (lane >> 5) + sel, d = j >> 1;
            if (d < D) {
                if (k < KM && bt.cnt[k] > 0.0) {
                    const double q = bt.q_t[(int64_t)d * KM + k];
                    v = (float)(((j & 1) ? q * bt.mean_t[(int64_t)d * KM + k] : -0.5 * q) * LOG2E);
                } else if (k == KM) {
                    v = (float)(((j & 1) ? f.prior_c[d] * f.prior_b[d] : -0.5 * f.prior_c[d]) * LOG2E);
                }
            }
        } else if (idx < G * 128 + 32) {
            v = (float)cst[idx - G * 128];
        }
        T[idx] = v;
    }
}

// fp16x2 form: the same per-slot rows and constants as k_fbb_tiles32, written as a plain float32 matrix
// [(K_max + 1), 2D] (interleaved [-pp/2, pp*mu] * log2 e) plus constants; consts16[K_max + 1] collects the
// largest squared row norm (the exponent of the fp16 scaling follows it).  segk_sp_prepare_tiles turns
// them into the operand image of k_kmeans_score_sp<.., 2, 1>.
// Columns of the fp16x2 operand image: the occupied slots in slot order, then the pseudo-component of the empty ones; everything
// behind is "absent".  (At configs[4] 419 of 1 000 slots are occupied in the settled chain: 14 tiles of components instead of 32
// for the span scores and the token likelihoods -- an empty slot's score is the pseudo-component's, multiplying its tile was
// wasted matrix work.)  The token-likelihood matrix has the same columns: the kernels that read it go through the map
// (fbb_cmap), which the waves of k_fbb_rows16 write.
__global__ void k_fbb_rows16(segk_fbgmm f, segk_fbatch bt, int D, double prior_alpha, const int32_t *cmap,
                             unsigned long long *tiles_fb)
{
    const double LOG2E = 1.4426950408889634;
    const int KM = f.K_max, D2 = 2 * D;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + w;
    if (k > KM) return;
    const bool occupied = k < KM && bt.cnt[k] > 0.0;
    // The column of slot k = the number of occupied slots below it (k == K_max, the pseudo-component: all of them), counted by
    // the slot's own wave -- sixteen counts per lane at K_max = 1 000 -- instead of a one-workgroup launch in front
    // (k_fbb_compact, 4.8 us); every wave also writes its part of the maps behind the constants.
    int col = k;
    if (cmap) {
        int below = 0;
        for (int j0 = 0; j0 < k; j0 += 64) {
            const int j = j0 + lane;
            below += __popcll(__ballot(j < k && bt.cnt[j < k ? j : 0] > 0.0));
        }
        col = (occupied || k == KM) ? below : -1;
        int32_t *cm = const_cast<int32_t *>(cmap), *inv = cm + KM + 2;
        if (lane == 0) {
            if (k < KM) cm[k] = col;
            if (col >= 0) inv[col] = k;
        }
        if (k == KM) {                                // the pseudo-component's wave: the header, and what lies behind its column
            const int n_col = below + 1, n_t = (n_col + 31) / 32;
            if (lane == 0) {
                cm[KM] = below;
                cm[KM + 1] = n_t;
                // (a host-mapped word: the launcher of the token likelihoods sizes its tile split by the last count it saw)
                if (tiles_fb) __hip_atomic_store(tiles_fb, (unsigned long long)n_t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // the columns behind it inside the last tile in use are multiplied too: "absent", and rows that cannot overflow the image
            const int r_hi = n_t * 32 < KM + 1 ? n_t * 32 : KM + 1;
            for (int c2 = n_col + lane; c2 < r_hi; c2 += 64) bt.consts16[c2] = -3.0e38;
            for (int i = lane; i < (r_hi - n_col) * D2; i += 64) bt.rows32[(int64_t)n_col * D2 + i] = 0.f;
        }
    }
    if (col < 0) return;                              // an empty slot has no column
    double s = 0.0, n2 = 0.0;
    for (int d = lane; d < D; d += 64) {
        double t0 = 0.0, t1 = 0.0;
        if (occupied) {
            const double q = bt.q_t[(int64_t)d * KM + k], m = bt.mean_t[(int64_t)d * KM + k];
            t0 = -0.5 * q * LOG2E;
            t1 = q * m * LOG2E;
            s += q * m * m;
        } else if (k == KM) {
            t0 = -0.5 * f.prior_c[d] * LOG2E;
            t1 = f.prior_c[d] * f.prior_b[d] * LOG2E;
            s += f.prior_c[d] * f.prior_b[d] * f.prior_b[d];
        }
        const float r0 = (float)t0, r1 = (float)t1;
        bt.rows32[(int64_t)col * D2 + 2 * d] = r0;
        bt.rows32[(int64_t)col * D2 + 2 * d + 1] = r1;
        n2 += (double)r0 * r0 + (double)r1 * r1;
    }
    s = fb_wave_sum(s);
    n2 = fb_wave_sum(n2);
    if (lane == 0) {
        const double norm = f.lms * log(bt.scal[0] + prior_alpha);
        double v = -3.0e38;
        if (occupied) v = (bt.zconst[k] - 0.5 * s - norm) * LOG2E;
        else if (k == KM) {
            const double n_empty = (double)KM - bt.scal[1];
            if (n_empty > 0.0) v = (f.lms * log(prior_alpha / (double)KM) + log(n_empty) + f.kconst[KM] - 0.5 * s - norm) * LOG2E;
        }
        bt.consts16[col] = v;
        atomicMax((unsigned long long *)&bt.consts16[KM + 1], (unsigned long long)__double_as_longlong(n2));
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

static int check_fbb(const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt)
{
    SEGK_REQUIRE(c && f && bt, "NULL corpus / fbgmm / batch state");
    SEGK_REQUIRE(f->cov_type == 0 || f->cov_type == 1, "cov_type must be 0 (fixed) or 1 (diag)");
    SEGK_REQUIRE(c->D > 0 && c->D <= 64 * FBB_MAXCH, "batch mode supports D <= 256");
    SEGK_REQUIRE(c->N_max > 0 && c->N_max <= 64, "batch mode supports at most 64 landmarks per utterance");
    SEGK_REQUIRE(bt->n_slices >= 1 && bt->n_slices <= 16, "n_slices must be in 1..16");
    SEGK_REQUIRE(bt->n_blocks >= 2, "n_blocks must be >= 2 (with one block nothing is conditioned on)");
    SEGK_REQUIRE(f->kconst != NULL, "kconst buffer missing");
    return SEGK_OK;
}

// host copies of the ranges are passed in: counts[s] = work items of local slice s_lo + s
static int make_map(FbbMap *m, int s_lo, int s_n, const int32_t *counts, int per_wg)
{
    SEGK_REQUIRE(s_n >= 1 && s_n <= 16, "at most 16 local slices");
    m->n = s_n;
    m->off[0] = 0;
    for (int s = 0; s < s_n; s++) {
        m->lo[s] = s_lo + s;
        m->off[s + 1] = m->off[s] + (counts[s] + per_wg - 1) / per_wg;
    }
    return SEGK_OK;
}

extern "C" {

int32_t segk_fbb_collect(segk_ctx *ctx, const segk_corpus *c, const uint8_t *boundaries, int32_t *new_tok,
                         int32_t *n_new, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(c && c->n_utt > 0, "corpus without utterances");
    hipLaunchKernelGGL(k_fbb_collect, dim3((c->n_utt + 127) / 128), dim3(128), 0, (hipStream_t)stream, *c, boundaries,
                       new_tok, n_new);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_partials(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt,
                          int32_t s_lo, int32_t s_n, int32_t b, const int32_t *new_tok, const int32_t *n_new,
                          void *stream)
{
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(s_lo >= 0 && s_n >= 1 && s_lo + s_n <= bt->n_slices && b >= 0 && b < bt->n_blocks, "slice / block range");
    const int64_t waves = (int64_t)s_n * f->K_max;
    // many slots: bucket the tokens by slot first (k_fbb_sort), then every slot sums its own list; SEGK_FBB_SORT=0: the
    // one-step kernel.  LDS of the sort: 16 waves x K_max counters.
    const char *se = getenv("SEGK_FBB_SORT");
    const size_t lds_sort = ((size_t)FBS_WAVES * f->K_max + FBS_WAVES + 1) * sizeof(int32_t);
    if (ctx && f->K_max >= 256 && f->K_max <= 1024 && lds_sort <= 150 * 1024 && !(se && atoi(se) == 0)) {
        const int64_t stride = (int64_t)c->n_utt * c->N_max;             // more than any (slice, block) can hold
        const size_t need = ((size_t)s_n * stride + (size_t)s_n * (f->K_max + 1)) * sizeof(int32_t);
        if (ctx->fbs_bytes < need) {
            if (ctx->fbs_buf) (void)hipFree(ctx->fbs_buf);
            ctx->fbs_buf = nullptr;
            ctx->fbs_bytes = 0;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->fbs_buf, need));
            ctx->fbs_bytes = need;
        }
        int32_t *sorted = ctx->fbs_buf, *koff = ctx->fbs_buf + (size_t)s_n * stride;
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_fbb_sort, lds_sort));
        hipLaunchKernelGGL(k_fbb_sort, dim3(s_n), dim3(FBS_THREADS), lds_sort, (hipStream_t)stream, *c, *f, *bt, s_lo, b, new_tok, n_new,
                           sorted, stride, koff);
        DISPATCH_XT(c, hipLaunchKernelGGL(k_fbb_partials_sorted<XT>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                                           (hipStream_t)stream, *c, *f, *bt, s_lo, s_n, b, sorted, stride, koff););
        SEGK_LAUNCH_CHECK();
        ctx->fbb_scal_zeroed = (const void *)bt->scal;             // (the kernel cleared the totals for the next segk_fbb_prepare
        ctx->fbb_scal_stream = stream;                             //  enqueued on THIS stream)
        return SEGK_OK;
    }
    DISPATCH_XT(c, hipLaunchKernelGGL(k_fbb_partials<XT>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0,
                                       (hipStream_t)stream, *c, *f, *bt, s_lo, s_n, b, new_tok, n_new););
    SEGK_LAUNCH_CHECK();
    if (ctx) { ctx->fbb_scal_zeroed = (const void *)bt->scal; ctx->fbb_scal_stream = stream; }
    return SEGK_OK;
}

int32_t segk_fbb_prepare(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t b,
                         void *stream)
{
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(b >= -1 && b < bt->n_blocks, "block");
    hipStream_t st = (hipStream_t)stream;
    // (the totals are zero already when the last thing enqueued for them was segk_fbb_partials on this very stream)
    if (!ctx || ctx->fbb_scal_zeroed != (const void *)bt->scal || ctx->fbb_scal_stream != stream)
        SEGK_CHECK_HIP(hipMemsetAsync(bt->scal, 0, 2 * sizeof(double), st));
    if (ctx) { ctx->fbb_scal_zeroed = nullptr; ctx->fbb_scal_stream = nullptr; }
    const double alpha = f->lm_unigram ? f->lm_a : f->alpha;
    hipLaunchKernelGGL(k_fbb_prepare, dim3(f->K_max), dim3(192), 0, st, *f, *bt, c->D, b, alpha);
    if (bt->tiles16 && bt->y16 && f->cov_type == 0) {
        SEGK_REQUIRE(bt->rows32 && bt->consts16, "rows32 / consts16 scratch missing");
        // (k_fbb_prepare cleared the running maximum of the rows' norms behind the constants: a memset of 8 bytes was a launch of
        // 4.8 us, and so was the one-workgroup kernel that counted the columns -- every slot's wave of k_fbb_rows16 counts its own)
        // the occupied slots packed into the leading columns; the maps live behind the constants and are read by the score and
        // token-score calls of this step
        int32_t *cmap = fbb_cmap(bt, f->K_max);
        hipLaunchKernelGGL(k_fbb_rows16, dim3((f->K_max + 1 + 3) / 4), dim3(256), 0, st, *f, *bt, c->D, alpha, (const int32_t *)cmap,
                           ctx && ctx->miss_dev ? ctx->miss_dev + 2 : (unsigned long long *)nullptr);
        SEGK_LAUNCH_CHECK();
        return segk_sp_prepare_tiles(bt->rows32, bt->consts16, bt->consts16 + f->K_max + 1, f->K_max + 1, 2 * c->D,
                                     bt->tiles16, bt->y16, stream);
    }
    if (bt->tiles32 && f->cov_type == 0)
        hipLaunchKernelGGL(k_fbb_tiles32, dim3(segk_n_tiles(f->K_max + 1)), dim3(256), 0, st, *f, *bt, c->D, alpha);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_prior_rows(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, double *out, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(c && f && out, "null argument");
    SEGK_REQUIRE(c->x_dtype == SEGK_F32 || c->x_dtype == SEGK_F64, "unsupported dtype");
    if (c->n_emb == 0) return SEGK_OK;
    const unsigned grid = (unsigned)((c->n_emb + 3) / 4);
    DISPATCH_XT(c, { hipLaunchKernelGGL(k_fbb_prior_rows<XT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *c, *f, out); });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_make_y(segk_ctx *ctx, const segk_corpus *c, const segk_fbatch *bt, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(c && bt && bt->y && bt->ldy >= 2 * c->D && (bt->ldy & 3) == 0, "y buffer / ldy");
    const int64_t tot = c->n_emb * bt->ldy;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_fbb_make_y<XT>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                                       *c, bt->y, bt->ldy););
    SEGK_LAUNCH_CHECK();
    if (bt->y16) {
        SEGK_REQUIRE(2 * c->D <= 208, "the fp16x2 span score supports 2D <= 208");
        return segk_sp_prepare_rows(bt->y, bt->ldy, c->n_emb, 2 * c->D, bt->y16, stream);
    }
    return SEGK_OK;
}

int32_t segk_fbb_score_f32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt,
                           const int32_t *rows, int64_t n, double *score, void *stream)
{
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(f->cov_type == 0, "the matrix-core score exists for fixed-variance components only");
    SEGK_REQUIRE(bt->y && (bt->tiles32 || bt->tiles16), "y / tile buffers missing");
    SEGK_REQUIRE(rows != NULL && n >= 0, "row list");
    if (bt->tiles16 && bt->y16)
        return segk_launch_score_lse_sp(ctx, bt->y16, 2 * c->D, rows, 0, n, bt->tiles16, segk_n_tiles(f->K_max + 1), 0.0, score,
                                        stream, fbb_cmap(bt, f->K_max) + f->K_max + 1);
    return segk_launch_score_lse(ctx, bt->y, bt->ldy, 2 * c->D, rows, 0, n, bt->tiles32, segk_n_tiles(f->K_max + 1), 0.0, score,
                                 stream);
}

int32_t segk_fbb_score(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                       int32_t s_n, int32_t b, const int32_t *n_rows, double *score, void *stream)
{
    (void)ctx;
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_rows, FBB_R);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    const double alpha = f->lm_unigram ? f->lm_a : f->alpha;
    const size_t lds = (size_t)(FBB_R * c->D + FBB_R + 16) * sizeof(double);
    DISPATCH_XT(c, {
        if (f->cov_type == 0)
            hipLaunchKernelGGL((k_fbb_score<XT, 0>), dim3(m.off[s_n]), dim3(256), lds, (hipStream_t)stream, *c, *f, *bt, m,
                               b, alpha, score);
        else
            hipLaunchKernelGGL((k_fbb_score<XT, 1>), dim3(m.off[s_n]), dim3(256), lds, (hipStream_t)stream, *c, *f, *bt, m,
                               b, alpha, score);
    });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_score_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                              int32_t s_n, int32_t b, const int32_t *n_rows, double *score, void *stream)
{
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(f->cov_type == 1, "the float32 span score of this entry point is the diagonal (Student-t) one");
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_rows, FBB_R32);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    const double alpha = f->lm_unigram ? f->lm_a : f->alpha;
    const size_t lds0 = (size_t)(8 * c->D + FBB_R32) * sizeof(double) + (size_t)(8 * FBB_R32 + FBB_R32 * c->D) * sizeof(float);
    const size_t tbl = 2 * (size_t)c->D * f->K_max * sizeof(float);
    const bool tlds = lds0 + tbl <= 60 * 1024;                // two workgroups per CU keep their tables
    const bool two_groups = f->K_max <= 128;
    const size_t lds = lds0 + (tlds ? tbl : 0);
    const int d32dbg = segk_dev_env("SEGK_D32_DBG");          // make DEV=1 builds only (timing ablations, results wrong)
    hipStream_t st = (hipStream_t)stream;
    const bool prof = ctx && segk_prof_now(ctx);
    const int slot = prof ? ctx->prof_n % SEGK_PROF_SLOTS : 0;
    int64_t rows = 0;
    for (int s = 0; s < s_n; s++) rows += n_rows[s];
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    DISPATCH_XT(c, {
        if (lds > 48 * 1024) {
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_score_diag32<XT, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_score_diag32<XT, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (tlds && two_groups) hipLaunchKernelGGL((k_fbb_score_diag32<XT, 8, true>), dim3(m.off[s_n]), dim3(256), lds, st, *c, *f, *bt, m, b, alpha, score, d32dbg);
        else if (tlds) hipLaunchKernelGGL((k_fbb_score_diag32<XT, 16, true>), dim3(m.off[s_n]), dim3(256), lds, st, *c, *f, *bt, m, b, alpha, score, d32dbg);
        else if (two_groups) hipLaunchKernelGGL((k_fbb_score_diag32<XT, 8, false>), dim3(m.off[s_n]), dim3(256), lds, st, *c, *f, *bt, m, b, alpha, score, d32dbg);
        else hipLaunchKernelGGL((k_fbb_score_diag32<XT, 16, false>), dim3(m.off[s_n]), dim3(256), lds, st, *c, *f, *bt, m, b, alpha, score, d32dbg);
    });
    if (prof) {
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = rows;
        ctx->prof_kind = 5;
        ctx->prof_launches = 1;
        ctx->prof_n++;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// terms per second the vector ALUs sustain on the inner term of k_fbb_score_diag32 (k_vlog_calibrate), measured with
// events on `stream`; synchronises.  out_terms_per_s [host] double.
int32_t segk_calibrate_vlog(segk_ctx *ctx, double *out_terms_per_s, void *stream)
{
    SEGK_REQUIRE(ctx && out_terms_per_s, "arguments");
    hipStream_t st = (hipStream_t)stream;
    float *dummy = nullptr;
    SEGK_CHECK_HIP(hipMalloc((void **)&dummy, 64));
    hipEvent_t e0, e1;
    SEGK_CHECK_HIP(hipEventCreate(&e0));
    SEGK_CHECK_HIP(hipEventCreate(&e1));
    const int iters = 4096, grid = ctx->n_cu * 8;
    hipLaunchKernelGGL(k_vlog_calibrate, dim3(grid), dim3(256), 0, st, iters, 0.37f, dummy);      // warm-up
    SEGK_CHECK_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_vlog_calibrate, dim3(grid), dim3(256), 0, st, iters, 0.37f, dummy);
    SEGK_CHECK_HIP(hipEventRecord(e1, st));
    SEGK_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SEGK_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(dummy);
    *out_terms_per_s = (double)grid * 256.0 * 8.0 * iters / ((double)ms * 1e-3);
    return SEGK_OK;
}

int32_t segk_fbb_segment(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                         int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep, int32_t n_slices_min,
                         int32_t n_slices_max, double wip, double time_power_term, double anneal_temp,
                         const double *score, uint8_t *boundaries, int32_t *new_tok, int32_t *n_new,
                         double *out_logprob, int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "context");
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1, "n_slices_min must be 0 or 1");
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_utts, 1);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    const int64_t triMax = (int64_t)c->N_max * (c->N_max + 1) / 2;
    const size_t lds = (size_t)(triMax + 3 * c->N_max + 2) * sizeof(double) + (size_t)(c->N_max + triMax) * sizeof(int32_t) +
                       (size_t)((c->N_max + 15) & ~15);
    SEGK_REQUIRE(lds <= 64 * 1024, "N_max too large for the LDS score vector");
    hipLaunchKernelGGL(k_fbb_segment, dim3(m.off[s_n]), dim3(128), lds, (hipStream_t)stream, *c, *bt, m, b, sweep,
                       n_slices_max, wip, time_power_term, anneal_temp, score, boundaries, new_tok, n_new, out_logprob,
                       status, ctx->probe_alpha);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_token_scores(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt,
                              const int32_t *tok_rows, int64_t n, float *ll_mat, int64_t ll_ld, void *stream)
{
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(f->cov_type == 0 && bt->y16 && bt->tiles16, "needs the fp16x2 images (fixed-variance components)");
    SEGK_REQUIRE(tok_rows && ll_mat && ll_ld >= 32 * segk_n_tiles(f->K_max + 1) && (ll_ld & 3) == 0, "matrix / leading dimension");
    // (the matrix has the image's packed columns -- segk_fbb_prepare's map; the tiles in use as the device last reported them:
    // a hint for the split of the tiles over workgroups, nothing depends on it being current)
    int tiles_hint = 0;
    if (ctx && ctx->miss_host) tiles_hint = (int)ctx->miss_host[2];
    return segk_launch_score_mat_sp(bt->y16, 2 * c->D, tok_rows, n, bt->tiles16, segk_n_tiles(f->K_max + 1), ll_mat, ll_ld,
                                    stream, fbb_cmap(bt, f->K_max) + f->K_max + 1, tiles_hint);
}

static int32_t fbb_assign_impl(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                               int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep, double anneal_temp,
                               const int32_t *new_tok, const int32_t *n_new, const float *ll_mat, int64_t ll_ld, int f32,
                               void *stream)
{
    SEGK_REQUIRE(ctx, "context");
    SEGK_REQUIRE(!ctx->probe_ll || ctx->probe_ll_ld >= f->K_max, "probe leading dimension");
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_utts, 1);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    const double alpha = f->lm_unigram ? f->lm_a : f->alpha;
    // timing-only ablation knob (development): 1 = one dimension, 2 = one token, 4 = no draw
    const int dbg = segk_dev_env("SEGK_FBB_DBG");       // -DSEGK_DEV builds only
    // tokens per chunk: as many likelihood rows as fit beside the logits (two workgroups per CU)
    int rcap = FBB_R;
    const size_t fixed_b = (size_t)(f->K_max + FBA_R * c->D + FBA_R + 16) * sizeof(double);
    while (rcap > 1 && fixed_b + (size_t)rcap * f->K_max * sizeof(double) > 80 * 1024) rcap >>= 1;
    // 512 threads where the slots alone would leave most of 256 idle and the tokens' draws are independent (no language
    // model): four row groups in the likelihood phase, a wave per token in the draw phase -- and chunks of sixteen tokens
    const int nt_assign = (f->K_max <= 128 && !f->lm_unigram && rcap == FBB_R) ? 512 : 256;
    if (nt_assign == 512 && fixed_b + (size_t)FBA_R * f->K_max * sizeof(double) <= 80 * 1024) rcap = FBA_R;
    const size_t lds = fixed_b + (size_t)rcap * f->K_max * sizeof(double);
    SEGK_REQUIRE(lds <= 160 * 1024, "K_max too large for the LDS logits buffer");
    // language model + matrix-core likelihoods: one wave per utterance (SEGK_FBB_ASSIGN_WAVE=0: the block-wide form)
    const char *awe = getenv("SEGK_FBB_ASSIGN_WAVE");
    if (f->lm_unigram && ll_mat && dbg == 0 && !(awe && atoi(awe) == 0) && 4 * (size_t)f->K_max * sizeof(double) <= 146 * 1024) {
        const size_t ldsw = 4 * (size_t)f->K_max * sizeof(double) + sizeof(int) * (size_t)(f->K_max + 2);
        const int n_items = m.off[s_n];
        // SEGK_FBB_ASSIGN_WAVE=2: the softmax with the fp64 library functions (the bits of the block-wide form)
        const bool lib64 = awe && atoi(awe) == 2;
#define SEGK_LM_WAVE(FF, PP)                                                                                                    \
    do {                                                                                                                        \
        if (ldsw > 48 * 1024)                                                                                                   \
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_assign_lm_wave<FF, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw)); \
        hipLaunchKernelGGL((k_fbb_assign_lm_wave<FF, PP>), dim3((n_items + 3) / 4), dim3(256), ldsw, (hipStream_t)stream, *c, *f, *bt, m, b, \
                           sweep, alpha, anneal_temp, new_tok, n_new, ll_mat, ll_ld, n_items, ctx->probe_ll, ctx->probe_ll_ld);  \
    } while (0)
        if (ctx->probe_ll) {
            if (lib64) SEGK_LM_WAVE(0, true);
            else SEGK_LM_WAVE(1, true);
        } else {
            if (lib64) SEGK_LM_WAVE(0, false);
            else SEGK_LM_WAVE(1, false);
        }
#undef SEGK_LM_WAVE
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    DISPATCH_XT(c, {
        if (f->cov_type == 0) {
            if (lds > 48 * 1024)
                SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_assign<XT, 0>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((k_fbb_assign<XT, 0>), dim3(m.off[s_n]), dim3(nt_assign), lds, (hipStream_t)stream, *c, *f, *bt, m,
                               b, sweep, alpha, anneal_temp, new_tok, n_new, rcap, dbg, ll_mat, ll_ld, ctx->probe_ll, ctx->probe_ll_ld);
        } else {
            if (lds > 48 * 1024)
                SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_assign<XT, 1>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (f32) {
                if (lds > 48 * 1024)
                    SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_fbb_assign<XT, 1, 1>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_fbb_assign<XT, 1, 1>), dim3(m.off[s_n]), dim3(nt_assign), lds, (hipStream_t)stream, *c, *f, *bt, m,
                                   b, sweep, alpha, anneal_temp, new_tok, n_new, rcap, dbg, ll_mat, ll_ld, ctx->probe_ll, ctx->probe_ll_ld);
            } else
            hipLaunchKernelGGL((k_fbb_assign<XT, 1>), dim3(m.off[s_n]), dim3(nt_assign), lds, (hipStream_t)stream, *c, *f, *bt, m,
                               b, sweep, alpha, anneal_temp, new_tok, n_new, rcap, dbg, ll_mat, ll_ld, ctx->probe_ll, ctx->probe_ll_ld);
        }
    });
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_assign(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                        int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep, double anneal_temp,
                        const int32_t *new_tok, const int32_t *n_new, const float *ll_mat, int64_t ll_ld, void *stream)
{
    return fbb_assign_impl(ctx, c, f, bt, s_lo, s_n, b, n_utts, sweep, anneal_temp, new_tok, n_new, ll_mat, ll_ld, 0, stream);
}

int32_t segk_fbb_assign_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                               int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep, double anneal_temp,
                               const int32_t *new_tok, const int32_t *n_new, void *stream)
{
    SEGK_REQUIRE(f && f->cov_type == 1, "the float32 token likelihoods of this entry point are the diagonal (Student-t) ones");
    return fbb_assign_impl(ctx, c, f, bt, s_lo, s_n, b, n_utts, sweep, anneal_temp, new_tok, n_new, nullptr, 0, 1, stream);
}

int32_t segk_fbb_step_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                             int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep, int32_t n_slices_min,
                             int32_t n_slices_max, double wip, double time_power_term, double anneal_temp_fb,
                             double anneal_temp_am, double *score, uint8_t *boundaries, int32_t *new_tok, int32_t *n_new,
                             double *out_logprob, int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "context");
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(f->cov_type == 1, "the fused float32 Gibbs step is the diagonal (Student-t) one");
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1, "n_slices_min must be 0 or 1");
    SEGK_REQUIRE(score && boundaries && new_tok && n_new && out_logprob && status, "step operands");
    const char *env = getenv("SEGK_FBB_FUSED");
    if (env && atoi(env) == 0) { segk_set_error("segk_fbb_step_diag32: disabled (SEGK_FBB_FUSED=0)"); return SEGK_ERR_UNSUPPORTED; }
    if (f->lm_unigram || !bt->prior_rows || f->K_max > 256 || c->D > 256) {
        segk_set_error("segk_fbb_step_diag32: needs prior_rows, no language model, K_max <= 256, D <= 256");
        return SEGK_ERR_UNSUPPORTED;
    }
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_utts, 1);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    const int KM = f->K_max, D = c->D, NM = c->N_max, nw = 8;
    const int64_t triMax = (int64_t)NM * (NM + 1) / 2;
    // spans with an embedding per utterance at most: the band (an utterance longer than the window) or the triangle of a short one
    int64_t r_cap = triMax;
    if (c->band_ids && c->band_dur && n_slices_max > 0 && n_slices_max < NM && c->band_W == n_slices_max) {
        const int64_t t_short = (int64_t)n_slices_max * (n_slices_max + 1) / 2;
        r_cap = (int64_t)NM * c->band_W > t_short ? (int64_t)NM * c->band_W : t_short;
        if (r_cap > triMax) r_cap = triMax;
    }
    r_cap = (r_cap + 7) & ~(int64_t)7;
    const size_t lds = sizeof(double) * (size_t)(2 * triMax + 3 * NM + 2 + 2 * r_cap + (int64_t)nw * KM + KM) +
                       sizeof(float) * (size_t)(2 * (int64_t)D * KM + D * r_cap + r_cap * KM + 8 * r_cap) + 16 +
                       sizeof(int32_t) * (size_t)(triMax + 2 * r_cap + 3 * NM) + sizeof(short) * (size_t)((triMax + 3) & ~(int64_t)3) +
                       (size_t)((NM + 15) & ~15) + sizeof(short) * (size_t)KM;
    if (lds > 150 * 1024) {
        segk_set_error("segk_fbb_step_diag32: the slot tables and the logits of an utterance (%lld spans x %d slots) do not fit in LDS",
                       (long long)r_cap, KM);
        return SEGK_ERR_UNSUPPORTED;
    }
    FbbStepArgs A{};
    A.sweep = sweep; A.b = b; A.n_max = n_slices_max; A.r_cap = (int)r_cap;
    A.wip = wip; A.time_power_term = time_power_term; A.anneal_fb = anneal_temp_fb; A.anneal_am = anneal_temp_am;
    A.prior_alpha = f->alpha;
    A.score = score; A.boundaries = boundaries; A.new_tok = new_tok; A.n_new = n_new; A.out_logprob = out_logprob; A.status = status;
    A.probe_alpha = ctx->probe_alpha; A.probe_ll = ctx->probe_ll; A.probe_ld = ctx->probe_ll_ld;
    A.dbg = segk_dev_env("SEGK_STEP_DBG");
    const bool prof = segk_prof_now(ctx);
    const int slot = prof ? ctx->prof_n % SEGK_PROF_SLOTS : 0;
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], (hipStream_t)stream));
    DISPATCH_XT(c, {
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_fbb_step_diag32<XT>, lds));
        hipLaunchKernelGGL(k_fbb_step_diag32<XT>, dim3(m.off[s_n]), dim3(64 * nw), lds, (hipStream_t)stream, *c, *f, *bt, m, A);
    });
    if (prof) {
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], (hipStream_t)stream));
        ctx->prof_rows[slot] = m.off[s_n];             // (utterances: the rows are the caller's to count)
        ctx->prof_kind = 6;
        ctx->prof_launches = 1;
        ctx->prof_n++;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_set_probe(segk_ctx *ctx, double *alpha_out, double *ll_out, int64_t ll_ld)
{
    SEGK_REQUIRE(ctx, "context");
    SEGK_REQUIRE(!ll_out || ll_ld > 0, "leading dimension of the likelihood probe");
    ctx->probe_alpha = alpha_out;
    ctx->probe_ll = ll_out;
    ctx->probe_ll_ld = ll_out ? ll_ld : 0;
    return SEGK_OK;
}

int32_t segk_fbb_lm_apply(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t b,
                          int32_t sign, void *stream)
{
    (void)ctx;
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(f->lm_unigram && bt->lm_tok && bt->u_max > 0, "no language model attached");
    SEGK_REQUIRE(sign == 1 || sign == -1, "sign");
    const int64_t rows = (int64_t)bt->n_slices * bt->u_max;
    hipLaunchKernelGGL(k_fbb_lm_apply, dim3((unsigned)((rows + 127) / 128)), dim3(128), 0, (hipStream_t)stream, *f, *bt,
                       c->N_max, b, sign);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_lm_fill(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt, int32_t s_lo,
                         int32_t s_n, int32_t b, const int32_t *n_utts, const int32_t *new_tok, const int32_t *n_new,
                         void *stream)
{
    (void)ctx;
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    SEGK_REQUIRE(bt->lm_tok && bt->u_max > 0, "no transcript store");
    FbbMap m;
    rc = make_map(&m, s_lo, s_n, n_utts, 1);
    if (rc) return rc;
    if (m.off[s_n] == 0) return SEGK_OK;
    hipLaunchKernelGGL(k_fbb_lm_fill, dim3((m.off[s_n] + 127) / 128), dim3(128), 0, (hipStream_t)stream, *c, *bt, m, b,
                       new_tok, n_new);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_fbb_canonical(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, const segk_fbatch *bt, int32_t *remap,
                           void *stream)
{
    (void)ctx;
    int rc = check_fbb(c, f, bt);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fbb_remap, dim3(1), dim3(64), 0, st, *f, *bt, remap);
    hipLaunchKernelGGL(k_fbb_apply_remap, dim3((unsigned)((c->n_emb + 255) / 256)), dim3(256), 0, st, *f, *bt, c->n_emb,
                       remap);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

}  // extern "C"
