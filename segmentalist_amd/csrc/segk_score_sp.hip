// segk_score_sp.hip -- A1 filter on the 16-bit matrix pipe: fp16x2 / bf16x3 splits, log-sum-exp and matrix-output modes
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"

// ======================================================================================
// Split-precision filter (float32 data, 8 <= D <= 128).  f[k] = x.m_k - |m_k|^2/2 as above, but the
// contraction runs on the 16-bit matrix pipe (v_mfma_f32_32x32x16_{f16,bf16}: 16x the MAC rate of
// 32x32x2_f32) on exact or almost exact splits of the float32 operands.  Products of two 16-bit
// pieces are exact in float32.
//
//   P = 3, bf16x3:  x = x1 + x2 + x3 exactly (8 significand bits each).  Kept: m1.x1 in its own
//       chain (seeded with -|m|^2/2; KP/16 MFMAs, KP roundings at worst), the five products of level
//       2^-8 and 2^-16 in a second chain whose rounding error is negligible; dropped: the three of
//       level <= 2^-24, bounded by 2u |x| M.  Six MFMAs per k-step.
//   P = 2, fp16x2 (default):  x' = 2^a x, m' = 2^b m with powers of two chosen so that the largest
//       element sits in [2^12, 2^13) (exact scaling, well inside fp16's range; the means' exponent
//       follows max|m| at every prepare).  x' = x1 + 2^-11 x2 + r with x1 = fp16(x'),
//       x2 = fp16(2^11 (x' - x1)), |r| <= 2^-22 |x'| (two of the 24 significand bits are dropped).
//       Kept: m1.x1 (main chain) and m1.x2 + m2.x1 (second chain, carried at 2^11 times its weight so
//       that the small pieces stay normal numbers; multiplied by 2^-11 when read); dropped: m2.x2 and
//       the r terms, bounded by 9u |x| M.  Three MFMAs per k-step -- half the matrix work of bf16x3.
//       Scaled elements below 2^-14 (2^-26 of the largest one) are subnormal in fp16; even if the pipe
//       flushed them all to zero the error would be at most 2^-14 (sum_d |m'_d| + sum_d |x'_d|) <=
//       2^-14 sqrt(D) (M' + |x'|), i.e. (sqrt(D)/2) u |x| M relative to |x'| M' >= 2^12 max(|x'|, M'):
//       5.7u for D = 128.  Budget for P = 2: 9u + 5.7u -> 16u.
//
// The margin below which two filter values cannot be ordered (filter_tau_sp):
//     E1' = (1.02 (KP + 16) + 16 [P = 2]) u (|x| M + M^2/2)      (fp32 chain: (D4 + 3) u (...))
// with E2 (the reference's own rounding) unchanged -- the filter stays only a filter, every decision
// it cannot make with certainty goes to the exact stage.  tests/test_gpu_kmeans.py checks that the
// observed error stays under a quarter of E1'.
// Layouts: segk_internal.h.  Structure as k_kmeans_score: rows register-resident as the B operand
// (P pieces), component tiles double-buffered in LDS, two accumulator sets so that the top-2 update of
// tile t-1 drains under the MFMAs of tile t.
// ======================================================================================
// SPLIT = 1 (second stage of the pre-filter only): workgroup b takes row block b / n_chunks and the tile range
// [(b % n_chunks) * tiles_per_split, ...): the few thousand queued rows then occupy n_chunks times as many CUs for a
// n_chunks-th of the 32 dependent tile steps (~1.7 us each) a workgroup otherwise walks alone.  Partial candidates go
// to part_f ([row block][split][row]: top1, top2, component); the workgroup that arrives last at the block's ticket
// (part_k) merges them and runs the epilogue, and clears the ticket for the next call.
#ifndef SEGK_LSE_WAVES
#define SEGK_LSE_WAVES 4
#endif
#define SEGK_LSE_CHUNK 2            /* tiles per chunk of the log-sum-exp mode's association (see SEGK_LSE_CLOSE) */
template <int KS, int WAVES, int P, int MODE = 0, int SPLIT = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES >= 8 ? 1 : 2) void k_kmeans_score_sp(ScoreArgs A)
{
    typedef typename SegkPiece<P>::T T;
    typedef typename SegkPiece<P>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int32_t *__restrict__ ids = A.ids;
    const int64_t row0 = A.row0;
    int64_t n = A.n;
    const int64_t rb_first = SPLIT ? (int64_t)(blockIdx.x / A.n_chunks) : (int64_t)blockIdx.x;
    // row blocks per workgroup: one (the launch covers the rows), or -- the pre-filter's queue, whose length only the
    // device knows -- every (grid)th: a launch sized for the queue's capacity spent ~40 us of its 57 dispatching
    // eight thousand workgroups that found nothing to do
    const int64_t rb_stride = A.n_dev ? (int64_t)(SPLIT ? gridDim.x / A.n_chunks : gridDim.x) : ((int64_t)1 << 40);
    const int split = SPLIT ? (int)(blockIdx.x % A.n_chunks) : 0;
    const int tile0 = SPLIT ? split * A.tiles_per_split : 0;
    if (A.n_dev) {                        // rows queued by the pre-filter: the count lives on the device
        const int64_t nd = *A.n_dev - A.n_dev_off;
        n = nd < n ? nd : n;
        if (rb_first * WAVES * 32 >= n) return;
        // this launch is the head of the longer branch of the score stage (second stage -> full scan) and shares its CUs
        // with the dozen latency-bound waves of the exact pair kernel: its few waves go first at every issue slot
        if (A.dbg != 64) __builtin_amdgcn_s_setprio(3);
    }
    constexpr int KP = KS * 16;
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;      // floats per tile image
    const float *__restrict__ tiles = A.tiles + 1024 + (int64_t)tile0 * STRIDE;
    int nt_all = A.n_tiles;
    if (A.n_tiles_dev) {                  // (workgroup-uniform) the images' leading tiles alone hold components
        const int nd = *A.n_tiles_dev;
        nt_all = nd < nt_all ? nd : nt_all;
    }
    const int n_tiles = SPLIT ? (nt_all - tile0 < A.tiles_per_split ? nt_all - tile0 : A.tiles_per_split) : nt_all;
    if (n_tiles <= 0) return;
    const int D = A.D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // scaled domain: accumulators hold 2^(a+b) f (P = 2), unscaled again before anything leaves the kernel
    const int e_ab = ((const int *)A.X32)[1] + ((const int *)A.tiles)[0];
    const float unscale = ldexpf(1.f, -e_ab);
    constexpr float LS = P == 2 ? 1.f / 2048.f : 1.f;

    for (int64_t rblock = rb_first; rblock * WAVES * 32 < n; rblock += rb_stride) {
    V8 xb[P][KS];
    const int64_t r = (rblock * WAVES + wave) * 32 + j;
    int32_t rowid = -1;
    if (r < n) rowid = ids ? ids[r] : (int32_t)(row0 + r);
    {
        const int64_t plane = *reinterpret_cast<const int64_t *>((const unsigned char *)A.X32 + 16);      // elements per piece plane
        const T *xp = (const T *)((const unsigned char *)A.X32 + SEGK_SP_HEADER) + (int64_t)(rowid >= 0 ? rowid : 0) * KP + 8 * h;
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int s = 0; s < KS; s++) xb[p][s] = *reinterpret_cast<const V8 *>(xp + p * plane + 16 * s);
    }
    // MODE 1 (log-sum-exp, base 2, of the UNSCALED accumulator values): m1 / m2 are the running maximum
    // (finite start) and sum, as in k_kmeans_score
    float m1 = MODE == 1 ? -3.0e38f : NEG_INF_F, m2 = MODE == 1 ? 0.f : NEG_INF_F;
    int32_t irow = 0, itile = 0;
    // MODE 1: the log-sum-exp is associated in CHUNKS of SEGK_LSE_CHUNK tiles whoever computes them -- (maximum, sum) of a
    // chunk per lane half in tile order, the two halves merged, the chunks folded in order into (RM, RS) -- so that the rows
    // behind the last whole round of workgroups can be split over the tiles (SPLIT = 1: one chunk per workgroup, the last to
    // arrive folds them) and still give the bits of a row that one wave walked alone
    float RM = -3.0e38f, RS = 0.f;
#define SEGK_LSE_CLOSE()                                                                                        \
    do {                                                                                                        \
        const float om_ = __shfl_xor(m1, 32), os_ = __shfl_xor(m2, 32);                                         \
        const float Mc_ = fmaxf(m1, om_);                                                                       \
        const float Sc_ = m2 * __builtin_amdgcn_exp2f(m1 - Mc_) + os_ * __builtin_amdgcn_exp2f(om_ - Mc_);      \
        const float Mn_ = fmaxf(RM, Mc_);                                                                       \
        RS = RS * __builtin_amdgcn_exp2f(RM - Mn_) + Sc_ * __builtin_amdgcn_exp2f(Mc_ - Mn_);                   \
        RM = Mn_;                                                                                               \
        m1 = -3.0e38f;                                                                                          \
        m2 = 0.f;                                                                                               \
    } while (0)

    constexpr int PASS = WAVES * 256;
    constexpr int NPASS = (STRIDE + PASS - 1) / PASS;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // Staging by LDS-DMA issued from inline asm: hipcc counts a builtin global_load_lds as a pending LDS
    // write and drains it (s_waitcnt vmcnt(0)) before the next ds_read, which serialises the copy of
    // tile t+1 with the MFMAs of tile t.  The asm form is outside its bookkeeping; the wait is explicit,
    // once per tile, right before the barrier that hands the buffer over (cdna_hip_programming.md,
    // "Pipelining across barriers").  M0 carries the wave-uniform LDS byte address.
#define SEGK_STAGE(tt, buf)                                                                         \
    do {                                                                                            \
        const float *src_ = tiles + (int64_t)(tt) * STRIDE + tid * 4;                               \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + ((buf) * STRIDE + wave * 256) * 4); \
        _Pragma("unroll") for (int p = 0; p < NPASS; p++)                                           \
            if (p * PASS + wave * 256 < STRIDE) {                                                   \
                unsigned keep_;                                                                     \
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                 \
                             "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                  \
                             : "=&s"(keep_)                                                         \
                             : "v"(src_ + p * PASS), "s"(dst_ + p * PASS * 4)                       \
                             : "memory");                                                           \
            }                                                                                       \
    } while (0)
    // (a third LDS buffer with the copy of tile t+2 kept in flight across the barrier -- counted
    // vmcnt -- was measured 2 % slower: the copy already lands within one tile time)
#define SEGK_TILE_SYNC()                                                      \
    do {                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();                                         \
    } while (0)

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds);
    SEGK_STAGE(0, 0);
    SEGK_TILE_SYNC();

    f32x16 accAm, accAl, accBm, accBl;
#pragma unroll
    for (int q = 0; q < 16; q++) { accAm[q] = NEG_INF_F; accAl[q] = 0.f; accBm[q] = NEG_INF_F; accBl[q] = 0.f; }

    constexpr int VPS = (16 + KS - 1) / KS;
    float st4[4] = {0.f, 0.f, 0.f, 0.f};
    int dtile = -1;                       // MODE 2: the tile whose values are being drained
#define SEGK_DRAIN(ACCM, ACCL, vi)                                                    \
    do {                                                                              \
        float v_ = ACCM[(vi)] + ACCL[(vi)] * LS;                                      \
        if constexpr (MODE == 2) {      /* store the unscaled values: row-major [row][component] */ \
            st4[(vi) & 3] = v_ * unscale;                                             \
            if ((((vi) & 3) == 3) && dtile >= 0 && r < n)                             \
                *reinterpret_cast<float4 *>(A.mat_out + r * A.mat_ld + (dtile + tile0) * 32 + 4 * h + 8 * ((vi) >> 2)) = \
                    make_float4(st4[0], st4[1], st4[2], st4[3]);                      \
        } else if constexpr (MODE == 1) {                                                    \
            (void)v_;                  /* the tile's maximum is in m1 already (SEGK_LSE_PREP): one exponential per value */ \
            m2 += __builtin_amdgcn_exp2f(vv_[(vi)] - m1);                             \
        } else                                                                        \
        asm volatile("v_cmp_ngt_f32 vcc, %3, %0\n\t"                                  \
                     "v_med3_f32 %1, %0, %1, %3\n\t"                                  \
                     "v_max_f32 %0, %0, %3\n\t"                                       \
                     "v_cndmask_b32 %2, %4, %2, vcc"                                  \
                     : "+v"(m1), "+v"(m2), "+v"(irow)                                 \
                     : "v"(v_), "n"((vi))                                             \
                     : "vcc");                                                        \
    } while (0)

    // MODE 1, a finished tile's sixteen values per lane: unscaled (absent components: -3e38), their maximum folded into the
    // running one and the sum rescaled ONCE -- the drain proper is then one v_exp_f32 and one addition per value (two
    // exponentials and a multiply-add per value, the maximum updated value by value, took as many vector cycles as the tile's
    // MFMAs and did not hide behind them once only the occupied tiles are multiplied)
#define SEGK_LSE_PREP(ACCM, ACCL)                                                                     \
    do {                                                                                              \
        _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) vv_[q_] = fmaxf((ACCM[q_] + ACCL[q_] * LS) * unscale, -3.0e38f); \
        float g_ = fmaxf(vv_[0], vv_[1]);                                                             \
        _Pragma("unroll") for (int q_ = 2; q_ < 16; q_ += 2) g_ = fmaxf(g_, fmaxf(vv_[q_], vv_[q_ + 1])); \
        const float nm_ = fmaxf(m1, g_);                                                              \
        m2 *= __builtin_amdgcn_exp2f(m1 - nm_);                                                       \
        m1 = nm_;                                                                                     \
    } while (0)
#define SEGK_TILE(NEWM, NEWL, OLDM, OLDL, t_)                                                         \
    do {                                                                                              \
        float vv_[16];                                                                                \
        if constexpr (MODE == 1) SEGK_LSE_PREP(OLDM, OLDL);                                           \
        const float *Tt = lds + ((t_) & 1) * STRIDE;                                                  \
        const T *Tb = (const T *)Tt;                                                                  \
        if ((t_) + 1 < n_tiles) SEGK_STAGE((t_) + 1, ((t_) + 1) & 1);                                 \
        {                                                                                             \
            const float *cv = Tt + KS * P * 256 + 4 * h;                                              \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                           \
                float4 c4 = *reinterpret_cast<const float4 *>(cv + 8 * q);                            \
                NEWM[4 * q + 0] = c4.x; NEWM[4 * q + 1] = c4.y; NEWM[4 * q + 2] = c4.z; NEWM[4 * q + 3] = c4.w; \
                NEWL[4 * q + 0] = 0.f; NEWL[4 * q + 1] = 0.f; NEWL[4 * q + 2] = 0.f; NEWL[4 * q + 3] = 0.f;     \
            }                                                                                         \
        }                                                                                             \
        const float m1s = m1;                                                                         \
        dtile = (t_) - 1;                                                                             \
        V8 nx[P];                                                                                     \
        _Pragma("unroll") for (int p = 0; p < P; p++)                                                 \
            nx[p] = *reinterpret_cast<const V8 *>(Tb + (p * 64 + lane) * 8);                          \
        _Pragma("unroll") for (int s = 0; s < KS; s++) {                                              \
            V8 a[P];                                                                                  \
            _Pragma("unroll") for (int p = 0; p < P; p++) a[p] = nx[p];                               \
            if (s + 1 < KS) {          /* operands of the next k-step, in flight under this step's MFMAs */ \
                _Pragma("unroll") for (int p = 0; p < P; p++)                                         \
                    nx[p] = *reinterpret_cast<const V8 *>(Tb + (((s + 1) * P + p) * 64 + lane) * 8);  \
            }                                                                                         \
            NEWM = mfma_piece<P>(a[0], xb[0][s], NEWM);                                               \
            NEWL = mfma_piece<P>(a[0], xb[1][s], NEWL);                                               \
            NEWL = mfma_piece<P>(a[1], xb[0][s], NEWL);                                               \
            if constexpr (P == 3) {                                                                   \
                NEWL = mfma_piece<P>(a[1], xb[1][s], NEWL);                                           \
                NEWL = mfma_piece<P>(a[0], xb[P - 1][s], NEWL);                                       \
                NEWL = mfma_piece<P>(a[P - 1], xb[0][s], NEWL);                                       \
            }                                                                                         \
            _Pragma("unroll") for (int q = 0; q < VPS; q++)                                           \
                if (s * VPS + q < 16) SEGK_DRAIN(OLDM, OLDL, s * VPS + q);                            \
        }                                                                                             \
        itile = (m1 > m1s) ? ((t_) - 1) : itile;                                                      \
        if constexpr (MODE == 1)            /* tile t_ - 1 (global number tile0 + t_ - 1) closed a chunk */          \
            if ((t_) >= 1 && ((tile0 + (t_)) % SEGK_LSE_CHUNK) == 0) SEGK_LSE_CLOSE();                \
        SEGK_TILE_SYNC();                                                                             \
    } while (0)

    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
        SEGK_TILE(accAm, accAl, accBm, accBl, t);
        SEGK_TILE(accBm, accBl, accAm, accAl, t + 1);
    }
    {
        float m1s = m1;
        float vv_[16];
        if (t < n_tiles) {
            SEGK_TILE(accAm, accAl, accBm, accBl, t);
            m1s = m1;
            dtile = n_tiles - 1;
            if constexpr (MODE == 1) SEGK_LSE_PREP(accAm, accAl);
#pragma unroll
            for (int vi = 0; vi < 16; vi++) SEGK_DRAIN(accAm, accAl, vi);
        } else {
            dtile = n_tiles - 1;
            if constexpr (MODE == 1) SEGK_LSE_PREP(accBm, accBl);
#pragma unroll
            for (int vi = 0; vi < 16; vi++) SEGK_DRAIN(accBm, accBl, vi);
        }
        (void)vv_;
        itile = (m1 > m1s) ? (n_tiles - 1) : itile;
    }
#undef SEGK_TILE
#undef SEGK_LSE_PREP
#undef SEGK_DRAIN
#undef SEGK_STAGE
#undef SEGK_TILE_SYNC
    if constexpr (MODE == 2) continue;
    if constexpr (MODE == 1) {
        SEGK_LSE_CLOSE();                      // the last chunk (the two lane halves summed disjoint components of the same row)
        if constexpr (SPLIT) {
            // one chunk per workgroup: (maximum, sum) per row to part_f; the last to arrive folds the chunks in order
            __shared__ int s_last1;
            const int S_dev = (nt_all + A.tiles_per_split - 1) / A.tiles_per_split;
            float *part = A.part_f + ((int64_t)rblock * A.n_chunks * (WAVES * 32) + wave * 32 + j) * 2;
            if (h == 0) {
                float *pp = part + (int64_t)split * (WAVES * 32 * 2);
                __hip_atomic_store(pp + 0, RM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pp + 1, RS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_last1 = __hip_atomic_fetch_add(A.part_k + rblock, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S_dev - 1;
            __syncthreads();
            if (!s_last1) continue;
            if (tid == 0) __hip_atomic_store(A.part_k + rblock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            RM = -3.0e38f;
            RS = 0.f;
            for (int sp = 0; sp < S_dev; sp++) {
                const float *pp = part + (int64_t)sp * (WAVES * 32 * 2);
                const float Mc = __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float Sc = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float Mn = fmaxf(RM, Mc);
                RS = RS * __builtin_amdgcn_exp2f(RM - Mn) + Sc * __builtin_amdgcn_exp2f(Mc - Mn);
                RM = Mn;
            }
        }
        if (h == 0 && rowid >= 0) A.lse_out[rowid] = (double)(RM + log2f(RS)) * 0.6931471805599453 - A.lse_norm;
        continue;
    }
    const int32_t i1 = (itile + tile0) * 32 + 4 * h + (irow & 3) + 8 * (irow >> 2);
    const float o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
    const int oi = __shfl_xor(i1, 32);
    float top1 = fmaxf(m1, o1) * unscale;                          // powers of two: exact
    float top2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2)) * unscale;
    int idx = (o1 > m1 || (o1 == m1 && oi < i1)) ? oi : i1;
    if constexpr (SPLIT) {
        __shared__ int s_last;
        const int S = A.n_chunks;
        float *part = A.part_f + ((rblock * S) * (WAVES * 32) + wave * 32 + j) * 4;      // + split * WAVES * 32 * 4
        if (h == 0) {
            float *pp = part + (int64_t)split * (WAVES * 32 * 4);
            __hip_atomic_store(pp + 0, top1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, top2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 2, __int_as_float(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // No __threadfence here: at agent scope it writes back and invalidates the whole L2 of the XCD, and with the exact
        // stage writing its results beside this kernel every such fence cost ~1 us of everybody's time (634 us for the
        // stage).  The partials are write-through stores and coherent loads (agent-scope atomics, relaxed), ordered
        // against the ticket by waiting for the stores' acknowledgements.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(A.part_k + rblock, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1;
        __syncthreads();
        if (!s_last) continue;
        if (tid == 0) __hip_atomic_store(A.part_k + rblock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // per row the largest filter value (ties: the lower component) and the second largest over everything else
        top1 = NEG_INF_F; top2 = NEG_INF_F; idx = 0x7fffffff;
        for (int sp = 0; sp < S; sp++) {
            const float *pp = part + (int64_t)sp * (WAVES * 32 * 4);
            const float f1 = __hip_atomic_load(pp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float f2 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int k = __float_as_int(__hip_atomic_load(pp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (f1 > top1 || (f1 == top1 && k < idx)) {
                top2 = fmaxf(top2, top1);
                top1 = f1;
                idx = k;
            } else {
                top2 = fmaxf(top2, f1);
            }
            top2 = fmaxf(top2, f2);
        }
    }
    // Fused exact stage for the winner (D a multiple of 4): the reference's float32 -(deltas*deltas).sum()
    // in numpy's pairwise order.  This lane half owns the strided accumulators r_{4h..4h+3} in full
    // (segk_b3_dim); the row and the winner's mean are read as float32 from X32 / `means`.
    float sexact = __builtin_nanf("");
    if (A.fuse_exact)
        sexact = sp_exact_score<KS>(A.means32 + (int64_t)idx * D, A.xrows32 + (int64_t)(rowid >= 0 ? rowid : 0) * A.ld32, D, h);
    if (h == 0 && rowid >= 0) {
        A.cand.k[rowid] = idx;
        A.cand.f[2 * (int64_t)rowid + 0] = top1;
        A.cand.f[2 * (int64_t)rowid + 1] = top2;
        A.cand.s[rowid] = (double)sexact;                          // NaN when not fused: k_kmeans_exact_fill
        const float M = (float)(sqrt(*A.mnorm2) * (1.0 + 1e-6)) + 1e-30f;
        const float tau = filter_tau_sp(A.xnorm[rowid], M, D, P);
        if (!(top1 - top2 > tau)) {
            int q = atomicAdd(A.cand.count, 1);
            if (q < A.amb_cap) A.cand.queue[q] = rowid;
        }
    }
    }   // row blocks of this workgroup
}
#undef SEGK_LSE_CLOSE

// split-precision filter: whole rounds (and any larger remainder) to k_kmeans_score_sp, a remainder of
// fewer than SEGK_TAIL_QUEUE rows to the ambiguity queue.
template <int KS, int P>
static int launch_score_sp(segk_ctx *ctx, ScoreArgs A, hipStream_t st)
{
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    int wg_per_cu = 1;
    SEGK_CHECK_HIP(segk_occupancy((const void *)k_kmeans_score_sp<KS, 4, P>, 256, lds, &wg_per_cu));
    const int64_t slots = (int64_t)wg_per_cu * ctx->n_cu;
    const int64_t chunks = (A.n + 127) / 128;
    int64_t main_chunks = (chunks / slots) * slots;
    int64_t n_main = main_chunks * 128 < A.n ? main_chunks * 128 : A.n;
    if (A.n - n_main >= SEGK_TAIL_QUEUE) {            // a large remainder: one more (partial) round
        main_chunks = chunks;
        n_main = A.n;
    }
    if (main_chunks > 0) {
        ScoreArgs M = A;
        M.n = n_main;
        const bool prof = segk_prof_now(ctx);
        const int slot = ctx->prof_n % SEGK_PROF_SLOTS;
        if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, P>), dim3((unsigned)main_chunks), dim3(256), lds, st, M);
        if (prof) {
            SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
            ctx->prof_rows[slot] = n_main;
            ctx->prof_kind = P;
            ctx->prof_n++;
        }
    }
    if (A.n > n_main) {
        ScoreArgs T = A;
        T.n = A.n - n_main;
        T.row0 = A.row0 + n_main;
        T.ids = A.ids ? A.ids + n_main : nullptr;
        hipLaunchKernelGGL(k_score_queue_rows, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, st, T);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

template <int P>
static int dispatch_score_sp(segk_ctx *ctx, const ScoreArgs &A, int ks, hipStream_t st)
{
    switch (ks) {
        case 1: return launch_score_sp<1, P>(ctx, A, st);
        case 2: return launch_score_sp<2, P>(ctx, A, st);
        case 3: return launch_score_sp<3, P>(ctx, A, st);
        case 4: return launch_score_sp<4, P>(ctx, A, st);
        case 5: return launch_score_sp<5, P>(ctx, A, st);
        case 6: return launch_score_sp<6, P>(ctx, A, st);
        case 7: return launch_score_sp<7, P>(ctx, A, st);
        case 8: return launch_score_sp<8, P>(ctx, A, st);
        default: break;
    }
    segk_set_error("split-precision filter: D out of range");
    return SEGK_ERR_UNSUPPORTED;
}

template <int KS>
static int launch_score_lse_sp(segk_ctx *ctx, const ScoreArgs &A, hipStream_t st)
{
    constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    // waves per workgroup (row blocks of 32 LSE_W rows).  Eight -- one workgroup per CU, every tile staged once for 256 rows --
    // measured the same as four (82.6 against 80 us at configs[4]): the kernel waits for its rows at the start of every block
    // (SQ_WAIT_INST_ANY 44 % of the wave cycles, SQ_WAIT_INST_LDS 3 %, matrix pipe 41 % busy), not for the tiles
    constexpr int LSE_W = SEGK_LSE_WAVES;
    constexpr int RB = 32 * LSE_W;
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, LSE_W, 2, 1>, lds));
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, LSE_W, 2, 1, 1>, lds));
    const int64_t chunks = (A.n + RB - 1) / RB;
    const bool prof = ctx && segk_prof_now(ctx);
    const int slot = prof ? ctx->prof_n % SEGK_PROF_SLOTS : 0;
    // The row blocks behind the last whole round of workgroups (configs[4]: 1 024 blocks on 512 slots in six of the eight Gibbs
    // steps, 1 031 in the other two -- seven blocks would hold a third round of 38 us alone) go first, split over the tiles: one
    // chunk of SEGK_LSE_CHUNK tiles per workgroup, 11 us for all of them; the association of a row's sum is the same either way
    // (SEGK_LSE_CLOSE)
    int64_t rem = 0;
    if (ctx && ctx->n_cu > 0) {
        int wg_per_cu = 1;
        SEGK_CHECK_HIP(segk_occupancy((const void *)k_kmeans_score_sp<KS, LSE_W, 2, 1>, 64 * LSE_W, lds, &wg_per_cu));
        const int64_t slots = (int64_t)wg_per_cu * ctx->n_cu;
        const int64_t whole = (chunks / slots) * slots;
        if (whole > 0 && chunks - whole > 0 && chunks - whole <= slots / 8 && !ctx->capturing) rem = chunks - whole;
        const char *e = getenv("SEGK_LSE_SPLIT");               // 0: every row block by a workgroup of its own (same bits, the tests compare)
        if (e && atoi(e) == 0) rem = 0;
    }
    // (the partials' buffer is the second stage's, in its units: 8 x 128 x 4 floats and one ticket per block of 128 rows)
    const int64_t units = rem * (RB / 128);
    if (rem > 0 && ctx->sp2_blocks < units) {
        if (ctx->sp2_part) (void)hipFree(ctx->sp2_part);
        if (ctx->sp2_ticket) (void)hipFree(ctx->sp2_ticket);
        ctx->sp2_part = nullptr; ctx->sp2_ticket = nullptr; ctx->sp2_blocks = 0;
        const int64_t blocks = units > 64 ? units : 64;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->sp2_part, (size_t)blocks * 8 * 128 * 4 * sizeof(float)));
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->sp2_ticket, (size_t)blocks * sizeof(int32_t)));
        SEGK_CHECK_HIP(hipMemsetAsync(ctx->sp2_ticket, 0, (size_t)blocks * sizeof(int32_t), st));
        ctx->sp2_blocks = blocks;
    }
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    if (rem > 0) {
        const int64_t n_main = (chunks - rem) * RB;
        ScoreArgs S = A;
        S.n = A.n - n_main;
        S.row0 = A.row0 + n_main;
        S.ids = A.ids ? A.ids + n_main : nullptr;
        S.tiles_per_split = SEGK_LSE_CHUNK;
        S.n_chunks = (A.n_tiles + SEGK_LSE_CHUNK - 1) / SEGK_LSE_CHUNK;         // 16 at most (32 tiles): 32 floats per row of part_f
        S.part_f = ctx->sp2_part;
        S.part_k = ctx->sp2_ticket;
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, LSE_W, 2, 1, 1>), dim3((unsigned)(rem * S.n_chunks)), dim3(64 * LSE_W), lds, st, S);
        ScoreArgs M = A;
        M.n = n_main;
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, LSE_W, 2, 1>), dim3((unsigned)(chunks - rem)), dim3(64 * LSE_W), lds, st, M);
    } else
    hipLaunchKernelGGL((k_kmeans_score_sp<KS, LSE_W, 2, 1>), dim3((unsigned)chunks), dim3(64 * LSE_W), lds, st, A);
    if (prof) {
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = A.n;
        ctx->prof_kind = 4;
        ctx->prof_n++;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

template <int KS>
static int launch_score_mat_sp(const ScoreArgs &A, hipStream_t st, int tiles_hint)
{
    constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, 4, 2, 2>, lds));
    // few rows (the new tokens of one Gibbs block: ~10 k): the tiles of a row block over several workgroups -- every
    // workgroup writes its own columns of the matrix, nothing to merge (78 workgroups walking 32 tiles each: 54 us)
    const int64_t blocks = (A.n + 127) / 128;
    int n_split = blocks > 0 ? (int)(512 / blocks) : 1;
    if (n_split > 8) n_split = 8;
    if (n_split > A.n_tiles) n_split = A.n_tiles;
    // tiles that hold components (the packed images of the batch sampler: 14 of 32 at configs[4]), as the device last reported:
    // the split is over THOSE -- two splits of sixteen left one of them the fourteen and the other nothing
    const int t_used = tiles_hint > 0 && tiles_hint < A.n_tiles ? tiles_hint : A.n_tiles;
    if (n_split > t_used) n_split = t_used;
    if (n_split >= 2) {
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, 4, 2, 2, 1>, lds));
        ScoreArgs S = A;
        S.tiles_per_split = (t_used + n_split - 1) / n_split;
        S.n_chunks = (A.n_tiles + S.tiles_per_split - 1) / S.tiles_per_split;
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2, 2, 1>), dim3((unsigned)(blocks * S.n_chunks)), dim3(256), lds, st, S);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2, 2>), dim3((unsigned)blocks), dim3(256), lds, st, A);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// mat[r][k] = acc_k of row ids[r] (r < n), k < 32 n_tiles: the contraction itself, for callers that need
// every component's value (the token likelihoods of the batch sampler's assignment step)
int segk_launch_score_mat_sp(const void *ximg, int D2, const int32_t *ids, int64_t n, const float *tiles_sp, int n_tiles,
                             float *mat, int64_t mat_ld, void *stream, const int32_t *n_tiles_dev, int tiles_hint)
{
    if (n <= 0) return SEGK_OK;
    ScoreArgs A{};
    memset(&A, 0, sizeof(A));
    A.n_tiles_dev = n_tiles_dev;
    A.X32 = (const float *)ximg; A.ids = ids; A.row0 = 0; A.n = n;
    A.tiles = tiles_sp; A.n_tiles = n_tiles; A.tile_stride = segk_sp_tile_stride(D2, 2);
    A.D = D2;
    A.mat_out = mat; A.mat_ld = mat_ld;
    hipStream_t st = (hipStream_t)stream;
    switch (segk_b3_kp(D2) / 16) {
#define SEGK_CASE(k) \
    case k: return launch_score_mat_sp<k>(A, st, tiles_hint);
        SEGK_CASE(1) SEGK_CASE(2) SEGK_CASE(3) SEGK_CASE(4) SEGK_CASE(5) SEGK_CASE(6) SEGK_CASE(7) SEGK_CASE(8) SEGK_CASE(9)
        SEGK_CASE(10) SEGK_CASE(11) SEGK_CASE(12) SEGK_CASE(13)
#undef SEGK_CASE
        default: break;
    }
    segk_set_error("segk_launch_score_mat_sp: 2D=%d > 208 is not supported", D2);
    return SEGK_ERR_UNSUPPORTED;
}

// the log-sum-exp score on fp16x2 images: out[row] = ln sum_k 2^(acc_k) - norm, D2 <= 208
int segk_launch_score_lse_sp(segk_ctx *ctx, const void *ximg, int D2, const int32_t *ids, int64_t row0, int64_t n,
                             const float *tiles_sp, int n_tiles, double norm, double *out, void *stream, const int32_t *n_tiles_dev)
{
    if (n <= 0) return SEGK_OK;
    ScoreArgs A{};
    memset(&A, 0, sizeof(A));
    A.n_tiles_dev = n_tiles_dev;
    A.X32 = (const float *)ximg; A.ids = ids; A.row0 = row0; A.n = n;
    A.tiles = tiles_sp; A.n_tiles = n_tiles; A.tile_stride = segk_sp_tile_stride(D2, 2);
    A.D = D2;
    A.lse_out = out; A.lse_norm = norm;
    hipStream_t st = (hipStream_t)stream;
    switch (segk_b3_kp(D2) / 16) {
#define SEGK_CASE(k) \
    case k: return launch_score_lse_sp<k>(ctx, A, st);
        SEGK_CASE(1) SEGK_CASE(2) SEGK_CASE(3) SEGK_CASE(4) SEGK_CASE(5) SEGK_CASE(6) SEGK_CASE(7) SEGK_CASE(8) SEGK_CASE(9)
        SEGK_CASE(10) SEGK_CASE(11) SEGK_CASE(12) SEGK_CASE(13)
#undef SEGK_CASE
        default: break;
    }
    segk_set_error("segk_launch_score_lse_sp: 2D=%d > 208 is not supported", D2);
    return SEGK_ERR_UNSUPPORTED;
}

int segk_dispatch_score_sp(segk_ctx *ctx, const ScoreArgs &A, int ks, int pieces, hipStream_t st)
{
    return pieces == 2 ? dispatch_score_sp<2>(ctx, A, ks, st) : dispatch_score_sp<3>(ctx, A, ks, st);
}

// second stage of the one-product pre-filter (segk_score_h1.hip): all three fp16x2 products for the rows it
// queued; B.n_dev holds the row count on the device, the launch covers B.n rows
template <int KS>
static int launch_sp_second(segk_ctx *ctx, const ScoreArgs &B, hipStream_t st)
{
    constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds2 = 2 * (size_t)STRIDE * sizeof(float);
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, 4, 2>, lds2));
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score_sp<KS, 4, 2, 0, 1>, lds2));
    // ranges of component tiles per row block (1: the plain kernel), by the size of the launch: a workgroup walks the 32 tiles
    // of its row block in ~55 us however few blocks there are, so that short queues (a multi-GPU shard) gain from the split
    // although the stage is matrix-bound on the whole corpus (1.05 M rows: 1 588 unsplit against 1 500 split); two workgroups
    // (row blocks in flight) per CU
    const int64_t per_cu = ctx->n_cu > 0 ? B.n / ctx->n_cu : B.n;
    int n_split = per_cu <= 1536 ? 4 : per_cu <= 3072 ? 2 : 1;
    if (n_split > B.n_tiles) n_split = B.n_tiles;
    if (n_split > 8) n_split = 8;
    const int64_t blocks = (B.n + 127) / 128;
    int64_t grid = 2 * (int64_t)ctx->n_cu;
    if (grid > blocks) grid = blocks;
    if (grid < 1) grid = 1;
    if (n_split <= 1 || ctx->capturing) {
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2>), dim3((unsigned)grid), dim3(256), lds2, st, B);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    if (ctx->sp2_blocks < blocks) {
        if (ctx->sp2_part) (void)hipFree(ctx->sp2_part);
        if (ctx->sp2_ticket) (void)hipFree(ctx->sp2_ticket);
        ctx->sp2_part = nullptr; ctx->sp2_ticket = nullptr; ctx->sp2_blocks = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->sp2_part, (size_t)blocks * 8 * 128 * 4 * sizeof(float)));
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->sp2_ticket, (size_t)blocks * sizeof(int32_t)));
        SEGK_CHECK_HIP(hipMemsetAsync(ctx->sp2_ticket, 0, (size_t)blocks * sizeof(int32_t), st));
        ctx->sp2_blocks = blocks;
    }
    ScoreArgs S = B;
    S.tiles_per_split = (B.n_tiles + n_split - 1) / n_split;
    S.n_chunks = (B.n_tiles + S.tiles_per_split - 1) / S.tiles_per_split;
    S.part_f = ctx->sp2_part;
    S.part_k = ctx->sp2_ticket;
    hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2, 0, 1>), dim3((unsigned)(grid * S.n_chunks)), dim3(256), lds2, st, S);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int segk_launch_sp_second(segk_ctx *ctx, const ScoreArgs &B, int ks, hipStream_t st)
{
    switch (ks) {
        case 1: return launch_sp_second<1>(ctx, B, st);
        case 2: return launch_sp_second<2>(ctx, B, st);
        case 3: return launch_sp_second<3>(ctx, B, st);
        case 4: return launch_sp_second<4>(ctx, B, st);
        case 5: return launch_sp_second<5>(ctx, B, st);
        case 6: return launch_sp_second<6>(ctx, B, st);
        case 7: return launch_sp_second<7>(ctx, B, st);
        case 8: return launch_sp_second<8>(ctx, B, st);
        default: break;
    }
    segk_set_error("pre-filter second stage: D out of range");
    return SEGK_ERR_UNSUPPORTED;
}
