// segk_kmeans.hip -- gfx950 kernels of the segmental k-means hot path.
//
//   k_corpus_prepare      X -> X32 (zero padded) + row norms
//   k_kmeans_prepare      means -> MFMA operand tiles, -|m|^2/2, max norm
//   k_kmeans_score        A1 filter: fp32 MFMA (v_mfma_f32_32x32x2_f32) X32 . means^T with a fused
//                         running top-2 / argmax per embedding
//   k_kmeans_segment      per utterance: exact A1 of the candidates (reference arithmetic),
//                         A5 vector, A8 max-plus DP, new tokens + their argmax components
//   k_kmeans_update_utt   A11 sequential del/add/clean for one utterance (reference order)
//   k_kmeans_batch_*      A11 batch-synchronous statistics (fixed summation tree)
//
// Compiled with -ffp-contract=off: the exact stage must round every operation separately,
// as numpy does (DESIGN.md "bit-exact contract").
#include <stdlib.h>

#include "segk_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NEG_INF_D (-__builtin_huge_val())
#define NEG_INF_F (-__builtin_huge_valf())

// ======================================================================================
// Exact stage: numpy's pairwise summation of (m[d]-x[d])^2, identical evaluation order
// (kmeans_components.py:225-226; numpy pairwise_sum: n<8 sequential, n<=128 eight strided
// accumulators + fixed tree + sequential tail, n>128 split at n/2 rounded down to 8).
// ======================================================================================
template <typename T, typename TM, typename TX>
__device__ __forceinline__ T sqd(const TM &m, const TX *x, int d)
{
    T delta = (T)m[d] - (T)x[d];
    return delta * delta;
}

// TM: anything indexable (`const float*`, `const double*`, TileRow)
template <typename T, typename TM, typename TX>
__device__ T pw_base(const TM &m, const TX *x, int n)
{
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; i++) res += sqd<T>(m, x, i);
        return res;
    }
    T r0 = sqd<T>(m, x, 0), r1 = sqd<T>(m, x, 1), r2 = sqd<T>(m, x, 2), r3 = sqd<T>(m, x, 3);
    T r4 = sqd<T>(m, x, 4), r5 = sqd<T>(m, x, 5), r6 = sqd<T>(m, x, 6), r7 = sqd<T>(m, x, 7);
    int i;
    const int nfull = n - (n % 8);
    for (i = 8; i < nfull; i += 8) {
        r0 += sqd<T>(m, x, i + 0);
        r1 += sqd<T>(m, x, i + 1);
        r2 += sqd<T>(m, x, i + 2);
        r3 += sqd<T>(m, x, i + 3);
        r4 += sqd<T>(m, x, i + 4);
        r5 += sqd<T>(m, x, i + 5);
        r6 += sqd<T>(m, x, i + 6);
        r7 += sqd<T>(m, x, i + 7);
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += sqd<T>(m, x, i);
    return res;
}

// -sum_d (m[d]-x[d])^2 in the dtype T of the reference's `means`/X.  Offsets into m are
// multiples of 8 (numpy's split points), which TileRow::operator+ relies on.
template <typename T, typename TM, typename TX>
__device__ T neg_sqd_exact(const TM &m, const TX *x, int n)
{
    if (n <= 128) return -pw_base<T>(m, x, n);
    struct Frame { int off, n, state; T left; };
    Frame st[28];
    int sp = 0;
    st[0].off = 0; st[0].n = n; st[0].state = 0; st[0].left = (T)0;
    T ret = (T)0;
    while (sp >= 0) {
        Frame &f = st[sp];
        if (f.state == 0) {
            if (f.n <= 128) {
                ret = pw_base<T>(m + f.off, x + f.off, f.n);
                sp--;
            } else {
                int n2 = f.n / 2;
                n2 -= n2 % 8;
                f.state = 1;
                st[sp + 1].off = f.off; st[sp + 1].n = n2; st[sp + 1].state = 0;
                sp++;
            }
        } else if (f.state == 1) {
            f.left = ret;
            f.state = 2;
            int n2 = f.n / 2;
            n2 -= n2 % 8;
            st[sp + 1].off = f.off + n2; st[sp + 1].n = f.n - n2; st[sp + 1].state = 0;
            sp++;
        } else {
            ret = f.left + ret;
            sp--;
        }
    }
    return -ret;
}

// Four rows at once for 8 <= n <= 128 (numpy's single-block case): identical arithmetic per
// row, interleaved so that 4 x 8 loads are in flight per step.
template <typename T, typename TM, typename TX>
__device__ void neg_sqd_exact_x4(const TM &m0, const TM &m1, const TM &m2, const TM &m3, const TX *x, int n, T *out)
{
    T r[4][8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        r[0][j] = sqd<T>(m0, x, j);
        r[1][j] = sqd<T>(m1, x, j);
        r[2][j] = sqd<T>(m2, x, j);
        r[3][j] = sqd<T>(m3, x, j);
    }
    int i;
    const int nfull = n - (n % 8);
    for (i = 8; i < nfull; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            r[0][j] += sqd<T>(m0, x, i + j);
            r[1][j] += sqd<T>(m1, x, i + j);
            r[2][j] += sqd<T>(m2, x, i + j);
            r[3][j] += sqd<T>(m3, x, i + j);
        }
    }
    T res[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        res[q] = ((r[q][0] + r[q][1]) + (r[q][2] + r[q][3])) + ((r[q][4] + r[q][5]) + (r[q][6] + r[q][7]));
    for (; i < n; i++) {
        res[0] += sqd<T>(m0, x, i);
        res[1] += sqd<T>(m1, x, i);
        res[2] += sqd<T>(m2, x, i);
        res[3] += sqd<T>(m3, x, i);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) out[q] = -res[q];
}

// A component's row read from the MFMA tile image instead of from `means`: the image holds
// the same float32 values with the component index contiguous (stride 2 floats), so that
// consecutive lanes scanning consecutive components touch a few cache lines per load instead
// of one line per lane.  Only valid when the means are float32 (the image is a float copy).
struct TileRow {
    const float *base;      // tiles + tile*stride + 2*(k & 31)
    __device__ __forceinline__ float operator[](int d) const
    {
        return base[(d >> 2) * 128 + ((d >> 1) & 1) * 64 + (d & 1)];
    }
    __device__ __forceinline__ TileRow operator+(int off) const { return TileRow{base + (off >> 2) * 128}; }
};
__device__ __forceinline__ TileRow tile_row(const float *tiles, int tile_stride, int k)
{
    return TileRow{tiles + (int64_t)(k >> 5) * tile_stride + 2 * (k & 31)};
}

// Margin below which two fp32-filter values cannot be ordered with certainty
// (DESIGN.md "filter margin"): tau = 1.25 * (2*E1 + E2) where
//   E1 = (D4+3) u (|x| M + M^2/2)          fp32 fma chain of the MFMA + operand rounding
//   E2 = c2 u (|x| + M)^2                   rounding of the REFERENCE's own float32 evaluation
// (E2 ~ 0 when the reference computes in float64).  u = 2^-24.
__device__ __forceinline__ float filter_tau(float xn, float M, int D, int is_f64)
{
    const float u = 5.9604645e-8f;
    const int D4 = (D + 3) & ~3;
    float e1 = (float)(D4 + 3 + (is_f64 ? 4 : 0)) * u * (xn * M + 0.5f * M * M);
    int levels = 0;
    for (int n = D; n > 128; n = (n + 1) / 2) levels++;
    int deff = D < 128 ? D : 128;
    float c2 = is_f64 ? 1e-6f : (float)(deff / 8 + 13 + 2 * levels);
    float s = xn + M;
    float e2 = c2 * u * s * s;
    return 1.25f * (2.0f * e1 + e2) + 1e-37f;
}

// ======================================================================================
// corpus prepare: X (f32/f64, ldx) -> X32 [n_emb, ld32] zero padded, xnorm upper bound
// ======================================================================================
template <typename XT>
__global__ void k_corpus_prepare(const XT *X, int64_t ldx, int64_t n_emb, int D, int64_t ld32,
                                 float *X32, float *xnorm)
{
    int64_t e = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (e >= n_emb) return;
    int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int d = lane; d < (int)ld32; d += 64) {
        float v = 0.f;
        if (d < D) {
            XT xv = X[e * ldx + d];
            v = (float)xv;
            s += (double)xv * (double)xv;
        }
        if (X32) X32[e * ld32 + d] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) xnorm[e] = (float)(sqrt(s) * (1.0 + 1e-6)) + 1e-30f;
}

// ======================================================================================
// means -> tiles
// ======================================================================================
// value hash of one element of a row of `means` (k_kmeans_mark_dups): -0 and +0 hash alike, the per-element
// terms add up commutatively, so lanes can hash strided parts of a row and sum
__device__ __forceinline__ unsigned long long segk_elem_hash(double v, int d)
{
    unsigned long long z = (unsigned long long)__double_as_longlong(v + 0.0) + 0x9E3779B97F4A7C15ull * (unsigned long long)(d + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <typename XT>
__global__ void k_kmeans_prepare(const XT *means, int K_max, int D, float *tiles,
                                 unsigned long long *mnorm2_bits, unsigned int *zero_slot, unsigned long long *row_hash)
{
    const int tile = blockIdx.x;
    if (zero_slot && tile == 0 && threadIdx.x == 0) *zero_slot = 0u;     // E_m of the fp16 tile image: k_kmeans_prepare_sp, next on the stream
    const int G = segk_gmax(D);          // bucket extent; dims >= D are zero filled
    const int stride = segk_tile_stride(D);
    float *T = tiles + (int64_t)tile * stride;
    __shared__ double nrm[32];
    // |m|^2 of the tile's 32 components: 8 lanes per component, fp64
    {
        const int ci = threadIdx.x >> 3, sub = threadIdx.x & 7;      // 256 threads = 32 x 8
        const int comp = tile * 32 + ci;
        double s = 0.0;
        unsigned long long hh = 0ull;
        if (comp < K_max)
            for (int d = sub; d < D; d += 8) {
                double v = (double)means[(int64_t)comp * D + d];
                s += v * v;
                hh += segk_elem_hash(v, d);
            }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        hh += __shfl_xor(hh, 1);
        hh += __shfl_xor(hh, 2);
        hh += __shfl_xor(hh, 4);
        if (sub == 0) {
            nrm[ci] = s;
            if (comp < K_max) atomicMax(mnorm2_bits, (unsigned long long)__double_as_longlong(s));
            if (row_hash && comp < K_max) row_hash[comp] = hh | 1ull;      // never 0: the empty key of the hash table
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < stride; idx += blockDim.x) {
        float v = 0.f;
        if (idx < G * 128) {
            int g = idx >> 7, rem = idx & 127, lane = rem >> 1, s = rem & 1;
            int comp = tile * 32 + (lane & 31);
            int d = 4 * g + 2 * (lane >> 5) + s;
            if (comp < K_max && d < D) v = (float)means[(int64_t)comp * D + d];
        } else if (idx < G * 128 + 32) {
            int i = idx - G * 128;
            int comp = tile * 32 + i;
            v = (comp < K_max) ? (float)(-0.5 * nrm[i]) : -3.0e38f;
        }
        T[idx] = v;
    }
}

// single-instruction max (fmaxf() makes hipcc add a canonicalising v_max on MFMA outputs)
__device__ __forceinline__ float vmax_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ======================================================================================
// A1 filter: fused fp32 MFMA contraction + running top-2/argmax.
//   workgroup = 4 waves; wave w owns NB blocks of 32 embeddings whose X32 rows live in
//   registers for the whole kernel as the MFMA B operand (lane (j,h): dims 4g+2h+{0,1});
//   the 32-component tiles of the means stream through a double-buffered LDS image and are
//   the A operand, so the 32x32 accumulator has the component on the register index and the
//   embedding on the lane: the running max over components is lane-local.
//   Accumulators start at -|m|^2/2, so acc = x.m - |m|^2/2 with no epilogue arithmetic.
// ======================================================================================
#define SEGK_PAIR_PENDING 0x40000000      /* cand.k: pair (c, c + 1) named by the pre-filter, member not yet chosen */
struct ScoreArgs {
    const float *X32;
    int64_t ld32;
    const int32_t *ids;
    int64_t row0, n;
    const float *tiles;
    int n_tiles, tile_stride, G /* groups present in X32 rows */, D, fuse_exact, is_f64;
    int dbg;                     /* timing-only ablation bits, 0 in production */
    const float *xnorm;
    const double *mnorm2;
    segk_cand cand;
    int amb_cap;
    // split-K launch (SPLIT = 1): workgroup b scores chunk b % n_chunks against the tiles
    // [(b / n_chunks) * tiles_per_split, ...) and writes its partial candidates to part_k / part_f
    int n_chunks, tiles_per_split;
    int32_t *part_k;
    float *part_f;
    // MODE = 1 (log-sum-exp over the components instead of the top-2): out[row] = ln2 * log2 sum_k 2^acc - lse_norm
    double *lse_out;
    double lse_norm;
    const float *means32;        /* split-precision filter: float32 `means` and rows for the fused exact score */
    const float *xrows32;
    float *mat_out;              /* MODE 2: the accumulator values themselves, [n rows][mat_ld], mat_ld >= 32 n_tiles */
    int64_t mat_ld;
    // one-product pre-filter (k_kmeans_score_h1): its undecided rows go to pre_queue (pre_cap entries, then to
    // cand.queue); the split-precision kernel that follows reads its row count from n_dev
    const int32_t *n_dev;
    int32_t *pre_queue, *pre_count;
    int pre_cap, K_max;
    const float *xerr;           /* pre-filter: |x - x1| per row (k_corpus_resid_sp) */
    unsigned long long *stamp;   /* -DSEGK_STAMP development builds: s_memtime at phase boundaries, 8 per workgroup */
};

// SPLIT = 0: the whole component range per workgroup, winner + margin test + fused exact score.
// SPLIT = 1: the tail of a launch whose last round would leave most of the chip idle (or a launch
//            smaller than one round): every 32*NB*WAVES-row chunk is scored by several workgroups,
//            each against a slice of the component tiles; k_score_merge combines the partial top-2.
// MODE = 0: running top-2 / argmax (the k-means filter).
// MODE = 1: online log-sum-exp of the accumulator values, base 2 (the operands are pre-scaled by
//           log2 e): the span score of the fixed-variance FBGMM batch sampler, whose logit is a
//           contraction of [x^2, x] with per-component [-pp/2, pp*mu] plus a constant
//           (segk_fbbatch.hip k_fbb_tiles32); within the 1e-4 contract of that path.
template <int GMAX, int NB, int WAVES, int SPLIT, int MODE = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 2) void k_kmeans_score(ScoreArgs A)
{
    static_assert(!(MODE == 1 && SPLIT == 1), "the log-sum-exp mode has no split-K variant");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float *__restrict__ X32 = A.X32;
    const int64_t ld32 = A.ld32;
    const int32_t *__restrict__ ids = A.ids;
    const int64_t row0 = A.row0, n = A.n;
    const int tile_stride = A.tile_stride, G = A.G, D = A.D, fuse_exact = A.fuse_exact, dbg = A.dbg;
    const int chunk = SPLIT ? (int)(blockIdx.x % A.n_chunks) : (int)blockIdx.x;
    const int split = SPLIT ? (int)(blockIdx.x / A.n_chunks) : 0;
    const int tile0 = SPLIT ? split * A.tiles_per_split : 0;
    const int n_tiles = SPLIT ? (A.n_tiles - tile0 < A.tiles_per_split ? A.n_tiles - tile0 : A.tiles_per_split) : A.n_tiles;
    const float *__restrict__ tiles = A.tiles + (int64_t)tile0 * tile_stride;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    float2 xb[NB][GMAX];
    int32_t rowid[NB];
    const int64_t base = ((int64_t)chunk * WAVES + wave) * (32 * NB);
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        int64_t r = base + nb * 32 + j;
        int32_t id = -1;
        if (r < n) id = ids ? ids[r] : (int32_t)(row0 + r);
        rowid[nb] = id;
        const float *xp = X32 + (int64_t)(id >= 0 ? id : 0) * ld32 + 2 * h;
        // unconditional loads (a select around a load makes hipcc branch and wait per element):
        // groups beyond the row's G are read from a clamped in-row offset and zeroed afterwards
#pragma unroll
        for (int g = 0; g < GMAX; g++) xb[nb][g] = *reinterpret_cast<const float2 *>(xp + 4 * (g < G ? g : 0));
        if (G < GMAX) {
#pragma unroll
            for (int g = 0; g < GMAX; g++)
                if (g >= G) xb[nb][g] = make_float2(0.f, 0.f);
        }
    }
    // running top-2 values, and the argmax as (tile, row code) -- per lane
    float m1[NB], m2[NB];
    int32_t irow[NB], itile[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        // MODE 1 reuses m1 / m2 as the running maximum (finite start: -inf - -inf would be NaN) and sum
        m1[nb] = MODE == 1 ? -3.0e38f : NEG_INF_F;
        m2[nb] = MODE == 1 ? 0.f : NEG_INF_F;
        irow[nb] = 0;
        itile[nb] = 0;
    }

    constexpr int STRIDE = (GMAX * 128 + 32 + 1023) / 1024 * 1024;   // == tile_stride (segk_tile_stride)
    constexpr int PASS = WAVES * 256;                                 // floats moved per pass by the workgroup
    constexpr int NPASS = (STRIDE + PASS - 1) / PASS;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // stage tile `tt` into LDS buffer `buf` with direct global->LDS loads: one wave instruction
    // moves 64 x 16 B = 1 KiB to a wave-uniform base + lane*16, i.e. a straight copy of the image
    // LDS-DMA issued from inline asm (see k_kmeans_score_b3): outside hipcc's waitcnt bookkeeping, so the
    // copy of tile t+1 is not drained before the ds_reads of tile t; explicit wait before the barrier.
#define SEGK_STAGE(tt, buf)                                                                         \
    do {                                                                                            \
        const float *src_ = tiles + (int64_t)(tt) * tile_stride + tid * 4;                          \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + ((buf) * tile_stride + wave * 256) * 4); \
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
#define SEGK_TILE_SYNC()                                                      \
    do {                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();                                         \
    } while (0)

    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds);
    SEGK_STAGE(0, 0);
    SEGK_TILE_SYNC();

    // Software pipeline over the component tiles with two accumulator sets: while the MFMAs of
    // tile t fill one set, the top-2/argmax update (VALU) of tile t-1 drains the other, a slice
    // per k-step, so that the matrix and vector pipes overlap inside one wave.  The drained set
    // starts at -inf, which makes the first drain a no-op.
    f32x16 accA[NB], accB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) { accA[nb][r] = NEG_INF_F; accB[nb][r] = NEG_INF_F; }

    constexpr int VPS = (16 * NB + GMAX - 1) / GMAX;     // drained values per k-step

    // One drained value = one asm statement of 4 VALU instructions, so that the compiler can
    // neither sink it out of its k-step nor split it.  Order: the compare and the median read
    // the OLD running maximum; two instructions separate v_cmp (writes VCC) from v_cndmask
    // (reads VCC), which covers the 2 wait states gfx950 needs there.
    //   vcc   = !(v > m1);  m2 = med3(m1, m2, v);  m1 = max(m1, v);  irow = vcc ? irow : code
#define SEGK_DRAIN(ACC, vi)                                                           \
    do {                                                                              \
        const int nb_ = (vi) >> 4;                                                    \
        if constexpr (MODE == 1) {                                                    \
            /* nm = max(mx, v); sm = sm * 2^(mx - nm) + 2^(v - nm); mx = nm */         \
            const float v_ = ACC[nb_][(vi) & 15];                                     \
            const float nm_ = vmax_f32(m1[nb_], v_);                                  \
            m2[nb_] = m2[nb_] * __builtin_amdgcn_exp2f(m1[nb_] - nm_) + __builtin_amdgcn_exp2f(v_ - nm_); \
            m1[nb_] = nm_;                                                            \
        } else                                                                        \
        asm volatile("v_cmp_ngt_f32 vcc, %3, %0\n\t"                                  \
                     "v_med3_f32 %1, %0, %1, %3\n\t"                                  \
                     "v_max_f32 %0, %0, %3\n\t"                                       \
                     "v_cndmask_b32 %2, %4, %2, vcc"                                  \
                     : "+v"(m1[nb_]), "+v"(m2[nb_]), "+v"(irow[nb_])                  \
                     : "v"(ACC[nb_][(vi) & 15]), "n"((vi) & 15)                       \
                     : "vcc");                                                        \
    } while (0)

#define SEGK_TILE(ACC_NEW, ACC_OLD, t_)                                                               \
    do {                                                                                              \
        const float *T = lds + ((t_) & 1) * tile_stride;                                              \
        /* 18 wait states between the last MFMA that wrote ACC_OLD and its first VALU reader */       \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 1" ::: "memory");                                  \
        if ((t_) + 1 < n_tiles && !(dbg & 8)) SEGK_STAGE((t_) + 1, ((t_) + 1) & 1);               \
        {                                                                                             \
            const float *cv = T + GMAX * 128 + 4 * h;                                                 \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                           \
                float4 c4 = *reinterpret_cast<const float4 *>(cv + 8 * q);                            \
                _Pragma("unroll") for (int nb = 0; nb < NB; nb++) {                                   \
                    ACC_NEW[nb][4 * q + 0] = c4.x;                                                    \
                    ACC_NEW[nb][4 * q + 1] = c4.y;                                                    \
                    ACC_NEW[nb][4 * q + 2] = c4.z;                                                    \
                    ACC_NEW[nb][4 * q + 3] = c4.w;                                                    \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
        float m1s[NB];                                                                                \
        _Pragma("unroll") for (int nb = 0; nb < NB; nb++) m1s[nb] = m1[nb];                           \
        float2 a_cur = *reinterpret_cast<const float2 *>(T + lane * 2);                               \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int g = 0; g < GMAX; g++) {                                            \
            float2 a_nxt = a_cur;                                                                     \
            if (g + 1 < GMAX) a_nxt = *reinterpret_cast<const float2 *>(T + ((g + 1) * 64 + lane) * 2); \
            _Pragma("unroll") for (int nb = 0; nb < NB; nb++)                                         \
                ACC_NEW[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, xb[nb][g].x, ACC_NEW[nb], 0, 0, 0); \
            _Pragma("unroll") for (int nb = 0; nb < NB; nb++)                                         \
                ACC_NEW[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, xb[nb][g].y, ACC_NEW[nb], 0, 0, 0); \
            if (!(dbg & 4)) {                                                                     \
                _Pragma("unroll") for (int q = 0; q < VPS; q++)                                       \
                    if (g * VPS + q < 16 * NB) SEGK_DRAIN(ACC_OLD, g * VPS + q);                      \
            }                                                                                         \
            a_cur = a_nxt;                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
        /* the drained tile was t-1: fix up the tile id where the maximum moved */                    \
        _Pragma("unroll") for (int nb = 0; nb < NB; nb++)                                             \
            itile[nb] = (m1[nb] > m1s[nb]) ? ((t_) - 1) : itile[nb];                                  \
        SEGK_TILE_SYNC();                                                                         \
    } while (0)

    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
        SEGK_TILE(accA, accB, t);
        SEGK_TILE(accB, accA, t + 1);
    }
    {
        float m1s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) m1s[nb] = m1[nb];
        if (t < n_tiles) {
            SEGK_TILE(accA, accB, t);
#pragma unroll
            for (int nb = 0; nb < NB; nb++) m1s[nb] = m1[nb];
#pragma unroll
            for (int vi = 0; vi < 16 * NB; vi++) SEGK_DRAIN(accA, vi);     // last tile, held by accA
        } else {
#pragma unroll
            for (int vi = 0; vi < 16 * NB; vi++) SEGK_DRAIN(accB, vi);     // last tile, held by accB
        }
#pragma unroll
        for (int nb = 0; nb < NB; nb++) itile[nb] = (m1[nb] > m1s[nb]) ? (n_tiles - 1) : itile[nb];
    }
#undef SEGK_TILE
#undef SEGK_DRAIN
#undef SEGK_TILE_SYNC
#undef SEGK_STAGE
    // component index of (tile, row code) on this lane half
    int32_t i1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) i1[nb] = itile[nb] * 32 + 4 * h + (irow[nb] & 3) + 8 * (irow[nb] >> 2);

    // the two lane halves hold disjoint component subsets of the same embedding
    const int nb8 = D >> 3;            // full blocks of 8 dims (numpy's strided accumulators)
    const int rem = D & 7;             // sequential tail
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        if constexpr (MODE == 1) {
            // the two lane halves summed disjoint component subsets of the same row
            const float om = __shfl_xor(m1[nb], 32), os = __shfl_xor(m2[nb], 32);
            const float M = fmaxf(m1[nb], om);
            const float S = m2[nb] * exp2f(m1[nb] - M) + os * exp2f(om - M);
            if (h == 0 && rowid[nb] >= 0)
                A.lse_out[rowid[nb]] = (double)(M + log2f(S)) * 0.6931471805599453 - A.lse_norm;
            continue;
        }
        float o1 = __shfl_xor(m1[nb], 32), o2 = __shfl_xor(m2[nb], 32);
        int oi = __shfl_xor(i1[nb], 32);
        float top1 = fmaxf(m1[nb], o1);
        float top2 = fmaxf(fminf(m1[nb], o1), fmaxf(m2[nb], o2));
        int idx = (o1 > m1[nb] || (o1 == m1[nb] && oi < i1[nb])) ? oi : i1[nb];
        if (SPLIT) {
            const int64_t r = base + nb * 32 + j;
            if (h == 0 && r < n) {
                const int64_t e = (int64_t)split * n + r;
                A.part_k[e] = idx + tile0 * 32;
                A.part_f[2 * e + 0] = top1;
                A.part_f[2 * e + 1] = top2;
            }
            continue;
        }
        // Fused exact stage for the winner (float32 data, 8 <= D <= 128): the reference's
        // float32 -(deltas*deltas).sum() in numpy's pairwise order.  Dim d = 4g+2h+s sits on
        // lane half h, and d mod 8 = 4(g&1)+2h+s selects the strided accumulator, so half 0
        // owns r0,r1,r4,r5 and half 1 owns r2,r3,r6,r7; the combine tree and the sequential
        // tail exchange values between the two halves with lane^32 shuffles.
        float sexact = __builtin_nanf("");
        if (fuse_exact) {
            const float *mrow = tiles + (int64_t)(idx >> 5) * tile_stride + (h * 32 + (idx & 31)) * 2;
            float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f, T0 = 0.f, T1 = 0.f, T2 = 0.f, T3 = 0.f;
#pragma unroll
            for (int g = 0; g < GMAX; g++) {
                const int i8 = g >> 1;
                float2 mv = *reinterpret_cast<const float2 *>(mrow + g * 128);
                float dx = mv.x - xb[nb][g].x, dy = mv.y - xb[nb][g].y;
                float qx = dx * dx, qy = dy * dy;
                if ((g & 1) == 0) {
                    if (i8 == 0) { A0 = qx; A1 = qy; }
                    else if (i8 < nb8) { A0 += qx; A1 += qy; }
                    if (i8 == nb8) { T0 = qx; T1 = qy; }
                } else {
                    if (i8 == 0) { A2 = qx; A3 = qy; }
                    else if (i8 < nb8) { A2 += qx; A3 += qy; }
                    if (i8 == nb8) { T2 = qx; T3 = qy; }
                }
            }
            float p = A0 + A1, q = A2 + A3;                      // (r0+r1),(r4+r5) | (r2+r3),(r6+r7)
            float po = __shfl_xor(p, 32), qo = __shfl_xor(q, 32);
            float res = (h == 0) ? ((p + po) + (q + qo)) : ((po + p) + (qo + q));
            float U0 = __shfl_xor(T0, 32), U1 = __shfl_xor(T1, 32), U2 = __shfl_xor(T2, 32),
                  U3 = __shfl_xor(T3, 32);
            // tail element jj (dim 8*nb8 + jj) lives on half (jj>>1)&1, slot (jj&1) + 2*(jj>>2)
            const float t0 = h == 0 ? T0 : U0, t1 = h == 0 ? T1 : U1, t2 = h == 0 ? U0 : T0,
                        t3 = h == 0 ? U1 : T1, t4 = h == 0 ? T2 : U2, t5 = h == 0 ? T3 : U3,
                        t6 = h == 0 ? U2 : T2;
            if (rem > 0) res += t0;
            if (rem > 1) res += t1;
            if (rem > 2) res += t2;
            if (rem > 3) res += t3;
            if (rem > 4) res += t4;
            if (rem > 5) res += t5;
            if (rem > 6) res += t6;
            sexact = -res;
        }
        if (h == 0 && rowid[nb] >= 0) {
            const int32_t id = rowid[nb];
            A.cand.k[id] = idx;
            A.cand.f[2 * (int64_t)id + 0] = top1;
            A.cand.f[2 * (int64_t)id + 1] = top2;
            A.cand.s[id] = (double)sexact;        // NaN when not fused
            // the filter cannot order the two best components with certainty: queue the row for
            // the full reference-arithmetic scan (k_kmeans_brute)
            const float M = (float)(sqrt(*A.mnorm2) * (1.0 + 1e-6)) + 1e-30f;
            const float tau = filter_tau(A.xnorm[id], M, D, A.is_f64);
            if (!(top1 - top2 > tau)) {
                int q = atomicAdd(A.cand.count, 1);
                if (q < A.amb_cap) A.cand.queue[q] = id;
            }
        }
    }
}

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
template <int P> struct SegkPiece;
template <> struct SegkPiece<3> {
    typedef __bf16 T;
    typedef __bf16 V8 __attribute__((ext_vector_type(8)));
};
template <> struct SegkPiece<2> {
    typedef _Float16 T;
    typedef _Float16 V8 __attribute__((ext_vector_type(8)));
};
template <int P>
__device__ __forceinline__ f32x16 mfma_piece(typename SegkPiece<P>::V8 a, typename SegkPiece<P>::V8 b, f32x16 c)
{
    if constexpr (P == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float filter_tau_sp(float xn, float M, int D, int pieces)
{
    const float u = 5.9604645e-8f;
    const int KP = (D + 15) & ~15;
    float e1 = (1.02f * (float)(KP + 16) + (pieces == 2 ? 16.f : 0.f)) * u * (xn * M + 0.5f * M * M);
    int levels = 0;
    for (int n = D; n > 128; n = (n + 1) / 2) levels++;
    int deff = D < 128 ? D : 128;
    float c2 = (float)(deff / 8 + 13 + 2 * levels);
    float s = xn + M;
    float e2 = c2 * u * s * s;
    return 1.25f * (2.0f * e1 + e2) + 1e-30f;
}

// pieces of one value (already scaled by its power of two for P = 2)
template <int P>
__device__ __forceinline__ void split_sp(float x, typename SegkPiece<P>::T *pc)
{
    typedef typename SegkPiece<P>::T T;
    if constexpr (P == 3) {
        const T a = (T)x;
        const float r1 = x - (float)a;
        const T b = (T)r1;
        const float r2 = r1 - (float)b;
        pc[0] = a;
        pc[1] = b;
        pc[2] = (T)r2;
    } else {
        const T a = (T)x;
        const float r1 = x - (float)a;                   // exact
        pc[0] = a;
        pc[1] = (T)(r1 * 2048.f);                        // 2^11 r1: exact scaling, then 11 of its <= 13 bits
    }
}

// exponent e such that 2^e * vmax lies in [2^12, 2^13); 0 for vmax = 0 / P = 3
__device__ __forceinline__ int sp_exponent(float vmax)
{
    if (!(vmax > 0.f)) return 0;
    int ex;
    frexpf(vmax, &ex);                                   // vmax = f * 2^ex, f in [0.5, 1)
    return 13 - ex;
}

// header of the row image: int32 {pieces, exponent a, bits of max |x_d|}
__global__ void k_corpus_maxabs(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned int *hdr)
{
    __shared__ float part[4];
    float v = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_emb * D; idx += (int64_t)gridDim.x * blockDim.x)
        v = fmaxf(v, fabsf(X[(idx / D) * ldx + (idx % D)]));
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (v > 0.f) atomicMax(hdr + 2, __float_as_uint(v));
    }
}

template <int P>
__global__ void k_corpus_split_sp(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned char *img)
{
    typedef typename SegkPiece<P>::T T;
    const int KP = segk_b3_kp(D);
    int *hdr = (int *)img;
    const int ea = P == 2 ? sp_exponent(__uint_as_float(((unsigned int *)img)[2])) : 0;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) { hdr[0] = P; hdr[1] = ea; }
    if (idx >= n_emb * KP) return;
    const int64_t e = idx / KP;
    const int pos = (int)(idx - e * KP), d = segk_b3_dim(pos);
    const float x = d < D ? ldexpf(X[e * ldx + d], ea) : 0.f;
    T pc[P];
    split_sp<P>(x, pc);
    T *row = (T *)(img + SEGK_SP_HEADER) + e * P * KP;
#pragma unroll
    for (int q = 0; q < P; q++) row[q * KP + pos] = pc[q];
}

// |x - x1| per row (x1 = the leading fp16 piece, unscaled): the operand-rounding term of the one-product
// pre-filter's margin is (|x| + e_x) E_m + e_x M by Cauchy-Schwarz on the actual residual vectors, about a
// third of the worst case 2^-10 |x| M.  An element whose piece is zero or subnormal in fp16 counts with its
// full magnitude, which covers a matrix pipe that flushes subnormal inputs as well as one that does not.
// Stored as float [n_emb] after the two piece planes (the image is sized for three).
__device__ __forceinline__ double sp_resid2(float scaled)
{
    const _Float16 a = (_Float16)scaled;
    const float af = (float)a;
    const float r = fabsf(af) < 6.103515625e-5f ? fabsf(scaled) : fabsf(scaled - af);    // 2^-14: smallest normal
    return (double)r * (double)r;
}
__global__ void k_corpus_resid_sp(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned char *img)
{
    const int ea = ((const int *)img)[1];
    const int KP = segk_b3_kp(D);
    float *xerr = (float *)(img + SEGK_SP_HEADER + n_emb * 2 * (int64_t)KP * 2);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_emb) return;
    double s = 0.0;
    for (int d = 0; d < D; d++) s += sp_resid2(ldexpf(X[e * ldx + d], ea));
    xerr[e] = (float)(ldexp(sqrt(s), -ea) * (1.0 + 1e-6)) + 1e-37f;
}

// tiles image: [header 1024 floats: int32 exponent b at [0]] then per tile [s][p][lane][8] pieces + 32 constants
// consts == NULL: the k-means constants -|m|^2/2; otherwise consts[k] (< -1e37: component absent) -- the
// log-sum-exp use of the kernel (segk_fbbatch.hip), whose rows are not means.
template <int P>
__global__ void k_kmeans_prepare_sp(const float *means, int K_max, int D, float *tiles, const double *mnorm2,
                                    const unsigned char *ximg, const double *consts)
{
    typedef typename SegkPiece<P>::T T;
    const int tile = blockIdx.x;
    const int KS = segk_b3_kp(D) / 16;
    const int stride = segk_sp_tile_stride(D, P);
    // max |m_d| <= sqrt(max |m|^2): every block derives the same exponent
    const int eb = P == 2 ? sp_exponent((float)(sqrt(*mnorm2) * (1.0 + 1e-6))) : 0;
    const int ea = ((const int *)ximg)[1];
    if (tile == 0 && threadIdx.x == 0) ((int *)tiles)[0] = eb;
    float *Tt = tiles + 1024 + (int64_t)tile * stride;
    T *Tb = (T *)Tt;
    __shared__ double nrm[32];
    {
        const int ci = threadIdx.x >> 3, sub = threadIdx.x & 7;      // 256 threads = 32 x 8
        const int comp = tile * 32 + ci;
        double s = 0.0, rs = 0.0;
        if (comp < K_max)
            for (int d = sub; d < D; d += 8) {
                const float mv = means[(int64_t)comp * D + d];
                double v = (double)mv;
                s += v * v;
                if (P == 2) rs += sp_resid2(ldexpf(mv, eb));
            }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (sub == 0) nrm[ci] = s;
        if (P == 2) {                     // E_m = max_k |m_k - m1_k|: tiles header [1], zeroed by k_kmeans_prepare just before
            rs += __shfl_xor(rs, 1);
            rs += __shfl_xor(rs, 2);
            rs += __shfl_xor(rs, 4);
            const float em = (float)(ldexp(sqrt(rs), -eb) * (1.0 + 1e-6));
            if (sub == 0 && comp < K_max) atomicMax((unsigned int *)tiles + 1, __float_as_uint(em));
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < KS * 64 * 8; idx += blockDim.x) {
        const int sidx = idx >> 9, lane = (idx >> 3) & 63, i = idx & 7;
        const int comp = tile * 32 + (lane & 31);
        const int d = segk_b3_dim(16 * sidx + 8 * (lane >> 5) + i);
        const float v = (comp < K_max && d < D) ? ldexpf(means[(int64_t)comp * D + d], eb) : 0.f;
        T pc[P];
        split_sp<P>(v, pc);
#pragma unroll
        for (int q = 0; q < P; q++) Tb[((sidx * P + q) * 64 + lane) * 8 + i] = pc[q];
    }
    for (int idx = threadIdx.x; idx < stride - KS * P * 256; idx += blockDim.x) {
        float v = 0.f;
        if (idx < 32) {
            const int comp = tile * 32 + idx;
            // the accumulators live in the scaled domain 2^(a+b) f
            if (consts) v = (comp < K_max && consts[comp] > -1e37) ? (float)ldexp(consts[comp], ea + eb) : -3.0e38f;
            else v = (comp < K_max) ? (float)ldexp(-0.5 * nrm[idx], ea + eb) : -3.0e38f;
        }
        Tt[KS * P * 256 + idx] = v;
    }
}

// The reference's float32 -(deltas*deltas).sum() of one (row, mean) pair in numpy's pairwise order, D a
// multiple of 4, by two lanes PART apart (h = 0, 1): lane h owns the strided accumulators r_{4h..4h+3} in
// full (segk_b3_dim); both return the same value.
template <int KS, int PART>
__device__ __forceinline__ float sp_exact_score_x(const float *mean, const float *xr, int D, int h)
{
    const float *mrow = mean + 4 * h, *xrow = xr + 4 * h;
    const int nfull = D & ~7, nblk = nfull >> 3;               // whole blocks of 8: both lanes, wave-uniform
    float r4[4] = {0.f, 0.f, 0.f, 0.f}, tt[4] = {0.f, 0.f, 0.f, 0.f};
    // the operands of block b + 1 are fetched before block b is accumulated (LDS or global latency under the
    // arithmetic); the accumulation order is untouched
    float4 mv = make_float4(0.f, 0.f, 0.f, 0.f), xv = mv;
    if (nblk > 0) {
        mv = *reinterpret_cast<const float4 *>(mrow);
        xv = *reinterpret_cast<const float4 *>(xrow);
    }
#pragma unroll
    for (int b = 0; b < 2 * KS; b++) {
        if (b < nblk) {
            float4 mn = mv, xn = xv;
            if (b + 1 < nblk) {
                mn = *reinterpret_cast<const float4 *>(mrow + 8 * (b + 1));
                xn = *reinterpret_cast<const float4 *>(xrow + 8 * (b + 1));
            }
            const float mvv[4] = {mv.x, mv.y, mv.z, mv.w}, xvv[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float delta = mvv[q] - xvv[q];
                const float t2 = delta * delta;
                r4[q] = b == 0 ? t2 : r4[q] + t2;
            }
            mv = mn;
            xv = xn;
        }
    }
    if (nfull + 4 * h < D) {                                   // the sequential tail block (D % 4 == 0: lane 0 only)
        const float4 mt = *reinterpret_cast<const float4 *>(mrow + nfull);
        const float4 xt = *reinterpret_cast<const float4 *>(xrow + nfull);
        const float mvv[4] = {mt.x, mt.y, mt.z, mt.w}, xvv[4] = {xt.x, xt.y, xt.z, xt.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float delta = mvv[q] - xvv[q];
            tt[q] = delta * delta;
        }
    }
    const int rem = D & 7;
    float res = (r4[0] + r4[1]) + (r4[2] + r4[3]);
    const float ro = __shfl_xor(res, PART);
    res = (h == 0) ? res + ro : ro + res;                      // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
    const float u0 = __shfl_xor(tt[0], PART), u1 = __shfl_xor(tt[1], PART), u2 = __shfl_xor(tt[2], PART);
    // tail dimension nfull + jj lives on half jj >> 2, slot jj & 3 (rem < 8, D % 4 == 0: rem is 0 or 4)
    const float t0 = h == 0 ? tt[0] : u0, t1 = h == 0 ? tt[1] : u1, t2 = h == 0 ? tt[2] : u2;
    const float t3 = h == 0 ? tt[3] : __shfl_xor(tt[3], PART);
    if (rem > 0) res += t0;
    if (rem > 1) res += t1;
    if (rem > 2) res += t2;
    if (rem > 3) res += t3;
    return -res;
}
template <int KS>
__device__ __forceinline__ float sp_exact_score(const float *mean, const float *xr, int D, int h)
{
    return sp_exact_score_x<KS, 32>(mean, xr, D, h);      // the two 32-lane halves of a wave
}

template <int KS, int WAVES, int P, int MODE = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void k_kmeans_score_sp(ScoreArgs A)
{
    typedef typename SegkPiece<P>::T T;
    typedef typename SegkPiece<P>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int32_t *__restrict__ ids = A.ids;
    const int64_t row0 = A.row0;
    int64_t n = A.n;
    if (A.n_dev) {                        // rows queued by the pre-filter: the count lives on the device
        const int64_t nd = *A.n_dev;
        n = nd < n ? nd : n;
        if ((int64_t)blockIdx.x * WAVES * 32 >= n) return;
    }
    const float *__restrict__ tiles = A.tiles + 1024;
    const int n_tiles = A.n_tiles, D = A.D;
    constexpr int KP = KS * 16;
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;      // floats per tile image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    // scaled domain: accumulators hold 2^(a+b) f (P = 2), unscaled again before anything leaves the kernel
    const int e_ab = ((const int *)A.X32)[1] + ((const int *)A.tiles)[0];
    const float unscale = ldexpf(1.f, -e_ab);
    constexpr float LS = P == 2 ? 1.f / 2048.f : 1.f;

    V8 xb[P][KS];
    const int64_t r = ((int64_t)blockIdx.x * WAVES + wave) * 32 + j;
    int32_t rowid = -1;
    if (r < n) rowid = ids ? ids[r] : (int32_t)(row0 + r);
    {
        const T *xp = (const T *)((const unsigned char *)A.X32 + SEGK_SP_HEADER) + (int64_t)(rowid >= 0 ? rowid : 0) * (P * KP) + 8 * h;
#pragma unroll
        for (int p = 0; p < P; p++)
#pragma unroll
            for (int s = 0; s < KS; s++) xb[p][s] = *reinterpret_cast<const V8 *>(xp + p * KP + 16 * s);
    }
    // MODE 1 (log-sum-exp, base 2, of the UNSCALED accumulator values): m1 / m2 are the running maximum
    // (finite start) and sum, as in k_kmeans_score
    float m1 = MODE == 1 ? -3.0e38f : NEG_INF_F, m2 = MODE == 1 ? 0.f : NEG_INF_F;
    int32_t irow = 0, itile = 0;

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
                *reinterpret_cast<float4 *>(A.mat_out + r * A.mat_ld + dtile * 32 + 4 * h + 8 * ((vi) >> 2)) = \
                    make_float4(st4[0], st4[1], st4[2], st4[3]);                      \
        } else if constexpr (MODE == 1) {                                                    \
            v_ = fmaxf(v_ * unscale, -3.0e38f);                                       \
            const float nm_ = vmax_f32(m1, v_);                                       \
            m2 = m2 * __builtin_amdgcn_exp2f(m1 - nm_) + __builtin_amdgcn_exp2f(v_ - nm_); \
            m1 = nm_;                                                                 \
        } else                                                                        \
        asm volatile("v_cmp_ngt_f32 vcc, %3, %0\n\t"                                  \
                     "v_med3_f32 %1, %0, %1, %3\n\t"                                  \
                     "v_max_f32 %0, %0, %3\n\t"                                       \
                     "v_cndmask_b32 %2, %4, %2, vcc"                                  \
                     : "+v"(m1), "+v"(m2), "+v"(irow)                                 \
                     : "v"(v_), "n"((vi))                                             \
                     : "vcc");                                                        \
    } while (0)

#define SEGK_TILE(NEWM, NEWL, OLDM, OLDL, t_)                                                         \
    do {                                                                                              \
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
        SEGK_TILE_SYNC();                                                                             \
    } while (0)

    int t = 0;
    for (; t + 1 < n_tiles; t += 2) {
        SEGK_TILE(accAm, accAl, accBm, accBl, t);
        SEGK_TILE(accBm, accBl, accAm, accAl, t + 1);
    }
    {
        float m1s = m1;
        if (t < n_tiles) {
            SEGK_TILE(accAm, accAl, accBm, accBl, t);
            m1s = m1;
            dtile = n_tiles - 1;
#pragma unroll
            for (int vi = 0; vi < 16; vi++) SEGK_DRAIN(accAm, accAl, vi);
        } else {
            dtile = n_tiles - 1;
#pragma unroll
            for (int vi = 0; vi < 16; vi++) SEGK_DRAIN(accBm, accBl, vi);
        }
        itile = (m1 > m1s) ? (n_tiles - 1) : itile;
    }
#undef SEGK_TILE
#undef SEGK_DRAIN
#undef SEGK_STAGE
#undef SEGK_TILE_SYNC
    if constexpr (MODE == 2) return;
    if constexpr (MODE == 1) {
        // the two lane halves summed disjoint component subsets of the same row
        const float om = __shfl_xor(m1, 32), os = __shfl_xor(m2, 32);
        const float M = fmaxf(m1, om);
        const float S = m2 * exp2f(m1 - M) + os * exp2f(om - M);
        if (h == 0 && rowid >= 0) A.lse_out[rowid] = (double)(M + log2f(S)) * 0.6931471805599453 - A.lse_norm;
        return;
    }
    const int32_t i1 = itile * 32 + 4 * h + (irow & 3) + 8 * (irow >> 2);
    const float o1 = __shfl_xor(m1, 32), o2 = __shfl_xor(m2, 32);
    const int oi = __shfl_xor(i1, 32);
    const float top1 = fmaxf(m1, o1) * unscale;                    // powers of two: exact
    const float top2 = fmaxf(fminf(m1, o1), fmaxf(m2, o2)) * unscale;
    const int idx = (o1 > m1 || (o1 == m1 && oi < i1)) ? oi : i1;
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
}

// ======================================================================================
// A1 pre-filter: ONE fp16 product.
// The split-precision score kernel is power-bound (the same instruction stream on all-zero rows runs
// 24 % faster, profiles/README.md r01_h): what shortens it is fewer matrix operations, not a better
// schedule.  Most rows are decided by far less precision than fp16x2 carries: with only the leading
// pieces, sum_d x1_d m1_d, both operands are rounded once to fp16 (unit roundoff 2^-11), so
//     |sum x1 m1 - sum x m| <= (2^-10 + 2^-21) sum |x_d||m_d| + (flushed elements)
//                           <= 1.01 * 2^-10 |x| M                               (Cauchy-Schwarz)
// (elements below the fp16 normal range after the power-of-two scaling, max element in [2^12, 2^13),
// are off by at most 2^-25 in the scaled domain: < 2^-33 |x| M for D <= 128, inside the 1.01).  A row
// whose two largest values differ by more than tau_A = tau' + 2.5 * 1.01 * 2^-10 |x| M (tau' the
// split-precision margin, which covers the fp32 accumulation and the exact stage's own rounding) has
// the reference's argmax as its winner; on the bench corpus that is 94 % of the rows.  The others are
// queued for k_kmeans_score_sp (all three products), whose own undecided rows take the full scan.
//
// One third of the matrix work makes the top-2 update the cost that matters, so it is done on PAIRS
// of values: m1' = max3(m1, a, b), m2' = max(m2, med3(m1, a, b)), and the index kept is the pair's --
// five vector operations per two values instead of eight.  Which of the pair won is settled by the
// exact stage, which scores both members in reference arithmetic (a decisive winner beats its
// partner there as well).  A wave owns NBLK blocks of 32 rows (tile fragments stay in registers
// across the blocks; staging, barriers and fragment reads amortise over NBLK x 7 MFMAs), two
// accumulators: block b's MFMAs run over the drain of block b - 1.
// Reads the two-piece images (segk_internal.h): piece 0 of the rows, the piece-0 blocks and the
// constants of the tile image -- each a 1 KiB LDS-DMA piece.
// ======================================================================================
__device__ __forceinline__ float filter_tau_h1(float xn, float M, int D, float ex, float Em)
{
    // operand rounding: |sum x1 m1 - sum x m| <= |x1| |m1 - m| + |x1 - x| |m| <= (|x| + e_x) E_m + e_x M with the
    // residual norms of THIS row and the worst component (k_corpus_resid_sp / k_kmeans_prepare_sp), never more
    // than the a-priori 1.01 * 2^-10 |x| M
    const float meas = (xn + ex) * Em + ex * M;
    const float apriori = 1.01f * 9.765625e-4f * xn * M;
    return filter_tau_sp(xn, M, D, 2) + 2.5f * 1.00001f * fminf(meas, apriori);
}

template <int KS, int NBLK>
__global__ __launch_bounds__(256, 2) void k_kmeans_score_h1(ScoreArgs A)
{
    static_assert(NBLK == 2 || NBLK == 4, "an even number of row blocks per wave (static accumulator parity)");
    typedef _Float16 T;
    typedef SegkPiece<2>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int32_t *__restrict__ ids = A.ids;
    const int64_t row0 = A.row0, n = A.n;
    const float *__restrict__ tiles = A.tiles + 1024;
    const int n_tiles = A.n_tiles, D = A.D;
    constexpr int P = 2, KP = KS * 16;
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;      // floats per tile image (global)
    constexpr int TS = (KS + 1) * 256;                                    // floats per LDS buffer: KS blocks + constants
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int e_ab = ((const int *)A.X32)[1] + ((const int *)A.tiles)[0];
    const float unscale = ldexpf(1.f, -e_ab);

#ifdef SEGK_STAMP
#define SEGK_STAMP_AT(i) do { if (A.stamp && tid == 0) A.stamp[(int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEGK_STAMP_AT(i) do { } while (0)
#endif
    SEGK_STAMP_AT(0);
    V8 xb[NBLK][KS];
    int64_t r[NBLK];
    int32_t rowid[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
        r[b] = ((int64_t)blockIdx.x * 4 + wave) * (32 * NBLK) + 32 * b + j;
        rowid[b] = -1;
        if (r[b] < n) rowid[b] = ids ? ids[r[b]] : (int32_t)(row0 + r[b]);
        const T *xp = (const T *)((const unsigned char *)A.X32 + SEGK_SP_HEADER) + (int64_t)(rowid[b] >= 0 ? rowid[b] : 0) * (P * KP) + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; s++) xb[b][s] = *reinterpret_cast<const V8 *>(xp + 16 * s);
    }
    float m1[NBLK], m2[NBLK];
    int32_t ipr[NBLK], itile[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; b++) { m1[b] = NEG_INF_F; m2[b] = NEG_INF_F; ipr[b] = 0; itile[b] = 0; }

    typedef __attribute__((address_space(3))) void *lptr_t;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)lds);
    constexpr int NPASS = (KS + 1 + 3) / 4;
    // piece q < KS: the piece-0 block of k-step q; piece KS: the constants.  Wave q % 4 copies it.
    // (LDS-DMA from inline asm, one explicit wait per tile: see k_kmeans_score_sp)
#define SEGK_STAGE(tt, buf)                                                                         \
    do {                                                                                            \
        _Pragma("unroll") for (int p = 0; p < NPASS; p++) {                                         \
            const int q_ = p * 4 + wave;                                                            \
            if (q_ <= KS) {                                                                         \
                const float *src_ = tiles + (int64_t)(tt) * STRIDE + (q_ < KS ? q_ * P * 256 : KS * P * 256) + lane * 4; \
                const unsigned dst_ = __builtin_amdgcn_readfirstlane(lds_base + ((buf) * TS + q_ * 256) * 4); \
                unsigned keep_;                                                                     \
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"                 \
                             "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"                  \
                             : "=&s"(keep_)                                                         \
                             : "v"(src_), "s"(dst_)                                                 \
                             : "memory");                                                           \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define SEGK_TILE_SYNC()                                                      \
    do {                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();                                         \
    } while (0)

    SEGK_STAGE(0, 0);
    SEGK_TILE_SYNC();
    SEGK_STAMP_AT(1);

    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 16; q++) { acc[0][q] = NEG_INF_F; acc[1][q] = NEG_INF_F; }

    // values 2 pi, 2 pi + 1 of block O_'s accumulator: components c, c + 1 of this lane half
#define SEGK_DRAIN2(O_, ACC, pi)                                                      \
    do {                                                                              \
        /* the first read of the MFMA results is a compiler-visible instruction: the hazard recogniser */ \
        /* does not look inside inline asm, and these values can be a few cycles old (block b - 1)     */ \
        const float tmp_ = __builtin_amdgcn_fmed3f(m1[O_], ACC[2 * (pi)], ACC[2 * (pi) + 1]);          \
        float nm_;                                                                    \
        asm volatile("v_max_f32 %1, %1, %3\n\t"                                       \
                     "v_max3_f32 %0, %4, %5, %6\n\t"                                  \
                     "v_cmp_nlt_f32 vcc, %4, %0\n\t"                                  \
                     "v_cndmask_b32 %2, %7, %2, vcc"                                  \
                     : "=&v"(nm_), "+v"(m2[O_]), "+v"(ipr[O_])                        \
                     : "v"(tmp_), "v"(m1[O_]), "v"(ACC[2 * (pi)]), "v"(ACC[2 * (pi) + 1]), "n"((pi)) \
                     : "vcc");                                                        \
        m1[O_] = nm_;                                                                 \
    } while (0)

    constexpr int PPS = (8 + KS - 1) / KS;
    // MFMAs of block N_ on the current tile over the drain of block O_'s values of tile dt_
#define SEGK_UNIT(N_, O_, dt_)                                                                        \
    do {                                                                                              \
        {                                                                                             \
            const float *cv = Tt + KS * 256 + 4 * h;                                                  \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                           \
                float4 c4 = *reinterpret_cast<const float4 *>(cv + 8 * q);                            \
                acc[(N_) & 1][4 * q + 0] = c4.x; acc[(N_) & 1][4 * q + 1] = c4.y;                     \
                acc[(N_) & 1][4 * q + 2] = c4.z; acc[(N_) & 1][4 * q + 3] = c4.w;                     \
            }                                                                                         \
        }                                                                                             \
        const float m1s = m1[O_];                                                                     \
        _Pragma("unroll") for (int s = 0; s < KS; s++) {                                              \
            acc[(N_) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], xb[N_][s], acc[(N_) & 1], 0, 0, 0); \
            _Pragma("unroll") for (int q = 0; q < PPS; q++)                                           \
                if (s * PPS + q < 8) SEGK_DRAIN2(O_, acc[((N_) & 1) ^ 1], s * PPS + q);               \
        }                                                                                             \
        itile[O_] = (m1[O_] > m1s) ? (dt_) : itile[O_];                                               \
    } while (0)

    for (int t = 0; t < n_tiles; t++) {
        if (t + 1 < n_tiles) SEGK_STAGE(t + 1, (t + 1) & 1);
        const float *Tt = lds + (t & 1) * TS;
        const T *Tb = (const T *)Tt;
        V8 a[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) a[s] = *reinterpret_cast<const V8 *>(Tb + (s * 64 + lane) * 8);
        SEGK_UNIT(0, NBLK - 1, t - 1);
        SEGK_UNIT(1, 0, t);
        if constexpr (NBLK == 4) {
            SEGK_UNIT(2, 1, t);
            SEGK_UNIT(3, 2, t);
        }
        if (t == 15) SEGK_STAMP_AT(4);
        SEGK_TILE_SYNC();
        if (t == 15) SEGK_STAMP_AT(5);
    }
    SEGK_STAMP_AT(2);
    {
        const float m1s = m1[NBLK - 1];
#pragma unroll
        for (int pi = 0; pi < 8; pi++) SEGK_DRAIN2(NBLK - 1, acc[(NBLK - 1) & 1], pi);
        itile[NBLK - 1] = (m1[NBLK - 1] > m1s) ? (n_tiles - 1) : itile[NBLK - 1];
    }
#undef SEGK_UNIT
#undef SEGK_DRAIN2
#undef SEGK_STAGE
#undef SEGK_TILE_SYNC

    const float M = (float)(sqrt(*A.mnorm2) * (1.0 + 1e-6)) + 1e-30f;
    const float Em = ((const float *)A.tiles)[1];
    bool undecided[NBLK];
    int n_und = 0;
#pragma unroll
    for (int b = 0; b < NBLK; b++) {
        // the winning pair: components c, c + 1 (accumulator elements 2 ipr, 2 ipr + 1 of tile itile)
        const int32_t c0 = itile[b] * 32 + 4 * h + 2 * (ipr[b] & 1) + 8 * (ipr[b] >> 1);
        const float o1 = __shfl_xor(m1[b], 32), o2 = __shfl_xor(m2[b], 32);
        const int oc = __shfl_xor(c0, 32);
        const float top1 = fmaxf(m1[b], o1) * unscale;                 // powers of two: exact
        const float top2 = fmaxf(fminf(m1[b], o1), fmaxf(m2[b], o2)) * unscale;
        // equal maxima on the two halves leave a zero margin: the row is queued whichever pair is named
        const int cw = (o1 > m1[b]) ? oc : c0;
        undecided[b] = false;
        if (h == 0 && rowid[b] >= 0) {
            const int32_t rid = rowid[b];
            const float tau = filter_tau_h1(A.xnorm[rid], M, D, A.xerr[rid], Em);
            if (top1 - top2 > tau) {
                // decided up to the member of the pair: k_kmeans_exact_pair scores both in reference
                // arithmetic (a decisive winner beats its partner there as well) and clears the mark
                A.cand.k[rid] = cw | SEGK_PAIR_PENDING;
                A.cand.f[2 * (int64_t)rid + 0] = top1;
                A.cand.f[2 * (int64_t)rid + 1] = top2;
            } else {
                undecided[b] = true;
            }
        }
        n_und += __popcll(__ballot(undecided[b]));
    }
    // ONE queue reservation per wave (a returning atomic is a round trip to L2; one per row block kept the
    // wave waiting four times over)
    if (n_und > 0) {                                                   // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(A.pre_count, n_und);
        base = __shfl(base, 0);
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            const unsigned long long mask = __ballot(undecided[b]);
            if (undecided[b]) {
                const int q = base + __popcll(mask & ((1ull << lane) - 1ull));
                if (q < A.pre_cap) A.pre_queue[q] = rowid[b];
                else {                                                 // beyond the second stage's launch: full scan
                    const int q2 = atomicAdd(A.cand.count, 1);
                    if (q2 < A.amb_cap) A.cand.queue[q2] = rowid[b];
                }
            }
            base += __popcll(mask);
        }
    }
    SEGK_STAMP_AT(3);
#undef SEGK_STAMP_AT
}

// A handful of left-over rows (fewer than SEGK_TAIL_QUEUE): not worth three more launches -- they
// are appended to the ambiguity queue and take the full reference-arithmetic scan.
#define SEGK_TAIL_QUEUE 2048
__global__ void k_score_queue_rows(ScoreArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n) return;
    const int32_t id = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
    if (id < 0) return;
    const int q = atomicAdd(A.cand.count, 1);
    if (q < A.amb_cap) A.cand.queue[q] = id;
}

// Combine the partial candidates of a split-K launch: per row the largest filter value (ties: the
// lower component), the second largest over everything else, then the same margin test as the
// unsplit epilogue.  The winner's exact score is left to k_kmeans_exact_fill (cand.s = NaN).
__global__ void k_score_merge(ScoreArgs A, int n_split)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n) return;
    const int32_t id = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
    if (id < 0) return;
    float top1 = NEG_INF_F, top2 = NEG_INF_F;
    int idx = 0x7fffffff;
    for (int sp = 0; sp < n_split; sp++) {
        const int64_t e = (int64_t)sp * A.n + r;
        const float f1 = A.part_f[2 * e], f2 = A.part_f[2 * e + 1];
        const int k = A.part_k[e];
        if (f1 > top1 || (f1 == top1 && k < idx)) {
            top2 = fmaxf(top2, top1);
            top1 = f1;
            idx = k;
        } else {
            top2 = fmaxf(top2, f1);
        }
        top2 = fmaxf(top2, f2);
    }
    A.cand.k[id] = idx;
    A.cand.f[2 * (int64_t)id + 0] = top1;
    A.cand.f[2 * (int64_t)id + 1] = top2;
    A.cand.s[id] = (double)__builtin_nanf("");
    const float M = (float)(sqrt(*A.mnorm2) * (1.0 + 1e-6)) + 1e-30f;
    const float tau = filter_tau(A.xnorm[id], M, A.D, A.is_f64);
    if (!(top1 - top2 > tau)) {
        int q = atomicAdd(A.cand.count, 1);
        if (q < A.amb_cap) A.cand.queue[q] = id;
    }
}

// ======================================================================================
// Exact stage.
//   k_kmeans_brute       every queued (ambiguous) row: the reference's own computation for ALL
//                        K_max components, first maximum (np.argmax); one workgroup per row,
//                        components contiguous across lanes (tile image) for float32 data
//   k_kmeans_exact_fill  rows whose winner was not evaluated in the score kernel's epilogue
//                        (float64 data, D < 8 or D > 128): exact score of the winner
// After these, cand.k / cand.s hold np.argmax / np.max of neg_sqrd_norm for every scored row.
// ======================================================================================
template <typename XT>
__global__ void k_kmeans_brute(segk_corpus c, segk_kmeans m, segk_cand cand, int cap, int32_t *n_brute)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    double *red_v = (double *)smem;                  // [nt]
    XT *xrow = (XT *)(red_v + nt);                   // [D]
    int32_t *red_k = (int32_t *)(xrow + ((c.D + 1) & ~1));   // [nt]
    const XT *X = (const XT *)c.X;
    const XT *means = (const XT *)m.means;
    const int D = c.D;
    int nq = *cand.count;
    if (nq > cap) nq = cap;
    if (blockIdx.x == 0 && tid == 0 && n_brute && nq > 0) atomicAdd(n_brute, nq);
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        const int32_t id = cand.queue[q];
        __syncthreads();
        for (int d = tid; d < D; d += nt) xrow[d] = X[(int64_t)id * c.ldx + d];
        __syncthreads();
        XT best = (XT)NEG_INF_D;
        int32_t bk = 0x7fffffff;
        if constexpr (sizeof(XT) == 4) {
            const int tstride = segk_tile_stride(D);
            int k = tid;
            if (D >= 8 && D <= 128) {
                // four components per thread in flight: the same numpy-ordered accumulation for
                // each, but their loads are independent, which hides the L2 latency
                for (; k + 3 * nt < m.K_max; k += 4 * nt) {
                    XT sc[4];
                    neg_sqd_exact_x4<XT>(tile_row(m.tiles, tstride, k), tile_row(m.tiles, tstride, k + nt),
                                         tile_row(m.tiles, tstride, k + 2 * nt), tile_row(m.tiles, tstride, k + 3 * nt),
                                         xrow, D, sc);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (sc[q] > best || bk == 0x7fffffff) { best = sc[q]; bk = k + q * nt; }
                }
            }
            for (; k < m.K_max; k += nt) {
                XT sc = neg_sqd_exact<XT>(tile_row(m.tiles, tstride, k), xrow, D);
                if (sc > best || bk == 0x7fffffff) { best = sc; bk = k; }   // first max within the thread
            }
        } else {
            for (int k = tid; k < m.K_max; k += nt) {
                XT sc = neg_sqd_exact<XT>(means + (int64_t)k * D, xrow, D);
                if (sc > best || bk == 0x7fffffff) { best = sc; bk = k; }
            }
        }
        red_v[tid] = (double)best;
        red_k[tid] = bk;
        __syncthreads();
        for (int o = nt >> 1; o > 0; o >>= 1) {
            if (tid < o) {
                double v2 = red_v[tid + o];
                int32_t k2 = red_k[tid + o];
                bool take = (k2 != 0x7fffffff) &&
                            (red_k[tid] == 0x7fffffff || v2 > red_v[tid] || (v2 == red_v[tid] && k2 < red_k[tid]));
                if (take) { red_v[tid] = v2; red_k[tid] = k2; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            cand.k[id] = red_k[0];
            cand.s[id] = red_v[0];
        }
    }
}

// Full scan of BR queued rows per workgroup for float32 data with 8 <= D <= 128 (numpy's
// single-block case): a thread walks the components tid, tid + nt, ... and evaluates each against
// the BR rows held in LDS -- every component value is fetched once for BR rows, which takes the scan
// from L2-bandwidth bound (one pass over the tile image per row) to latency/compute bound.  Per
// (row, component) the arithmetic is neg_sqd_exact's: eight strided accumulators, the fixed combine
// tree, the sequential tail; first maximum per row.
// BR = 4 (was 8: 234 VGPRs): in the pre-filter path the scan runs on the second stream beside the exact pair
// kernel, whose waves hold 144 VGPRs each -- with 8 rows its workgroups could not be placed until those
// waves ended (121 us in the trace against 65 alone); with 4 the sweep gains 4 %.
#define SEGK_BR 4
// component slices of the full scan for a queue of nq rows on a grid of `grid` workgroups (at most max_split)
__device__ __forceinline__ int segk_brute_split(int nq, int grid, int max_split)
{
    const int groups = (nq + SEGK_BR - 1) / SEGK_BR;
    int ks = groups > 0 ? grid / groups : 1;
    if (ks > max_split) ks = max_split;
    return ks < 1 ? 1 : ks;
}

__global__ __launch_bounds__(256) void k_kmeans_brute_rows(segk_corpus c, segk_kmeans m, segk_cand cand, int cap, int32_t *n_brute,
                                                          int n_groups, int ksplit, unsigned long long *ws, int ws_cap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const int DP = (D + 3) & ~3;                             // row pitch: float4 reads of the staged rows
    float *xs = (float *)smem;                               // [BR][DP]
    float *red_v = xs + SEGK_BR * DP;                        // [nt]
    int32_t *red_k = (int32_t *)(red_v + nt);                // [nt]
    __shared__ int32_t ids[SEGK_BR];
    const float *X = (const float *)c.X;
    int nq = *cand.count;
    if (nq > cap) nq = cap;
    if (blockIdx.x == 0 && tid == 0 && n_brute && nq > 0) atomicAdd(n_brute, nq);
    const int tstride = segk_tile_stride(D);
    const int nfull = D - (D % 8);
    // workgroup = (row group, component slice): the slices of a row meet in ws[] through a 64-bit
    // atomicMax on (orderable score bits, ~component) -- the largest score, the lowest component on ties;
    // k_brute_finish unpacks.  Queue entries beyond ws_cap keep the unsplit form (slice 0 scans all).
    // The host does not know the queue length (it lives on the device), so the split is chosen here, from
    // the launched grid: as many component slices (up to `ksplit`, one component per thread and slice) as
    // the grid has workgroups per row group.  segk_brute_split() is shared with k_brute_finish.
    ksplit = segk_brute_split(nq, (int)gridDim.x, ksplit);
    n_groups = (int)gridDim.x / ksplit;
    const int grp0 = blockIdx.x % n_groups, slice = blockIdx.x / n_groups;
    if (slice >= ksplit) return;
    const int k_per = (m.K_max + ksplit - 1) / ksplit;
    for (int q0 = grp0 * SEGK_BR; q0 < nq; q0 += n_groups * SEGK_BR) {
        const bool split = ksplit > 1 && q0 + SEGK_BR <= ws_cap;
        if (!split && slice != 0) continue;
        const int k_lo = split ? slice * k_per : 0;
        const int k_hi = split ? (k_lo + k_per < m.K_max ? k_lo + k_per : m.K_max) : m.K_max;
        const int nr = nq - q0 < SEGK_BR ? nq - q0 : SEGK_BR;
        __syncthreads();
        if (tid < SEGK_BR) ids[tid] = cand.queue[q0 + (tid < nr ? tid : nr - 1)];
        __syncthreads();
        for (int j = tid; j < SEGK_BR * D; j += nt) {
            const int r = j / D, d = j - r * D;
            xs[r * DP + d] = X[(int64_t)ids[r] * c.ldx + d];
        }
        __syncthreads();
        float best[SEGK_BR];
        int32_t bk[SEGK_BR];
#pragma unroll
        for (int r = 0; r < SEGK_BR; r++) { best[r] = NEG_INF_F; bk[r] = 0x7fffffff; }
        for (int k = k_lo + tid; k < k_hi; k += nt) {
            const TileRow mr = tile_row(m.tiles, tstride, k);
            float acc[SEGK_BR][8];
            float mv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) mv[j] = mr[j];
#pragma unroll
            for (int r = 0; r < SEGK_BR; r++) {
                const float4 x0 = *reinterpret_cast<const float4 *>(xs + r * DP), x1 = *reinterpret_cast<const float4 *>(xs + r * DP + 4);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float delta = mv[j] - xv[j];
                    acc[r][j] = delta * delta;
                }
            }
            int i;
            for (i = 8; i < nfull; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; j++) mv[j] = mr[i + j];
#pragma unroll
                for (int r = 0; r < SEGK_BR; r++) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xs + r * DP + i), x1 = *reinterpret_cast<const float4 *>(xs + r * DP + i + 4);
                    const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float delta = mv[j] - xv[j];
                        acc[r][j] += delta * delta;
                    }
                }
            }
            float res[SEGK_BR];
#pragma unroll
            for (int r = 0; r < SEGK_BR; r++)
                res[r] = ((acc[r][0] + acc[r][1]) + (acc[r][2] + acc[r][3])) + ((acc[r][4] + acc[r][5]) + (acc[r][6] + acc[r][7]));
            for (; i < D; i++) {
                const float mvi = mr[i];
#pragma unroll
                for (int r = 0; r < SEGK_BR; r++) {
                    const float delta = mvi - xs[r * DP + i];
                    res[r] += delta * delta;
                }
            }
#pragma unroll
            for (int r = 0; r < SEGK_BR; r++) {
                const float sc = -res[r];
                if (sc > best[r] || bk[r] == 0x7fffffff) { best[r] = sc; bk[r] = k; }   // first max within the thread
            }
        }
        for (int r = 0; r < nr; r++) {
            // wave butterfly (ties -> lower component), then the waves' results through LDS
            float v = best[0];
            int32_t kk = bk[0];
#pragma unroll
            for (int q = 1; q < SEGK_BR; q++)
                if (q == r) { v = best[q]; kk = bk[q]; }
            for (int o = 32; o > 0; o >>= 1) {
                const float v2 = __shfl_xor(v, o);
                const int32_t k2 = __shfl_xor(kk, o);
                const bool take = (k2 != 0x7fffffff) && (kk == 0x7fffffff || v2 > v || (v2 == v && k2 < kk));
                if (take) { v = v2; kk = k2; }
            }
            __syncthreads();
            if ((tid & 63) == 0) { red_v[tid >> 6] = v; red_k[tid >> 6] = kk; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < (nt >> 6); w++) {
                    const float v2 = red_v[w];
                    const int32_t k2 = red_k[w];
                    const bool take = (k2 != 0x7fffffff) && (kk == 0x7fffffff || v2 > v || (v2 == v && k2 < kk));
                    if (take) { v = v2; kk = k2; }
                }
                if (split) {
                    if (kk != 0x7fffffff) {
                        const unsigned int bits = __float_as_uint(v);
                        const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                        atomicMax(&ws[q0 + r], ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)kk));
                    }
                } else {
                    cand.k[ids[r]] = kk;
                    cand.s[ids[r]] = (double)v;
                }
            }
        }
    }
}

// unpack the split scan's (score, component) pairs into the candidates and clear the workspace
__global__ void k_brute_finish(segk_cand cand, int cap, unsigned long long *ws, int ws_cap, int scan_grid, int max_split)
{
    int nq = *cand.count;
    if (nq > cap) nq = cap;
    if (segk_brute_split(nq, scan_grid, max_split) <= 1) return;       // the scan wrote the candidates itself
    // the groups that were scanned in slices: q0 + SEGK_BR <= ws_cap
    const int lim = nq < (ws_cap / SEGK_BR) * SEGK_BR ? nq : (ws_cap / SEGK_BR) * SEGK_BR;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < lim; q += gridDim.x * blockDim.x) {
        const unsigned long long pk = ws[q];
        ws[q] = 0ull;
        const unsigned int ord = (unsigned int)(pk >> 32);
        const unsigned int bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
        const int32_t id = cand.queue[q];
        cand.k[id] = (int32_t)(0xffffffffu - (unsigned int)(pk & 0xffffffffu));
        cand.s[id] = (double)__uint_as_float(bits);
    }
}

template <typename XT>
__global__ void k_kmeans_exact_fill(segk_corpus c, segk_kmeans m, const int32_t *ids, int64_t row0, int64_t n,
                                    segk_cand cand)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t id = ids ? (int64_t)ids[r] : row0 + r;
    if (id < 0) return;
    const double sv = cand.s[id];
    if (sv == sv) return;
    cand.s[id] = (double)neg_sqd_exact<XT>((const XT *)m.means + (int64_t)cand.k[id] * c.D,
                                           (const XT *)c.X + id * c.ldx, c.D);
}

__global__ void k_kmeans_gather_cand(segk_cand cand, const int32_t *ids, int64_t n, double *out_max,
                                     int32_t *out_arg)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t id = ids ? (int64_t)ids[r] : r;
    out_max[r] = cand.s[id];
    out_arg[r] = cand.k[id];
}

// A1 full vector for one row (API: segk_kmeans_neg_sqrd_norm)
template <typename XT>
__global__ void k_kmeans_neg_sqrd_norm(segk_corpus c, segk_kmeans m, int64_t row, XT *out)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m.K_max) return;
    out[k] = neg_sqd_exact<XT>((const XT *)m.means + (int64_t)k * c.D, (const XT *)c.X + row * c.ldx, c.D);
}

// ======================================================================================
// Per-utterance kernel: A5 (vec from the candidates), A8 (max-plus DP), tokens.
//   ONE WAVE per utterance, no workgroup barriers: lanes gather the band of candidate spans,
//   lane 0 runs the DP on LDS, lanes write the results.
//   band layout: entry (t, w), t = 1..N (span end), w = 0..W-1 (span length w+1, start
//   s = t-1-w) at [(t-1)*W + w]; W = n_slices_max, or N when n_slices_max == 0.
// ======================================================================================
#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

__global__ void k_kmeans_segment(segk_corpus c, segk_kmeans m, const int32_t *utts, int utt0, int n_utts,
                                 int n_min, int n_max, double wip, segk_cand cand, uint8_t *boundaries,
                                 int32_t *old_tok, int32_t *new_tok, int32_t *new_k, int32_t *n_old,
                                 int32_t *n_new, int32_t *n_flag, double *out_total, int32_t *status, int band_cap,
                                 int wave_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    (void)n_min;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * (blockDim.x >> 6) + wv;
    if (slot >= n_utts) return;
    const int u = utts ? utts[slot] : utt0 + slot;
    const int N = c.lengths[u];
    const int W = (n_max > 0 && n_max < N) ? n_max : N;
    const int nb = N * W;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
    const double *dur = c.durations + (int64_t)u * triMax;
    uint8_t *gbnd = boundaries + (int64_t)u * c.N_max;

    char *base = smem + (size_t)wv * wave_bytes;
    double *bvec = (double *)base;                    // [band_cap]
    double *gam = bvec + band_cap;                    // [N_max + 1]
    int32_t *bk = (int32_t *)(gam + c.N_max + 1);     // [band_cap]
    int32_t *bid = bk + band_cap;                     // [band_cap]
    int32_t *l_old = bid + band_cap;                  // [N_max]
    int32_t *l_new = l_old + c.N_max;                 // [N_max]
    int32_t *l_newk = l_new + c.N_max;                // [N_max]
    int32_t *l_cnt = l_newk + c.N_max;                // [2]
    uint8_t *l_bnd = (uint8_t *)(l_cnt + 2);          // [N_max]

    for (int i = lane; i < nb; i += 64) {
        const int t = i / W + 1, w = i % W, s = t - 1 - w;
        int id = -1;
        double v = NEG_INF_D;
        int k = -1;
        if (s >= 0) {
            const int j = t * (t - 1) / 2 + s;
            id = vid[j];
            if (id >= 0) {
                k = cand.k[id];
                const double dd = dur[j];
                v = isnan(dd) ? NEG_INF_D : cand.s[id] * dd;      // :346-349
            }
        }
        bid[i] = id;
        bk[i] = k;
        bvec[i] = v + wip;                                       // :351
    }
    for (int j = lane; j < N; j += 64) l_bnd[j] = gbnd[j];
    WAVE_SYNC();
    if (lane == 0) {
#define V_(t, s) bvec[((t) - 1) * W + ((t) - 1 - (s))]
#define ID_(t, s) (((t) - 1 - (s)) < W ? bid[((t) - 1) * W + ((t) - 1 - (s))] : vid[(t) * ((t) - 1) / 2 + (s)])
        // ---- old tokens (utterances.py:159-174) before the boundaries are overwritten
        int no = 0, jp = 0;
        for (int j = 0; j < N; j++)
            if (l_bnd[j]) {
                int id = ID_(j + 1, jp);
                if (id >= 0) l_old[no++] = id;
                jp = j + 1;
            }
        // ---- A8 forward (kmeans_acoustic_wordseg.py:494-506)
        gam[0] = 0.0;
        for (int t = 1; t < N; t++) {
            int lo = t - W < 0 ? 0 : t - W;
            double best = NEG_INF_D;
            for (int s = lo; s < t; s++) {
                double v = V_(t, s) + gam[s];
                if (v > best) best = v;
            }
            gam[t] = best;
        }
        for (int j = 0; j < N; j++) l_bnd[j] = 0;
        l_bnd[N - 1] = 1;
        // ---- A8 backward (:510-553)
        int t = N;
        double total = 0.0;
        int lo = 0;
        for (;;) {
            lo = t - W < 0 ? 0 : t - W;
            bool all_inf = true;
            for (int s = lo; s < t; s++)
                if (V_(t, s) + gam[s] != NEG_INF_D) { all_inf = false; break; }
            if (all_inf) {
                while (all_inf) {
                    t = t - 1;
                    if (t == 0) break;
                    lo = t - W < 0 ? 0 : t - W;
                    all_inf = true;
                    for (int s = lo; s < t; s++)
                        if (V_(t, s) + gam[s] != NEG_INF_D) { all_inf = false; break; }
                }
                l_bnd[(t - 1 + N) % N] = 1;
            }
            int k = 1;
            if (t > 0) {
                double best = NEG_INF_D;
                bool first = true;
                for (int s = t - 1; s >= lo; s--) {
                    double v = V_(t, s) + gam[s];
                    if (first || v > best) { best = v; k = t - s; first = false; }
                }
                total += V_(t, t - k);
            } else {
                total += V_(N, N - 1);      // python vec[-1]: the last span [N-1, N)
            }
            if (t - k - 1 < 0) break;
            l_bnd[t - k - 1] = 1;
            t = t - k;
        }
        // ---- new tokens + their best components (:312-313)
        int nn = 0, bad = 0, nf = 0;
        const int Kact = *m.K;
        jp = 0;
        for (int j = 0; j < N; j++)
            if (l_bnd[j]) {
                int tt = j + 1, w = tt - 1 - jp;
                if (w >= W || bid[(tt - 1) * W + w] < 0) bad = 1;
                else {
                    l_new[nn] = bid[(tt - 1) * W + w];
                    l_newk[nn] = bk[(tt - 1) * W + w];
                    if (l_newk[nn] >= Kact) nf++;
                    nn++;
                }
                jp = j + 1;
            }
        out_total[u] = total;
        n_old[u] = no;
        n_new[u] = nn;
        if (n_flag) n_flag[u] = nf;
        l_cnt[0] = no;
        l_cnt[1] = nn;
        if (bad) atomicOr(status, 1);
#undef V_
#undef ID_
    }
    WAVE_SYNC();
    const int no = l_cnt[0], nn = l_cnt[1];
    for (int j = lane; j < N; j += 64) gbnd[j] = l_bnd[j];
    for (int j = lane; j < no; j += 64) old_tok[(int64_t)u * c.N_max + j] = l_old[j];
    for (int j = lane; j < nn; j += 64) {
        new_tok[(int64_t)u * c.N_max + j] = l_new[j];
        new_k[(int64_t)u * c.N_max + j] = l_newk[j];
    }
}

// Fast path of the per-utterance kernel for windows of at most 8 slices and utterances of at most 64
// landmarks (the common configuration: n_slices_max = 6).  Same arithmetic and the same decisions as
// k_kmeans_segment; what changes is how lane 0 gets at its operands.  The generic kernel walks the
// DP as a chain of dependent LDS round trips (store gamma[t], load it back for t+1, byte loads of the
// boundary flags with a branch on each); here the last eight gammas live in registers, the eight
// candidates of a step are fetched together (predicated, fully unrolled), and the boundary vectors
// are 64-bit masks.
__global__ void k_kmeans_segment_w8(segk_corpus c, segk_kmeans m, const int32_t *utts, int utt0, int n_utts,
                                    int n_max, double wip, segk_cand cand, uint8_t *boundaries, int32_t *old_tok,
                                    int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag,
                                    double *out_total, int32_t *status, int band_cap, int wave_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * (blockDim.x >> 6) + wv;
    if (slot >= n_utts) return;
    const int u = utts ? utts[slot] : utt0 + slot;
    const int N = c.lengths[u];
    const int W = (n_max > 0 && n_max < N) ? n_max : N;          // <= 8 (host checks n_max <= 8)
    const int nb = N * W;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
    const double *dur = c.durations + (int64_t)u * triMax;
    uint8_t *gbnd = boundaries + (int64_t)u * c.N_max;

    char *base = smem + (size_t)wv * wave_bytes;
    double *bvec = (double *)base;                    // [band_cap]
    double *gam = bvec + band_cap;                    // [N_max + 1]
    int32_t *bk = (int32_t *)(gam + c.N_max + 1);     // [band_cap]
    int32_t *bid = bk + band_cap;                     // [band_cap]
    int32_t *l_old = bid + band_cap;                  // [N_max]
    int32_t *l_new = l_old + c.N_max;                 // [N_max]
    int32_t *l_newk = l_new + c.N_max;                // [N_max]
    int32_t *l_cnt = l_newk + c.N_max;                // [4]: n_old, n_new, new boundary mask (2 words)

    for (int i = lane; i < nb; i += 64) {
        const int t = i / W + 1, w = i % W, s = t - 1 - w;
        int id = -1;
        double v = NEG_INF_D;
        int k = -1;
        if (s >= 0) {
            const int j = t * (t - 1) / 2 + s;
            id = vid[j];
            if (id >= 0) {
                k = cand.k[id];
                const double dd = dur[j];
                v = isnan(dd) ? NEG_INF_D : cand.s[id] * dd;      // :346-349
            }
        }
        bid[i] = id;
        bk[i] = k;
        bvec[i] = v + wip;                                       // :351
    }
    const unsigned long long oldb = __ballot(lane < N && gbnd[lane < N ? lane : 0] != 0);
    WAVE_SYNC();
    if (lane == 0) {
#define V_(t, s) bvec[((t) - 1) * W + ((t) - 1 - (s))]
#define ID_(t, s) (((t) - 1 - (s)) < W ? bid[((t) - 1) * W + ((t) - 1 - (s))] : vid[(t) * ((t) - 1) / 2 + (s)])
        // ---- old tokens (utterances.py:159-174)
        int no = 0, jp = 0;
        for (unsigned long long mb = oldb; mb; mb &= mb - 1) {
            const int j = __ffsll((long long)mb) - 1;
            const int id = ID_(j + 1, jp);
            if (id >= 0) l_old[no++] = id;
            jp = j + 1;
        }
        // ---- A8 forward (kmeans_acoustic_wordseg.py:494-506): g[w] = gamma[t - 1 - w]
        double g[8];
#pragma unroll
        for (int w = 0; w < 8; w++) g[w] = NEG_INF_D;
        g[0] = 0.0;
        gam[0] = 0.0;
        for (int t = 1; t < N; t++) {
            double v[8];
#pragma unroll
            for (int w = 0; w < 8; w++) {                  // unconditional loads from a clamped index, then the predicate
                const bool ok = w < W && t - 1 - w >= 0;
                v[w] = bvec[ok ? (t - 1) * W + w : 0];
            }
            double best = NEG_INF_D;
#pragma unroll
            for (int w = 7; w >= 0; w--) {                 // s ascending, as the reference's max() scans
                const bool ok = w < W && t - 1 - w >= 0;
                const double x = v[w] + g[w];
                if (ok && x > best) best = x;
            }
            gam[t] = best;
#pragma unroll
            for (int w = 7; w > 0; w--) g[w] = g[w - 1];
            g[0] = best;
        }
        unsigned long long newb = 1ull << (N - 1);
        // candidates of span end tt: are they all -inf; and the reversed np.argmax (shortest span on ties)
        auto eval = [&](int tt, int &kb) -> bool {
            double x[8];
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const bool ok = w < W && tt - 1 - w >= 0;
                x[w] = bvec[ok ? (tt - 1) * W + w : 0] + gam[ok ? tt - 1 - w : 0];
            }
            double best = NEG_INF_D;
            bool first = true, ai = true;
#pragma unroll
            for (int w = 0; w < 8; w++) {                  // s = tt - 1 - w descending
                const bool ok = w < W && tt - 1 - w >= 0;
                if (ok) {
                    if (x[w] != NEG_INF_D) ai = false;
                    if (first || x[w] > best) { best = x[w]; kb = w + 1; first = false; }
                }
            }
            return ai;
        };
        // ---- A8 backward (:510-553)
        int t = N;
        double total = 0.0;
        for (;;) {
            int kb = 1;
            bool all_inf = eval(t, kb);
            if (all_inf) {                                 // step back until some candidate is finite (:516-530)
                while (all_inf) {
                    t = t - 1;
                    if (t == 0) break;
                    all_inf = eval(t, kb);
                }
                newb |= 1ull << ((t - 1 + N) % N);
            }
            int k = 1;
            if (t > 0) {
                k = kb;
                total += V_(t, t - k);
            } else {
                total += V_(N, N - 1);      // python vec[-1]: the last span [N-1, N)
            }
            if (t - k - 1 < 0) break;
            newb |= 1ull << (t - k - 1);
            t = t - k;
        }
        // ---- new tokens + their best components (:312-313)
        int nn = 0, bad = 0, nf = 0;
        const int Kact = *m.K;
        jp = 0;
        for (unsigned long long mb = newb; mb; mb &= mb - 1) {
            const int j = __ffsll((long long)mb) - 1;
            const int tt = j + 1, w = tt - 1 - jp;
            if (w >= W || bid[(tt - 1) * W + w] < 0) bad = 1;
            else {
                l_new[nn] = bid[(tt - 1) * W + w];
                l_newk[nn] = bk[(tt - 1) * W + w];
                if (l_newk[nn] >= Kact) nf++;
                nn++;
            }
            jp = j + 1;
        }
        out_total[u] = total;
        n_old[u] = no;
        n_new[u] = nn;
        if (n_flag) n_flag[u] = nf;
        l_cnt[0] = no;
        l_cnt[1] = nn;
        l_cnt[2] = (int32_t)(newb & 0xffffffffull);
        l_cnt[3] = (int32_t)(newb >> 32);
        if (bad) atomicOr(status, 1);
#undef V_
#undef ID_
    }
    WAVE_SYNC();
    const int no = l_cnt[0], nn = l_cnt[1];
    const unsigned long long newb = ((unsigned long long)(unsigned int)l_cnt[3] << 32) | (unsigned int)l_cnt[2];
    if (lane < N) gbnd[lane] = (uint8_t)((newb >> lane) & 1ull);
    for (int j = lane; j < no; j += 64) old_tok[(int64_t)u * c.N_max + j] = l_old[j];
    for (int j = lane; j < nn; j += 64) {
        new_tok[(int64_t)u * c.N_max + j] = l_new[j];
        new_k[(int64_t)u * c.N_max + j] = l_newk[j];
    }
}

// The same with TWO utterances per wave (utterances of at most 32 landmarks): a launch over 10 000 utterances
// is two rounds of resident waves with one utterance each (7 waves per SIMD), and each round costs a wave's
// whole latency chain (dependent gathers, the serial DP); with two per wave it is one round.
__global__ void k_kmeans_segment_w8x2(segk_corpus c, segk_kmeans m, const int32_t *utts, int utt0, int n_utts,
                                    int n_max, double wip, segk_cand cand, uint8_t *boundaries, int32_t *old_tok,
                                    int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag,
                                    double *out_total, int32_t *status, int band_cap, int wave_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two utterances per wave: lanes 0..31 and 32..63 (N <= 32), the two DPs on lanes 0 and 32 in lockstep
    const int half = (threadIdx.x >> 5) & 1, lane = threadIdx.x & 31, wv = threadIdx.x >> 6;
    const int slot = (blockIdx.x * (blockDim.x >> 6) + wv) * 2 + half;
    const bool valid = slot < n_utts;
    const int u = valid ? (utts ? utts[slot] : utt0 + slot) : (utts ? utts[0] : utt0);
    const int N = valid ? c.lengths[u] : 0;
    const int W = (n_max > 0 && n_max < N) ? n_max : N;          // <= 8 (host checks n_max <= 8)
    const int nb = N * W;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
    const double *dur = c.durations + (int64_t)u * triMax;
    uint8_t *gbnd = boundaries + (int64_t)u * c.N_max;

    char *base = smem + (size_t)(wv * 2 + half) * wave_bytes;
    double *bvec = (double *)base;                    // [band_cap]
    double *gam = bvec + band_cap;                    // [N_max + 1]
    int32_t *bk = (int32_t *)(gam + c.N_max + 1);     // [band_cap]
    int32_t *bid = bk + band_cap;                     // [band_cap]
    int32_t *l_old = bid + band_cap;                  // [N_max]
    int32_t *l_new = l_old + c.N_max;                 // [N_max]
    int32_t *l_newk = l_new + c.N_max;                // [N_max]
    int32_t *l_cnt = l_newk + c.N_max;                // [4]: n_old, n_new, new boundary mask (2 words)

    for (int i = lane; i < nb; i += 32) {
        const int t = i / W + 1, w = i % W, s = t - 1 - w;
        int id = -1;
        double v = NEG_INF_D;
        int k = -1;
        if (s >= 0) {
            const int j = t * (t - 1) / 2 + s;
            id = vid[j];
            if (id >= 0) {
                k = cand.k[id];
                const double dd = dur[j];
                v = isnan(dd) ? NEG_INF_D : cand.s[id] * dd;      // :346-349
            }
        }
        bid[i] = id;
        bk[i] = k;
        bvec[i] = v + wip;                                       // :351
    }
    const unsigned long long oldb = (__ballot(lane < N && gbnd[lane < N ? lane : 0] != 0) >> (32 * half)) & 0xffffffffull;
    WAVE_SYNC();
    if (lane == 0 && valid) {
#define V_(t, s) bvec[((t) - 1) * W + ((t) - 1 - (s))]
#define ID_(t, s) (((t) - 1 - (s)) < W ? bid[((t) - 1) * W + ((t) - 1 - (s))] : vid[(t) * ((t) - 1) / 2 + (s)])
        // ---- old tokens (utterances.py:159-174)
        int no = 0, jp = 0;
        for (unsigned long long mb = oldb; mb; mb &= mb - 1) {
            const int j = __ffsll((long long)mb) - 1;
            const int id = ID_(j + 1, jp);
            if (id >= 0) l_old[no++] = id;
            jp = j + 1;
        }
        // ---- A8 forward (kmeans_acoustic_wordseg.py:494-506): g[w] = gamma[t - 1 - w]
        double g[8];
#pragma unroll
        for (int w = 0; w < 8; w++) g[w] = NEG_INF_D;
        g[0] = 0.0;
        gam[0] = 0.0;
        for (int t = 1; t < N; t++) {
            double v[8];
#pragma unroll
            for (int w = 0; w < 8; w++) {                  // unconditional loads from a clamped index, then the predicate
                const bool ok = w < W && t - 1 - w >= 0;
                v[w] = bvec[ok ? (t - 1) * W + w : 0];
            }
            double best = NEG_INF_D;
#pragma unroll
            for (int w = 7; w >= 0; w--) {                 // s ascending, as the reference's max() scans
                const bool ok = w < W && t - 1 - w >= 0;
                const double x = v[w] + g[w];
                if (ok && x > best) best = x;
            }
            gam[t] = best;
#pragma unroll
            for (int w = 7; w > 0; w--) g[w] = g[w - 1];
            g[0] = best;
        }
        unsigned long long newb = 1ull << (N - 1);
        // candidates of span end tt: are they all -inf; and the reversed np.argmax (shortest span on ties)
        auto eval = [&](int tt, int &kb) -> bool {
            double x[8];
#pragma unroll
            for (int w = 0; w < 8; w++) {
                const bool ok = w < W && tt - 1 - w >= 0;
                x[w] = bvec[ok ? (tt - 1) * W + w : 0] + gam[ok ? tt - 1 - w : 0];
            }
            double best = NEG_INF_D;
            bool first = true, ai = true;
#pragma unroll
            for (int w = 0; w < 8; w++) {                  // s = tt - 1 - w descending
                const bool ok = w < W && tt - 1 - w >= 0;
                if (ok) {
                    if (x[w] != NEG_INF_D) ai = false;
                    if (first || x[w] > best) { best = x[w]; kb = w + 1; first = false; }
                }
            }
            return ai;
        };
        // ---- A8 backward (:510-553)
        int t = N;
        double total = 0.0;
        for (;;) {
            int kb = 1;
            bool all_inf = eval(t, kb);
            if (all_inf) {                                 // step back until some candidate is finite (:516-530)
                while (all_inf) {
                    t = t - 1;
                    if (t == 0) break;
                    all_inf = eval(t, kb);
                }
                newb |= 1ull << ((t - 1 + N) % N);
            }
            int k = 1;
            if (t > 0) {
                k = kb;
                total += V_(t, t - k);
            } else {
                total += V_(N, N - 1);      // python vec[-1]: the last span [N-1, N)
            }
            if (t - k - 1 < 0) break;
            newb |= 1ull << (t - k - 1);
            t = t - k;
        }
        // ---- new tokens + their best components (:312-313)
        int nn = 0, bad = 0, nf = 0;
        const int Kact = *m.K;
        jp = 0;
        for (unsigned long long mb = newb; mb; mb &= mb - 1) {
            const int j = __ffsll((long long)mb) - 1;
            const int tt = j + 1, w = tt - 1 - jp;
            if (w >= W || bid[(tt - 1) * W + w] < 0) bad = 1;
            else {
                l_new[nn] = bid[(tt - 1) * W + w];
                l_newk[nn] = bk[(tt - 1) * W + w];
                if (l_newk[nn] >= Kact) nf++;
                nn++;
            }
            jp = j + 1;
        }
        out_total[u] = total;
        n_old[u] = no;
        n_new[u] = nn;
        if (n_flag) n_flag[u] = nf;
        l_cnt[0] = no;
        l_cnt[1] = nn;
        l_cnt[2] = (int32_t)(newb & 0xffffffffull);
        l_cnt[3] = (int32_t)(newb >> 32);
        if (bad) atomicOr(status, 1);
#undef V_
#undef ID_
    }
    WAVE_SYNC();
    if (!valid) return;
    const int no = l_cnt[0], nn = l_cnt[1];
    const unsigned long long newb = ((unsigned long long)(unsigned int)l_cnt[3] << 32) | (unsigned int)l_cnt[2];
    if (lane < N) gbnd[lane] = (uint8_t)((newb >> lane) & 1ull);
    for (int j = lane; j < no; j += 32) old_tok[(int64_t)u * c.N_max + j] = l_old[j];
    for (int j = lane; j < nn; j += 32) {
        new_tok[(int64_t)u * c.N_max + j] = l_new[j];
        new_k[(int64_t)u * c.N_max + j] = l_newk[j];
    }
}

// ======================================================================================
// A11 sequential: del_item / add_item / clean_components for ONE utterance, one workgroup,
// thread d owns dimension d (kmeans_components.py:93-166, 263-266).
// ======================================================================================
template <typename XT>
__device__ void dev_del_component(const segk_corpus &c, segk_kmeans &m, int k, int *shK)
{
    // caller guarantees uniform control flow; K already decremented into *shK by thread 0
    const int tid = threadIdx.x, nt = blockDim.x;
    const int D = c.D;
    const int K = *shK;
    XT *means = (XT *)m.means;
    const XT *rnd = (const XT *)m.random_means;
    if (k != K) {
        const double cntK = (double)m.counts[K];
        for (int d = tid; d < D; d += nt) {
            double v = m.mean_numerators[(int64_t)K * D + d];
            m.mean_numerators[(int64_t)k * D + d] = v;
            means[(int64_t)k * D + d] = (XT)(v / cntK);
        }
        for (int64_t e = tid; e < c.n_emb; e += nt)
            if (m.assignments[e] == K) m.assignments[e] = k;
    }
    __syncthreads();
    for (int d = tid; d < D; d += nt) {
        m.mean_numerators[(int64_t)K * D + d] = 0.0;
        means[(int64_t)K * D + d] = rnd[(int64_t)K * D + d];
    }
    if (tid == 0) {
        if (k != K) m.counts[k] = m.counts[K];
        m.counts[K] = 0;
    }
    __syncthreads();
}

template <typename XT>
__device__ void dev_clean_components(const segk_corpus &c, segk_kmeans &m, int *shK, int *sh_i)
{
    // kmeans_components.py:263-266: every empty component, highest index first.  The empties are
    // found with one parallel pass (a deletion moves the last ACTIVE row down, it never creates or
    // hides an empty row below the current one), then deleted one by one in the reference's order.
    __shared__ unsigned int empty_bits[1024];         // K_max <= 32768
    const int tid = threadIdx.x, nt = blockDim.x;
    const int K0 = *shK;
    const int nwords = (K0 + 31) >> 5;
    (void)sh_i;
    if (nwords > 1024) {                              // beyond the bitmap: the plain scan
        for (int k = K0 - 1; k >= 0; k--) {
            if (tid == 0) *sh_i = (m.counts[k] == 0) ? 1 : 0;
            __syncthreads();
            const int empty = *sh_i;
            __syncthreads();
            if (empty) {
                if (tid == 0) *shK = *shK - 1;
                __syncthreads();
                dev_del_component<XT>(c, m, k, shK);
            }
        }
        return;
    }
    for (int w = tid; w < nwords; w += nt) empty_bits[w] = 0;
    __syncthreads();
    for (int k = tid; k < K0; k += nt)
        if (m.counts[k] == 0) atomicOr(&empty_bits[k >> 5], 1u << (k & 31));
    __syncthreads();
    for (int w = nwords - 1; w >= 0; w--) {
        unsigned int bits = empty_bits[w];             // uniform across the workgroup
        while (bits) {
            const int bit = 31 - __clz((int)bits);
            bits &= ~(1u << bit);
            __syncthreads();
            if (tid == 0) *shK = *shK - 1;
            __syncthreads();
            dev_del_component<XT>(c, m, w * 32 + bit, shK);
        }
    }
}

template <typename XT>
__device__ void dev_add_item(const segk_corpus &c, segk_kmeans &m, int64_t e, int k_in, int *shK, int *sh_i,
                             int64_t *sh_l, int32_t *status)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int D = c.D;
    if (tid == 0) {
        int k = k_in;
        int K = *shK;
        if (k > K) k = K;
        if (k == K) *shK = K + 1;
        if (m.assignments[e] != -1) atomicOr(status, 2);     // kmeans_components.py:101 assert
        m.counts[k] += 1;
        m.assignments[e] = k;
        *sh_i = k;
        *sh_l = m.counts[k];
    }
    __syncthreads();
    const int k = *sh_i;
    const double cnt = (double)*sh_l;
    const XT *X = (const XT *)c.X;
    XT *means = (XT *)m.means;
    for (int d = tid; d < D; d += nt) {
        double v = m.mean_numerators[(int64_t)k * D + d] + (double)X[e * c.ldx + d];
        m.mean_numerators[(int64_t)k * D + d] = v;
        means[(int64_t)k * D + d] = (XT)(v / cnt);
    }
    __syncthreads();
}

template <typename XT>
__device__ void dev_del_item(const segk_corpus &c, segk_kmeans &m, int64_t e, int *sh_i, int64_t *sh_l)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int D = c.D;
    if (tid == 0) {
        int k = m.assignments[e];
        if (k != -1) {
            m.counts[k] -= 1;
            m.assignments[e] = -1;
            *sh_l = m.counts[k];
        }
        *sh_i = k;
    }
    __syncthreads();
    const int k = *sh_i;
    if (k != -1) {
        const int64_t cnt = *sh_l;
        const XT *X = (const XT *)c.X;
        XT *means = (XT *)m.means;
        for (int d = tid; d < D; d += nt) {
            double v = m.mean_numerators[(int64_t)k * D + d] - (double)X[e * c.ldx + d];
            m.mean_numerators[(int64_t)k * D + d] = v;
            if (cnt != 0) means[(int64_t)k * D + d] = (XT)(v / (double)cnt);
        }
    }
    __syncthreads();
}

// op: 0 = utterance update (del old, add new, clean), 1 = add_item(i,k), 2 = del_item(i),
//     3 = clean_components
template <typename XT>
__global__ void k_kmeans_update(segk_corpus c, segk_kmeans m, int op, int utt, int64_t item, int k_item,
                                const int32_t *old_tok, const int32_t *new_tok, const int32_t *new_k,
                                const int32_t *n_old, const int32_t *n_new, int32_t *status)
{
    __shared__ int shK, sh_i;
    __shared__ int64_t sh_l;
    if (threadIdx.x == 0) shK = *m.K;
    __syncthreads();
    if (op == 0) {
        const int no = n_old[utt], nn = n_new[utt];
        for (int t = 0; t < no; t++) dev_del_item<XT>(c, m, old_tok[(int64_t)utt * c.N_max + t], &sh_i, &sh_l);
        for (int t = 0; t < nn; t++)
            dev_add_item<XT>(c, m, new_tok[(int64_t)utt * c.N_max + t], new_k[(int64_t)utt * c.N_max + t],
                             &shK, &sh_i, &sh_l, status);
        dev_clean_components<XT>(c, m, &shK, &sh_i);
    } else if (op == 1) {
        dev_add_item<XT>(c, m, item, k_item, &shK, &sh_i, &sh_l, status);
    } else if (op == 2) {
        dev_del_item<XT>(c, m, item, &sh_i, &sh_l);
    } else if (op == 3) {
        dev_clean_components<XT>(c, m, &shK, &sh_i);
    } else if (op == 4) {     // del_component(k_item)  (kmeans_components.py:149-166)
        if (threadIdx.x == 0) shK = shK - 1;
        __syncthreads();
        dev_del_component<XT>(c, m, k_item, &shK);
    }
    __syncthreads();
    if (threadIdx.x == 0) *m.K = shK;
}

// ======================================================================================
// A11 batch-synchronous statistics (spec: oracle/np_oracle.py kmeans_batch_sweep)
// ======================================================================================
// (1) one workgroup: (a) exclusive prefix sum of n_new over the local utterances ->
//     tok_off[u - lo] (tok_off[hi - lo] = number of local tokens); (b) collect, in token order,
//     the new tokens whose argmax is an inactive row (k >= K; per-utterance counts n_flag come
//     from the segment kernel): flag_buf[0] = count, then (slot = utt*N_max + t, k) pairs.
__global__ void k_batch_collect(segk_corpus c, segk_kmeans m, int lo, int hi, const int32_t *new_k,
                                const int32_t *n_new, const int32_t *n_flag, int32_t *tok_off,
                                int32_t *flag_buf, int cap)
{
    // every thread owns a run of `per` consecutive utterances: local sums, ONE workgroup scan of the
    // 1024 run totals (wave scan + 16 wave totals), then the run is walked again with its offsets
    __shared__ int s_wave[16], s_wave2[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int K = *m.K;
    const int n = hi - lo;
    const int per = (n + nt - 1) / nt;
    const int u_lo = lo + tid * per, u_hi = (u_lo + per < hi) ? u_lo + per : hi;
    int mine = 0, ntok = 0;
    for (int u = u_lo; u < u_hi; u++) { ntok += n_new[u]; mine += n_flag[u]; }
    int incl = mine, incl2 = ntok;
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o), v2 = __shfl_up(incl2, o);
        if (lane >= o) { incl += v; incl2 += v2; }
    }
    if (lane == 63) { s_wave[wv] = incl; s_wave2[wv] = incl2; }
    __syncthreads();
    int woff = 0, woff2 = 0, total = 0, total2 = 0;
    for (int w2 = 0; w2 < nw; w2++) {
        if (w2 < wv) { woff += s_wave[w2]; woff2 += s_wave2[w2]; }
        total += s_wave[w2];
        total2 += s_wave2[w2];
    }
    int off2 = woff2 + incl2 - ntok;          // tokens before this run
    int off = woff + incl - mine;             // flagged tokens before this run
    for (int u = u_lo; u < u_hi; u++) {
        const int nt_u = n_new[u];
        tok_off[u - lo] = off2;
        off2 += nt_u;
        if (n_flag[u] > 0)
            for (int t = 0; t < nt_u; t++) {
                const int k = new_k[(int64_t)u * c.N_max + t];
                if (k >= K) {
                    if (off < cap) {
                        flag_buf[1 + 2 * off] = u * c.N_max + t;
                        flag_buf[2 + 2 * off] = k;
                    }
                    off++;
                }
            }
    }
    if (tid == 0) {
        flag_buf[0] = total;
        tok_off[n] = total2;
    }
}

// (2) replay the `k > K -> K` clamp (kmeans_components.py:103-106) over the flagged tokens of
//     ALL ranks in rank order; patch the local new_k; set K.  One wave: the (short) lists are
//     fetched in parallel, lane 0 replays them.
__global__ void k_batch_resolve(segk_kmeans m, const int32_t *flag_all, int n_ranks, int my_rank, int cap,
                                int32_t *new_k, int32_t *status)
{
    __shared__ int32_t l_k[1024], l_slot[1024];
    const int lane = threadIdx.x;
    int K = *m.K;
    for (int r = 0; r < n_ranks; r++) {
        const int32_t *fb = flag_all + (int64_t)r * (1 + 2 * cap);
        int cnt = fb[0];
        if (cnt > cap) { if (lane == 0) atomicOr(status, 4); cnt = cap; }
        for (int q0 = 0; q0 < cnt; q0 += 1024) {
            int nq = cnt - q0 < 1024 ? cnt - q0 : 1024;
            for (int q = lane; q < nq; q += 64) {
                l_slot[q] = fb[1 + 2 * (q0 + q)];
                l_k[q] = fb[2 + 2 * (q0 + q)];
            }
            __syncthreads();
            if (lane == 0) {
                for (int q = 0; q < nq; q++) {
                    int k = l_k[q];
                    if (k > K) k = K;
                    if (k == K) K++;
                    l_k[q] = k;
                }
            }
            __syncthreads();
            K = __shfl(K, 0);
            if (r == my_rank)
                for (int q = lane; q < nq; q += 64) new_k[l_slot[q]] = l_k[q];
            __syncthreads();
        }
    }
    if (lane == 0) *m.K = K;
}

// (3) compact the local tokens in token order: ctok[tok_off[u-lo] + t] = (embedding, component)
__global__ void k_batch_compact(segk_corpus c, int lo, int hi, const int32_t *new_tok, const int32_t *new_k,
                                const int32_t *n_new, const int32_t *tok_off, int32_t *ctok_id, int32_t *ctok_k)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)(hi - lo) * c.N_max;
    if (idx >= tot) return;
    int u = lo + (int)(idx / c.N_max), t = (int)(idx % c.N_max);
    if (t < n_new[u]) {
        int p = tok_off[u - lo] + t;
        ctok_id[p] = new_tok[(int64_t)u * c.N_max + t];
        ctok_k[p] = new_k[(int64_t)u * c.N_max + t];
    }
}

// (4) per statistics block and component: sequential fp64 sum over the block's tokens in
//     token order.  A workgroup = (block, 8 consecutive components), one wave per component;
//     lanes own dimensions.  The block's token keys are staged in LDS chunk by chunk with
//     coalesced loads; a wave compacts its matching token ids (token order) into an LDS list and
//     drains it 16 rows at a time -- the row loads are unconditional (clamped index, select after
//     the load) so that all 16 are in flight together; the adds stay strictly in order.
#define PART_CHUNK 8192
#define PART_MLIST 512
#define PART_BATCH 16
template <typename XT>
__global__ __launch_bounds__(512) void k_batch_partials(
    segk_corpus c, segk_kmeans m, const int32_t *blk_lo, int n_blocks, int lo, const int32_t *tok_off,
    const int32_t *ctok_id, const int32_t *ctok_k, const double *out_total, double *part_sum,
    int64_t *part_cnt, double *part_tot, int dbg)
{
    __shared__ __attribute__((aligned(16))) int32_t keys[PART_CHUNK];
    __shared__ int32_t mlists[8 * PART_MLIST];
    __shared__ int32_t wsum[2][8];
    const int groups = (m.K_max + 7) / 8;
    const int b = blockIdx.x / groups, kg = blockIdx.x % groups;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = kg * 8 + wv;
    const bool active = k < m.K_max;
    const int D = c.D;
    const XT *X = (const XT *)c.X;
    const int u0 = blk_lo[b], u1 = blk_lo[b + 1];
    const int p0 = tok_off[u0 - lo], p1 = tok_off[u1 - lo];
    int32_t *mlist = mlists + wv * PART_MLIST;
    constexpr int MAXR = 2;                       // 128 dims per pass over the tokens
    for (int d0 = 0; d0 < D; d0 += 64 * MAXR) {
        double acc[MAXR];
        int dcl[MAXR];                            // clamped dimension (always a valid address)
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            acc[r] = 0.0;
            const int d = d0 + r * 64 + lane;
            dcl[r] = d < D ? d : 0;
        }
        int64_t cnt = 0;
        for (int pc = p0; pc < p1; pc += PART_CHUNK) {
            const int nch = p1 - pc < PART_CHUNK ? p1 - pc : PART_CHUNK;
            __syncthreads();
            {   // coalesced staging, 4 independent loads in flight per thread
                const int nt4 = 4 * blockDim.x;
                for (int i0 = threadIdx.x; i0 < nch; i0 += nt4) {
                    int v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = i0 + q * blockDim.x;
                        v[q] = ctok_k[pc + (i < nch ? i : 0)];
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int i = i0 + q * blockDim.x;
                        if (i < nch) keys[i] = v[q];
                    }
                }
            }
            __syncthreads();
            // One cooperative pass of the workgroup over the chunk: the tokens of its EIGHT components, in token
            // order, compacted in place to the front of keys[] as (position in chunk) * 8 + (component & 7).
            // Before, every wave scanned every key for its own component -- 8 000 waves x 8 750 compares were
            // 32 of the kernel's 58 us.  A thread owns four consecutive keys; a sub-chunk of 2048 keys is read
            // into registers by everybody before anybody writes into its range (the barrier), and the write
            // cursor never passes the keys already consumed.
            int wgn = 0;         // workgroup-uniform: compacted entries so far
            for (int sb = 0, it = 0; sb < nch; sb += 2048, it++) {
                const int i0 = sb + 4 * threadIdx.x;
                int4 kv = make_int4(-1, -1, -1, -1);
                if (i0 + 3 < nch) kv = *reinterpret_cast<const int4 *>(keys + i0);
                else {
                    if (i0 < nch) kv.x = keys[i0];
                    if (i0 + 1 < nch) kv.y = keys[i0 + 1];
                    if (i0 + 2 < nch) kv.z = keys[i0 + 2];
                }
                const int kk[4] = {kv.x, kv.y, kv.z, kv.w};
                int mine = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) mine += (kk[j] >= 0 && (kk[j] >> 3) == kg);
                int incl = mine;                               // inclusive prefix over the wave's lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl, o);
                    if (lane >= o) incl += t;
                }
                if (lane == 63) wsum[it & 1][wv] = incl;
                __syncthreads();
                int wbase = wgn, tot = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    const int t = wsum[it & 1][w];
                    if (w < wv) wbase += t;
                    tot += t;
                }
                int pos = wbase + incl - mine;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (kk[j] >= 0 && (kk[j] >> 3) == kg) keys[pos++] = ((i0 + j) << 3) | (kk[j] & 7);
                wgn += tot;
            }
            __syncthreads();
            if (!active || (dbg & 1)) continue;
            int nm = 0;          // wave-uniform length of the match list
            for (int pb = 0; pb < wgn; pb += 256) {
                // four entries per lane per iteration (entries pb + lane + 64 j): token order = j-major
                int mt[4], ent[4];
                unsigned long long bal[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = pb + lane + 64 * j;
                    ent[j] = keys[i < wgn ? i : 0];
                    mt[j] = (i < wgn) && ((ent[j] & 7) == wv);
                    bal[j] = __ballot(mt[j]);
                }
                if (bal[0] | bal[1] | bal[2] | bal[3]) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (mt[j])
                            mlist[nm + __popcll(bal[j] & ((1ull << lane) - 1ull))] = pc + (ent[j] >> 3);   // the token's POSITION: its id is fetched in the drain
                        nm += __popcll(bal[j]);
                    }
                }
                if (nm > PART_MLIST - 256 || (pb + 256 >= wgn && nm > 0)) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (int q0 = 0; q0 < nm && !(dbg & 2); q0 += PART_BATCH) {
                        double xv[PART_BATCH][MAXR];
#pragma unroll
                        for (int q = 0; q < PART_BATCH; q++) {
                            const int e = ctok_id[mlist[q0 + q < nm ? q0 + q : q0]];       // clamped: always valid
#pragma unroll
                            for (int r = 0; r < MAXR; r++) xv[q][r] = (double)X[(int64_t)e * c.ldx + dcl[r]];
                        }
#pragma unroll
                        for (int q = 0; q < PART_BATCH; q++) {
                            const bool ok = q0 + q < nm;
#pragma unroll
                            for (int r = 0; r < MAXR; r++) acc[r] += ok ? xv[q][r] : 0.0;
                        }
                    }
                    cnt += nm;
                    nm = 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (active) {
#pragma unroll
            for (int r = 0; r < MAXR; r++) {
                int d = d0 + r * 64 + lane;
                if (d < D) part_sum[((int64_t)b * m.K_max + k) * D + d] = acc[r];
            }
            if (lane == 0 && d0 == 0) part_cnt[(int64_t)b * m.K_max + k] = cnt;
        }
    }
    if (kg == 0 && !(dbg & 4)) {
        // sequential (utterance order) sum of the block's totals, staged through LDS so that the
        // single summing thread never waits on global memory
        double *stage = reinterpret_cast<double *>(keys);
        double s = 0.0;
        for (int uc = u0; uc < u1; uc += PART_CHUNK / 2) {
            const int nu = u1 - uc < PART_CHUNK / 2 ? u1 - uc : PART_CHUNK / 2;
            __syncthreads();
            for (int i = threadIdx.x; i < nu; i += blockDim.x) stage[i] = out_total[uc + i];
            __syncthreads();
            if (threadIdx.x == 0) {
                // strictly sequential adds; the LDS reads are issued 16 at a time
                int i = 0;
                for (; i + 16 <= nu; i += 16) {
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = stage[i + q];
#pragma unroll
                    for (int q = 0; q < 16; q++) s += v[q];
                }
                for (; i < nu; i++) s += stage[i];
            }
        }
        if (threadIdx.x == 0) part_tot[b] = s;
    }
}

// Partials of block b live at  base + (b / nbl) * rank_stride + (b % nbl) * blk_stride
// (units: 8-byte words): `nbl` blocks per rank, packed rank after rank by the all-gather.
struct PartAddr {
    int nbl;
    int64_t rank_stride, blk_stride;
    __device__ __forceinline__ int64_t operator()(int b) const
    {
        return (int64_t)(b / nbl) * rank_stride + (int64_t)(b % nbl) * blk_stride;
    }
};

// balanced binary tree over n <= 64 parts, pairing neighbours level by level, odd one carried
__device__ __forceinline__ double tree_sum_d(const double *p, const PartAddr &pa, int n)
{
    double buf[64];
    for (int i = 0; i < n; i++) buf[i] = p[pa(i)];
    while (n > 1) {
        int o = 0;
        for (int i = 0; i + 1 < n; i += 2) buf[o++] = buf[i] + buf[i + 1];
        if (n & 1) buf[o++] = buf[n - 1];
        n = o;
    }
    return buf[0];
}

// (5a) combine the partials of all blocks, means = numerators / counts for active rows
template <typename XT>
__global__ void k_batch_combine(segk_corpus c, segk_kmeans m, const double *part_sum, const int64_t *part_cnt,
                                const double *part_tot, int n_blocks, int nbl, int64_t rank_stride,
                                double *out_scalars)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = c.D;
    const int K = *m.K;
    const PartAddr pa_sum{nbl, rank_stride, (int64_t)m.K_max * D};
    const PartAddr pa_cnt{nbl, rank_stride, (int64_t)m.K_max};
    const PartAddr pa_tot{nbl, rank_stride, 1};
    if (idx < (int64_t)m.K_max * D) {
        int k = (int)(idx / D);
        double v = tree_sum_d(part_sum + idx, pa_sum, n_blocks);
        int64_t cnt = 0;
        for (int b = 0; b < n_blocks; b++) cnt += part_cnt[pa_cnt(b) + k];
        m.mean_numerators[idx] = v;
        if (k < K && cnt != 0) ((XT *)m.means)[idx] = (XT)(v / (double)cnt);
        if (idx % D == 0) m.counts[k] = cnt;
    }
    if (idx == 0) out_scalars[0] = tree_sum_d(part_tot, pa_tot, n_blocks);
}

// (5b) clean_components (kmeans_components.py:263-266) with a relabel table instead of a scan
//      of `assignments` per deletion.  Single workgroup: the empty rows are found in parallel
//      (bitmap), then deleted one by one in descending order as the reference does;
//      remap [K_max]: original label -> final row.  Also n_tokens = sum(counts).
template <typename XT>
__global__ void k_batch_clean(segk_corpus c, segk_kmeans m, int32_t *remap, double *out_scalars)
{
    // The reference deletes the empty components one at a time in descending order, each time moving
    // the last active row into the hole (kmeans_components.py:129-151, 263-266).  Because the holes
    // above the current one are already gone, the row that moves is never empty, every moved row
    // originates at or above the final K and lands below it -- so the bookkeeping (which original row
    // ends where) is replayed serially on indices only, and the rows are then moved in parallel.
    __shared__ int shK, n_holes;
    __shared__ unsigned int bitmap[256];             // K_max <= 8192
    __shared__ unsigned short pos2orig[8192];        // position -> original row living there
    __shared__ unsigned short holes[8192];           // hole positions, descending
    __shared__ long long red[256];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int D = c.D;
    XT *means = (XT *)m.means;
    const XT *rnd = (const XT *)m.random_means;
    const int K0 = *m.K;
    const int nwords = (K0 + 31) / 32;
    for (int w = tid; w < nwords; w += nt) bitmap[w] = 0;
    for (int k = tid; k < m.K_max; k += nt) remap[k] = k;
    for (int k = tid; k < K0; k += nt) pos2orig[k] = (unsigned short)k;
    __syncthreads();
    long long csum = 0;
    for (int k = tid; k < m.K_max; k += nt) {
        long long cn = m.counts[k];
        csum += cn;
        if (k < K0 && cn == 0) atomicOr(&bitmap[k >> 5], 1u << (k & 31));
    }
    red[tid] = csum;
    __syncthreads();
    for (int o = nt >> 1; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        out_scalars[2] = (double)red[0];
        int K = K0, nh = 0;
        for (int w = nwords - 1; w >= 0; w--) {
            unsigned int bits = bitmap[w];
            while (bits) {
                const int bit = 31 - __clz((int)bits);
                bits &= ~(1u << bit);
                const int k = w * 32 + bit;
                K--;
                if (k != K) pos2orig[k] = pos2orig[K];
                holes[nh++] = (unsigned short)k;
            }
        }
        shK = K;
        n_holes = nh;
    }
    __syncthreads();
    const int K = shK, nh = n_holes;
    // move: one wave per filled hole below the final K
    for (int h = tid >> 6; h < nh; h += nt >> 6) {
        const int k = holes[h];
        if (k >= K) continue;
        const int src = pos2orig[k];
        const double cnt = (double)m.counts[src];
        for (int d = tid & 63; d < D; d += 64) {
            const double v = m.mean_numerators[(int64_t)src * D + d];
            m.mean_numerators[(int64_t)k * D + d] = v;
            means[(int64_t)k * D + d] = (XT)(v / cnt);
        }
        if ((tid & 63) == 0) {
            m.counts[k] = m.counts[src];
            remap[src] = k;
        }
    }
    __syncthreads();
    // rows [K, K0) are inactive again
    for (int64_t j = tid; j < (int64_t)(K0 - K) * D; j += nt) {
        const int64_t row = K + j / D, d = j % D;
        m.mean_numerators[row * D + d] = 0.0;
        means[row * D + d] = rnd[row * D + d];
    }
    for (int k = K + tid; k < K0; k += nt) m.counts[k] = 0;
    if (tid == 0) {
        *m.K = K;
        out_scalars[1] = (double)K;
    }
}

// (5c) final labels of the local tokens
__global__ void k_batch_relabel(segk_corpus c, int lo, int hi, int32_t *new_k, const int32_t *n_new,
                                const int32_t *remap, double *zero_me)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && zero_me) *zero_me = 0.0;          // max |m|^2 of the prepare that follows on the stream (saves its memset)
    int64_t tot = (int64_t)(hi - lo) * c.N_max;
    if (idx >= tot) return;
    int u = lo + (int)(idx / c.N_max), t = (int)(idx % c.N_max);
    if (t < n_new[u]) {
        int64_t p = (int64_t)u * c.N_max + t;
        new_k[p] = remap[new_k[p]];
    }
}

// `assignments` from the token lists of utterances [lo, hi) (everything else unassigned):
// the batch sweep does not touch `assignments`; it is materialised on demand.
__global__ void k_assign_from_tokens(segk_corpus c, segk_kmeans m, int lo, int hi, const int32_t *new_tok,
                                     const int32_t *new_k, const int32_t *n_new)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t tot = (int64_t)(hi - lo) * c.N_max;
    if (idx >= tot) return;
    int u = lo + (int)(idx / c.N_max), t = (int)(idx % c.N_max);
    if (t < n_new[u]) {
        int64_t p = (int64_t)u * c.N_max + t;
        m.assignments[new_tok[p]] = new_k[p];
    }
}

// sum_neg_sqrd_norm record metric (kmeans_components.py:234-247); tolerance-level parity
template <typename XT>
__global__ void k_kmeans_sum_neg_sqrd_norm(segk_corpus c, segk_kmeans m, double *out)
{
    const int64_t e = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    if (e < c.n_emb) {
        int k = m.assignments[e];
        if (k >= 0) {
            double cnt = (double)m.counts[k];
            for (int d = lane; d < c.D; d += 64) {
                double delta = m.mean_numerators[(int64_t)k * c.D + d] / cnt
                               - (double)((const XT *)c.X)[e * c.ldx + d];
                s += delta * delta;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[16];
    if (lane == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < (int)(blockDim.x / 64); w++) tot += part[w];
        if (tot != 0.0) atomicAdd(out, -tot);
    }
}

// ======================================================================================
// KMeansComponents.__init__ (kmeans_components.py:59-81): add_item(i, k) for k ascending and
// i ascending within k == per component a sequential fp64 sum over its items in ascending
// row order.  One wave per component scans `assignments`.
// ======================================================================================
template <typename XT>
__global__ void k_kmeans_init_stats(segk_corpus c, segk_kmeans m)
{
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (k >= m.K_max) return;
    const int D = c.D;
    const XT *X = (const XT *)c.X;
    XT *means = (XT *)m.means;
    const XT *rnd = (const XT *)m.random_means;
    constexpr int MAXR = 8;
    for (int d0 = 0; d0 < D; d0 += 64 * MAXR) {
        double acc[MAXR];
#pragma unroll
        for (int r = 0; r < MAXR; r++) acc[r] = 0.0;
        int64_t cnt = 0;
        for (int64_t e0 = 0; e0 < c.n_emb; e0 += 64) {
            int64_t e = e0 + lane;
            int match = (e < c.n_emb) && (m.assignments[e] == k);
            unsigned long long bal = __ballot(match);
            while (bal) {
                int src = __ffsll((long long)bal) - 1;
                bal &= bal - 1;
                int64_t ee = e0 + src;
                cnt++;
#pragma unroll
                for (int r = 0; r < MAXR; r++) {
                    int d = d0 + r * 64 + lane;
                    if (d < D) acc[r] += (double)X[ee * c.ldx + d];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            int d = d0 + r * 64 + lane;
            if (d < D) {
                m.mean_numerators[(int64_t)k * D + d] = acc[r];
                means[(int64_t)k * D + d] = cnt ? (XT)(acc[r] / (double)cnt) : rnd[(int64_t)k * D + d];
            }
        }
        if (lane == 0 && d0 == 0) {
            m.counts[k] = cnt;
            if (cnt) atomicMax(m.K, k + 1);
        }
    }
}

// ======================================================================================
// Function-level DPs on caller-supplied vectors (drop-in for the module functions
// forward_backward_kmeans_viterbi / forward_backward / forward_backward_viterbi):
// one thread per problem, triangular layout exactly as the reference receives it.
//   kind 0: A8 max-plus (kmeans_acoustic_wordseg.py:449-555)
//   kind 1: A7 viterbi  (unigram_acoustic_wordseg.py:759-864)
//   kind 2: A6 forward filtering / backward sampling (:653-756), uniforms supplied
// ======================================================================================
__device__ double dev_logsumexp(const double *a, int n)      // _cython_utils.pyx:13-25
{
    double mx = a[0], s = 0.0;
    for (int j = 1; j < n; j++)
        if (a[j] > mx) mx = a[j];
    for (int j = 0; j < n; j++) s += exp(a[j] - mx);
    return log(s) + mx;
}

__global__ void k_dp_tri(int kind, const double *vecs, const int32_t *Ns, const int64_t *offs, int n_prob,
                         int n_min, int n_max, double log_p_continue, double anneal_temp,
                         const double *uniforms, int64_t u_stride, uint8_t *bounds, int64_t b_stride,
                         double *totals, int32_t *n_draws, int32_t *status, double *work, int64_t w_stride)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_prob) return;
    const int N = Ns[p];
    const double *vec = vecs + offs[p];
    const int64_t L = (int64_t)N * (N + 1) / 2;
    uint8_t *bnd = bounds + p * b_stride;
    double *a = work + p * w_stride;            // [N]
    double *w = a + N;                          // [N+1]
    double *pr = w + N + 1;                     // [N+1]
    const double *us = uniforms ? uniforms + p * u_stride : nullptr;
    for (int j = 0; j < N; j++) { a[j] = 1.0; bnd[j] = 0; }
    bnd[N - 1] = 1;
    a[0] = 0.0;
    int64_t i = 0;
    for (int t = 1; t < N; t++) {
        int lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
        int n = t - lo;
        bool all_inf = true;
        double best = NEG_INF_D;
        for (int s = lo; s < t; s++) {
            double v = vec[i + s] + a[s];
            w[s - lo] = v;
            if (v != NEG_INF_D) all_inf = false;
            if (v > best) best = v;
        }
        if (kind == 2) a[t] = all_inf ? NEG_INF_D : dev_logsumexp(w, n) + log_p_continue;
        else a[t] = best;
        i += t;
    }
    int t = N, nd = 0, lo = 0;
    double total = 0.0;
    for (;;) {
        i = (int64_t)(t - 1) * t / 2;
        lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
        bool all_inf = true;
        for (int s = lo; s < t; s++)
            if (vec[i + s] + a[s] != NEG_INF_D) { all_inf = false; break; }
        if (all_inf) {
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (int64_t)(t - 1) * t / 2;
                lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
                all_inf = true;
                for (int s = lo; s < t; s++)
                    if (vec[i + s] + a[s] != NEG_INF_D) { all_inf = false; break; }
            }
            bnd[(t - 1 + N) % N] = 1;
        }
        int k = 1;
        int n = 1;
        if (t > 0) {
            n = t - lo;
            for (int s = lo; s < t; s++) w[s - lo] = vec[i + s] + a[s];
        } else {
            w[0] = NEG_INF_D;
        }
        if (kind == 0) {
            if (t > 0) {
                double best = NEG_INF_D;
                bool first = true;
                for (int s = t - 1; s >= lo; s--) {
                    double v = w[s - lo];
                    if (first || v > best) { best = v; k = t - s; first = false; }
                }
            }
        } else if (kind == 1) {
            if (t > 0) {
                double lse = dev_logsumexp(w, n);
                double best = 0.0;
                bool first = true;
                for (int s = t - 1; s >= lo; s--) {
                    double q = exp(w[s - lo] - lse);
                    if (first || q > best) { best = q; k = t - s; first = false; }
                }
            }
        } else {
            double lse = dev_logsumexp(w, n);
            if (anneal_temp != 1.0) {
                for (int j = 0; j < n; j++) pr[j] = w[n - 1 - j] - lse;
                double inv = 1. / anneal_temp;
                for (int j = 0; j < n; j++) w[j] = inv * pr[j];
                double lse2 = dev_logsumexp(w, n);
                for (int j = 0; j < n; j++) pr[j] = exp(w[j] - lse2);
            } else {
                for (int j = 0; j < n; j++) pr[j] = exp(w[n - 1 - j] - lse);
            }
            double uu = us[nd];
            nd++;
            int kk = n - 1;
            for (int j = 0; j < n; j++) {
                uu = uu - pr[j];
                if (uu < 0) { kk = j; break; }
            }
            k = kk + 1;
        }
        int64_t idx = i + t - k;
        if (idx < 0) idx += L;
        total += vec[idx];
        if (t - k - 1 < 0) break;
        bnd[t - k - 1] = 1;
        t = t - k;
    }
    totals[p] = total;
    if (n_draws) n_draws[p] = nd;
    if (status) status[p] = (kind == 2 && total == NEG_INF_D) ? 1 : 0;
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

static int check_corpus(const segk_corpus *c)
{
    SEGK_REQUIRE(c != nullptr, "corpus is NULL");
    SEGK_REQUIRE(c->x_dtype == SEGK_F32 || c->x_dtype == SEGK_F64, "x_dtype");
    SEGK_REQUIRE(c->D > 0 && c->n_emb > 0, "empty corpus");
    SEGK_REQUIRE(c->ld32 % 4 == 0 && c->ld32 >= c->D, "ld32 must be D rounded up to a multiple of 4");
    return SEGK_OK;
}

// Launch plan of the filter stage.  Let slots = resident workgroups of the chip and chunks =
// ceil(n / rows per workgroup).  The first floor(chunks / slots) * slots chunks go to the plain
// kernel (whole rounds); the remaining r < slots chunks -- a last round that would leave most of
// the chip idle while a few workgroups walk all component tiles, or the whole launch when n is
// small (a shard of a multi-GPU run, one utterance of the serial chain) -- are scored split-K:
// each chunk by `s` workgroups over disjoint tile ranges, merged by k_score_merge; fewer than
// SEGK_TAIL_QUEUE left-over rows simply join the ambiguity queue (full scan).
template <int GMAX, int NB, int WAVES>
static int launch_score(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, ScoreArgs A, hipStream_t st)
{
    const size_t lds = 2 * (size_t)A.tile_stride * sizeof(float);
    const int rows_per_wg = WAVES * 32 * NB;
    static int wg_per_cu = 0;
    if (!wg_per_cu) {
        if (lds > 48 * 1024) {
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score<GMAX, NB, WAVES, 0>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score<GMAX, NB, WAVES, 1>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        int occ = 0;
        SEGK_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_kmeans_score<GMAX, NB, WAVES, 0>,
                                                                    64 * WAVES, lds));
        wg_per_cu = occ > 0 ? occ : 1;
    }
    const int64_t slots = (int64_t)wg_per_cu * ctx->n_cu;
    const int64_t chunks = (A.n + rows_per_wg - 1) / rows_per_wg;
    int64_t main_chunks = (chunks / slots) * slots, tail_chunks = chunks - main_chunks;
    int n_split = 1;
    if (tail_chunks > 0) {
        n_split = (int)(slots / tail_chunks);
        if (n_split > A.n_tiles) n_split = A.n_tiles;
        if ((int64_t)n_split * tail_chunks * rows_per_wg > SEGK_WS_ENTRIES) n_split = (int)(SEGK_WS_ENTRIES / (tail_chunks * rows_per_wg));
    }
    const char *no_split = getenv("SEGK_SCORE_NOSPLIT");
    if (n_split < 2 || (no_split && atoi(no_split))) {          // the tail fills at least half a round: no split
        main_chunks = chunks;
        tail_chunks = 0;
    }
    const int64_t n_main = main_chunks * rows_per_wg < A.n ? main_chunks * rows_per_wg : A.n;
    if (main_chunks > 0) {
        ScoreArgs M = A;
        M.n = n_main;
        const bool prof = ctx->prof_on != 0;
        const int slot = ctx->prof_n % SEGK_PROF_SLOTS;
        if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
        hipLaunchKernelGGL((k_kmeans_score<GMAX, NB, WAVES, 0>), dim3((unsigned)main_chunks), dim3(64 * WAVES), lds, st, M);
        if (prof) {
            SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
            ctx->prof_rows[slot] = n_main;
            ctx->prof_kind = 0;
            ctx->prof_n++;
        }
    }
    if (tail_chunks > 0 && A.n - n_main < SEGK_TAIL_QUEUE && A.fuse_exact && !(no_split && atoi(no_split) == 2)) {
        ScoreArgs T = A;
        T.n = A.n - n_main;
        T.row0 = A.row0 + n_main;
        T.ids = A.ids ? A.ids + n_main : nullptr;
        hipLaunchKernelGGL(k_score_queue_rows, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, st, T);
    } else if (tail_chunks > 0) {
        ScoreArgs T = A;
        T.n = A.n - n_main;
        T.row0 = A.row0 + n_main;
        T.ids = A.ids ? A.ids + n_main : nullptr;
        T.n_chunks = (int)tail_chunks;
        T.tiles_per_split = (A.n_tiles + n_split - 1) / n_split;
        const int s_eff = (A.n_tiles + T.tiles_per_split - 1) / T.tiles_per_split;
        T.part_k = ctx->ws_k;
        T.part_f = ctx->ws_f;
        hipLaunchKernelGGL((k_kmeans_score<GMAX, NB, WAVES, 1>), dim3((unsigned)(tail_chunks * s_eff)), dim3(64 * WAVES), lds,
                           st, T);
        hipLaunchKernelGGL(k_score_merge, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, st, T, s_eff);
        if (A.fuse_exact)       // rows scored split-K get their winner's exact score here
            DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_exact_fill<XT>, dim3((unsigned)((T.n + 255) / 256)), dim3(256), 0, st,
                                               *c, *m, T.ids, T.row0, T.n, A.cand););
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static bool segk_use_b3(const segk_corpus *c, const segk_kmeans *m)
{
    const char *e = getenv("SEGK_SCORE_B3");
    if (e && atoi(e) == 0) return false;
    return c->Xb3 && (c->sp_pieces == 2 || c->sp_pieces == 3) && m->tiles_b3 && c->x_dtype == SEGK_F32 && c->D >= 8 &&
           c->D <= 128;
}

// split-precision filter: whole rounds (and any larger remainder) to k_kmeans_score_sp, a remainder of
// fewer than SEGK_TAIL_QUEUE rows to the ambiguity queue.
template <int KS, int P>
static int launch_score_sp(segk_ctx *ctx, ScoreArgs A, hipStream_t st)
{
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    static int wg_per_cu = 0;
    if (!wg_per_cu) {
        if (lds > 48 * 1024)
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score_sp<KS, 4, P>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0;
        SEGK_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_kmeans_score_sp<KS, 4, P>, 256, lds));
        wg_per_cu = occ > 0 ? occ : 1;
    }
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
        const bool prof = ctx->prof_on != 0;
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

// Exact stage of the pre-filter's decided rows: cand.k = (c | SEGK_PAIR_PENDING) names the pair (c, c + 1);
// both members are scored in the reference's float32 arithmetic (sp_exact_score_x) and the larger wins
// (the lower index on a tie, as np.argmax).  A kernel of its own because inside the MFMA kernel these
// reads -- 16 bytes per lane from 64 different rows per instruction, eight waves per CU, one row block
// after the other -- took 60 % of a workgroup's lifetime (s_memtime stamps, profiles/README.md r01_h).
// The arithmetic wants a lane to own whole strided accumulators of one (row, member), the memory system
// wants whole lines: rows and pairs of means (a row is 4 D contiguous bytes, a pair 8 D) are read with
// consecutive lanes on consecutive 16 bytes, written to LDS and scored from there.  One wave per
// workgroup, 16 rows per step, private LDS (no barrier); the loads of step i + 1 are in flight while
// step i is scored.  (Copying by LDS-DMA instead was measured at ~500 cycles per 1 KiB piece with 13
// waves per CU -- the copy engine, not latency, set the pace: 324 us.)  Row stride KS*16 + 8 floats: the
// float4 reads of 8 items x 2 lanes fall on distinct banks.
#define SEGK_PAIR_ROWS 16
template <int KS>
__global__ __launch_bounds__(64) void k_kmeans_exact_pair(ScoreArgs A)
{
    constexpr int R = SEGK_PAIR_ROWS;
    constexpr int C4 = KS * 4;                                     // 16-byte slots read per row (>= D / 4)
    constexpr int LD = KS * 16 + 8;                                // floats per staged row
    constexpr int RPI = 64 / C4;                                   // rows per load instruction (2 for KS 5..8)
    constexpr int NLA = (R + RPI - 1) / RPI;                       // load instructions per array
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [3][R][LD]: x rows, member 0, member 1
    const int lane = threadIdx.x, D = A.D, D4 = D >> 2;
    const int64_t n_steps = (A.n + R - 1) / R;
    const int sub = lane / C4, c4 = lane - sub * C4;               // this lane's row within an instruction, its slot

    // Row ids and pair bases of a step live on lanes 0..R-1.  They are fetched ahead of use and nothing
    // tests them in the iteration that issues the fetch (a test would wait for every older load as well):
    //   rid2 (step i + 2): issued in iteration i;  k1 = cand.k[rid1] (step i + 1): issued in iteration i,
    //   decoded in iteration i + 1 right before that step's loads.
    auto fetch_rid = [&](int64_t step) -> int32_t {
        const int64_t r = step * R + lane;
        int32_t rid = -1;
        if (step < n_steps && lane < R && r < A.n) rid = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
        return rid;
    };
    auto fetch_k = [&](int32_t rid) -> int32_t { return rid >= 0 ? A.cand.k[rid] : 0; };
    float4 v[3 * NLA];
    // rows 2t, 2t + 1 (RPI = 2) of array arr per instruction: consecutive lanes on consecutive 16 bytes
    uint64_t *rowp = reinterpret_cast<uint64_t *>(lds + 3 * R * LD);     // [3][R] row addresses of the step being loaded
    auto issue_loads = [&](int32_t rid, int32_t c, int64_t step_) {
        // 64-bit row addresses once per step, by the rows' own lanes, through LDS: the load instructions
        // below cost a ds_read_b64 and an add each -- no multiply, no lane exchange.  Every load is
        // unconditional (a branch around each of the 24 cost more than the loads): a row that is skipped
        // reads its own float32 row and the first means (never used), a lane without a slot re-reads its
        // neighbour's last 16 bytes (same cache line).
        if (lane < R) {
            const bool live = rid >= 0;
            int64_t r_any = rid;
            if (!live) {                                       // some valid row: this step's own when the rows are a range
                r_any = A.ids ? 0 : A.row0 + step_ * R + lane;
                if (r_any >= A.row0 + A.n || A.ids) r_any = A.ids ? 0 : A.row0;
            }
            const uint64_t xp = (uint64_t)(uintptr_t)(A.xrows32 + r_any * A.ld32);
            const uint64_t mp0 = (uint64_t)(uintptr_t)(A.means32 + (int64_t)(live ? c : 0) * D);
            rowp[lane] = xp;
            rowp[R + lane] = mp0;
            rowp[2 * R + lane] = (live && c + 1 < A.K_max) ? mp0 + (uint64_t)D * 4 : mp0;
        }
        const int sub_c = sub < RPI ? sub : RPI - 1;
        const unsigned off = 16u * (unsigned)(c4 < D4 ? c4 : D4 - 1);
        uint64_t base[3 * NLA];
#pragma unroll
        for (int arr = 0; arr < 3; arr++)
#pragma unroll
            for (int t = 0; t < NLA; t++)
                base[arr * NLA + t] = rowp[arr * R + ((RPI * t + sub_c) & (R - 1))];
#pragma unroll
        for (int i = 0; i < 3 * NLA; i++) {
            // an integer turned pointer is a FLAT pointer to the compiler (flat loads, and the staging array in
            // scratch: 900 us); say that it is global memory
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) f32x4_t *gptr_t;
            const f32x4_t t_ = *reinterpret_cast<gptr_t>((uintptr_t)(base[i] + off));
            v[i] = make_float4(t_.x, t_.y, t_.z, t_.w);
        }
    };
    auto decode = [&](int32_t &rid, int32_t k) -> int32_t {       // pair base, or -1 (and rid = -1) when not pending
        // (a valid mark only: non-negative, pair base inside the component range -- whatever else the caller's
        // candidate buffer holds for a row the pre-filter did not decide is left alone)
        if (rid >= 0 && k >= 0 && (k & SEGK_PAIR_PENDING) && (k & ~SEGK_PAIR_PENDING) < A.K_max) return k & ~SEGK_PAIR_PENDING;
        rid = -1;
        return -1;
    };

    int64_t step = blockIdx.x;
    int32_t rid0 = fetch_rid(step);
    int32_t c0 = decode(rid0, fetch_k(rid0));
    issue_loads(rid0, c0, step);
    int32_t rid1 = fetch_rid(step + gridDim.x);
    int32_t k1 = fetch_k(rid1);
    int32_t rid2 = fetch_rid(step + 2 * (int64_t)gridDim.x);
#ifdef SEGK_STAMP
#define SEGK_STAMP_P(i) do { if (A.stamp && lane == 0 && it_ == 3) A.stamp[65536 + (int64_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEGK_STAMP_P(i) do { } while (0)
#endif
    int it_ = 0;
    for (; step < n_steps; step += gridDim.x, it_++) {
        SEGK_STAMP_P(0);
#pragma unroll
        for (int arr = 0; arr < 3; arr++)
#pragma unroll
            for (int t = 0; t < NLA; t++)
                if (sub < RPI && RPI * t + sub < R)
                    *reinterpret_cast<float4 *>(lds + (arr * R + RPI * t + sub) * LD + 4 * c4) = v[arr * NLA + t];
        SEGK_STAMP_P(1);
        const int32_t ridc = rid0, cc0 = c0;
        const int32_t c1 = decode(rid1, k1);
        issue_loads(rid1, c1, step + gridDim.x);                   // the next step's rows: in flight under this step's arithmetic
        rid0 = rid1; c0 = c1;
        rid1 = rid2;
        k1 = fetch_k(rid1);
        rid2 = fetch_rid(step + 3 * (int64_t)gridDim.x);
        SEGK_STAMP_P(2);
        // lanes 4 r + {0, 1}: member 0 of row r; lanes 4 r + {2, 3}: member 1 (LDS operations of one wave
        // complete in order: the reads below see the writes above)
        const int item = lane >> 1, h = lane & 1, row = item >> 1, mem = item & 1;
        const float sc = sp_exact_score_x<KS, 1>(lds + ((1 + mem) * R + row) * LD, lds + row * LD, D, h);
        const float so = __shfl_xor(sc, 2);
        const int32_t rid = __shfl(ridc, row), c = __shfl(cc0, row);
        if ((lane & 3) == 0 && rid >= 0) {
            const bool second = c + 1 < A.K_max && so > sc;
            A.cand.k[rid] = second ? c + 1 : c;
            A.cand.s[rid] = (double)(second ? so : sc);
        }
        SEGK_STAMP_P(3);
    }
#undef SEGK_STAMP_P
}

// rows the pre-filter launch does not cover (fewer than SEGK_TAIL_QUEUE): straight to its second stage
__global__ void k_pre_queue_rows(ScoreArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n) return;
    const int32_t id = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
    if (id < 0) return;
    const int q = atomicAdd(A.pre_count, 1);
    if (q < A.pre_cap) A.pre_queue[q] = id;
    else {
        const int q2 = atomicAdd(A.cand.count, 1);
        if (q2 < A.amb_cap) A.cand.queue[q2] = id;
    }
}

// One-product pre-filter over all rows, then the split-precision kernel over the rows it queued.
template <int KS>
static int launch_score_pre(segk_ctx *ctx, ScoreArgs A, hipStream_t st)
{
    if (ctx->pre_cap < A.n) {                         // queue of the undecided rows, grown on demand
        if (ctx->pre_queue) SEGK_CHECK_HIP(hipFree(ctx->pre_queue));
        ctx->pre_queue = nullptr;
        ctx->pre_cap = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->pre_queue, sizeof(int32_t) * (size_t)(A.n + 16)));
        ctx->pre_cap = A.n;
        ctx->pre_zeroed = 0;                           // a new buffer: its counter has not been cleared
    }
    A.pre_queue = ctx->pre_queue + 16;
    A.pre_count = ctx->pre_queue;
    // the second stage is launched for every row (its row count is read on the device; the workgroups
    // beyond it leave at once), so the queue cannot overflow whatever the data
    const int64_t cap2 = A.n;
    A.pre_cap = (int)cap2;
#ifdef SEGK_STAMP
    A.stamp = getenv("SEGK_STAMP_PTR") ? (unsigned long long *)strtoull(getenv("SEGK_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
    if (ctx->pre_zeroed) ctx->pre_zeroed = 0;          // segk_kmeans_score cleared it together with the caller's queue length
    else SEGK_CHECK_HIP(hipMemsetAsync(A.pre_count, 0, sizeof(int32_t), st));

    constexpr size_t lds = 2 * (size_t)(KS + 1) * 256 * sizeof(float);
    const int64_t slots = 2 * (int64_t)ctx->n_cu;      // two 4-wave workgroups per CU (launch bounds)
    // whole rounds of 512-row workgroups (four row blocks per wave), the remainder in 256-row workgroups
    const int64_t round4 = slots * 512;
    int64_t n4 = (A.n / round4) * round4;
    if (getenv("SEGK_PRE_NBLK") && atoi(getenv("SEGK_PRE_NBLK")) == 2) n4 = 0;      // development: 256-row workgroups only
    const bool prof = ctx->prof_on != 0;
    const int slot = ctx->prof_n % SEGK_PROF_SLOTS;
    // the timed launch (segk_profile_*): the 512-row-workgroup launch when there is one, else the 256-row one
    auto prof_end = [&](int64_t rows) -> int {
        if (!prof) return SEGK_OK;
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = rows;
        ctx->prof_kind = 1;
        ctx->prof_n++;
        return SEGK_OK;
    };
    // a short remainder is only queued: first, so that nothing small sits between the big launch and the
    // kernels waiting for it
    const int64_t rem = A.n - n4;
    const bool rem_queued = rem > 0 && rem < SEGK_TAIL_QUEUE && n4 > 0;
    ScoreArgs T = A;
    T.n = rem;
    T.row0 = A.row0 + n4;
    T.ids = A.ids ? A.ids + n4 : nullptr;
    if (rem_queued) hipLaunchKernelGGL(k_pre_queue_rows, dim3((unsigned)((rem + 255) / 256)), dim3(256), 0, st, T);
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    if (n4 > 0) {
        ScoreArgs M = A;
        M.n = n4;
        hipLaunchKernelGGL((k_kmeans_score_h1<KS, 4>), dim3((unsigned)(n4 / 512)), dim3(256), lds, st, M);
        if (int rc = prof_end(n4)) return rc;
    }
    if (rem > 0 && !rem_queued) {
        hipLaunchKernelGGL((k_kmeans_score_h1<KS, 2>), dim3((unsigned)((rem + 255) / 256)), dim3(256), lds, st, T);
        if (n4 == 0)
            if (int rc = prof_end(rem)) return rc;
    }
    // The decided rows' exact stage (this stream) and the undecided rows' second stage + full scan (second
    // stream, segk_kmeans_score only) touch disjoint rows: side by side, the exact stage leaving LDS for one
    // second-stage workgroup per CU.  Joined at the end of segk_kmeans_score.
    const bool overlap = ctx->overlap_req != 0;
    hipStream_t st2 = st;
    if (overlap) {
        if (!ctx->aux) {
            SEGK_CHECK_HIP(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
            SEGK_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            SEGK_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        SEGK_CHECK_HIP(hipEventRecord(ctx->ev_fork, st));
        SEGK_CHECK_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        st2 = ctx->aux;
        ctx->aux_busy = 1;
    }
    // exact stage of the decided rows
    {
        const size_t lds_p = 3 * (size_t)SEGK_PAIR_ROWS * (KS * 16 + 8) * sizeof(float) + 3 * SEGK_PAIR_ROWS * sizeof(uint64_t);
        const int64_t steps = (A.n + SEGK_PAIR_ROWS - 1) / SEGK_PAIR_ROWS;
        int64_t waves = (int64_t)ctx->n_cu * (int64_t)(((overlap ? 124 : 160) * 1024) / lds_p);
        if (waves > 8 * (int64_t)ctx->n_cu) waves = 8 * (int64_t)ctx->n_cu;
        if (waves > steps) waves = steps;
        hipLaunchKernelGGL((k_kmeans_exact_pair<KS>), dim3((unsigned)waves), dim3(64), lds_p, st, A);
    }
    // second stage: all three products for the queued rows; the row count is read on the device
    {
        constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
        const size_t lds2 = 2 * (size_t)STRIDE * sizeof(float);
        static bool attr_set = false;
        if (!attr_set && lds2 > 48 * 1024) {
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score_sp<KS, 4, 2>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            attr_set = true;
        }
        ScoreArgs B = A;
        B.ids = A.pre_queue;
        B.row0 = 0;
        B.n = cap2;
        B.n_dev = A.pre_count;
        hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2>), dim3((unsigned)((cap2 + 127) / 128)), dim3(256), lds2, st2, B);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int dispatch_score_pre(segk_ctx *ctx, const ScoreArgs &A, int ks, hipStream_t st)
{
    switch (ks) {
        case 1: return launch_score_pre<1>(ctx, A, st);
        case 2: return launch_score_pre<2>(ctx, A, st);
        case 3: return launch_score_pre<3>(ctx, A, st);
        case 4: return launch_score_pre<4>(ctx, A, st);
        case 5: return launch_score_pre<5>(ctx, A, st);
        case 6: return launch_score_pre<6>(ctx, A, st);
        case 7: return launch_score_pre<7>(ctx, A, st);
        case 8: return launch_score_pre<8>(ctx, A, st);
        default: break;
    }
    segk_set_error("pre-filter: D out of range");
    return SEGK_ERR_UNSUPPORTED;
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

// ---- split-precision images of arbitrary float32 matrices (internal; used by segk_fbbatch.hip) ----
int segk_sp_prepare_rows(const float *Y, int64_t ldy, int64_t n, int D2, void *img, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(img, 0, SEGK_SP_HEADER, st));
    const int64_t nx = n * D2, tot = n * segk_b3_kp(D2);
    const int64_t blocks = (nx + 255) / 256;
    hipLaunchKernelGGL(k_corpus_maxabs, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, Y, ldy, n, D2,
                       (unsigned int *)img);
    hipLaunchKernelGGL(k_corpus_split_sp<2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Y, ldy, n, D2,
                       (unsigned char *)img);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int segk_sp_prepare_tiles(const float *rows, const double *consts, const double *rowmax2, int K, int D2, float *tiles_sp,
                          const void *ximg, void *stream)
{
    hipLaunchKernelGGL(k_kmeans_prepare_sp<2>, dim3(segk_n_tiles(K)), dim3(256), 0, (hipStream_t)stream, rows, K, D2, tiles_sp,
                       rowmax2, (const unsigned char *)ximg, consts);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

template <int KS>
static int launch_score_lse_sp(segk_ctx *ctx, const ScoreArgs &A, hipStream_t st)
{
    constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score_sp<KS, 4, 2, 1>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int64_t chunks = (A.n + 127) / 128;
    const bool prof = ctx && ctx->prof_on != 0;
    const int slot = prof ? ctx->prof_n % SEGK_PROF_SLOTS : 0;
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2, 1>), dim3((unsigned)chunks), dim3(256), lds, st, A);
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
static int launch_score_mat_sp(const ScoreArgs &A, hipStream_t st)
{
    constexpr int STRIDE = (KS * 2 * 256 + 32 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)STRIDE * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score_sp<KS, 4, 2, 2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_kmeans_score_sp<KS, 4, 2, 2>), dim3((unsigned)((A.n + 127) / 128)), dim3(256), lds, st, A);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// mat[r][k] = acc_k of row ids[r] (r < n), k < 32 n_tiles: the contraction itself, for callers that need
// every component's value (the token likelihoods of the batch sampler's assignment step)
int segk_launch_score_mat_sp(const void *ximg, int D2, const int32_t *ids, int64_t n, const float *tiles_sp, int n_tiles,
                             float *mat, int64_t mat_ld, void *stream)
{
    if (n <= 0) return SEGK_OK;
    ScoreArgs A{};
    memset(&A, 0, sizeof(A));
    A.X32 = (const float *)ximg; A.ids = ids; A.row0 = 0; A.n = n;
    A.tiles = tiles_sp; A.n_tiles = n_tiles; A.tile_stride = segk_sp_tile_stride(D2, 2);
    A.D = D2;
    A.mat_out = mat; A.mat_ld = mat_ld;
    hipStream_t st = (hipStream_t)stream;
    switch (segk_b3_kp(D2) / 16) {
#define SEGK_CASE(k) \
    case k: return launch_score_mat_sp<k>(A, st);
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
                             const float *tiles_sp, int n_tiles, double norm, double *out, void *stream)
{
    if (n <= 0) return SEGK_OK;
    ScoreArgs A{};
    memset(&A, 0, sizeof(A));
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

template <int GMAX>
static int launch_score_lse(segk_ctx *ctx, const ScoreArgs &A, hipStream_t st)
{
    const size_t lds = 2 * (size_t)A.tile_stride * sizeof(float);
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_score<GMAX, 1, 4, 0, 1>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int64_t chunks = (A.n + 127) / 128;
    const bool prof = ctx && ctx->prof_on != 0;
    const int slot = prof ? ctx->prof_n % SEGK_PROF_SLOTS : 0;
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    hipLaunchKernelGGL((k_kmeans_score<GMAX, 1, 4, 0, 1>), dim3((unsigned)chunks), dim3(256), lds, st, A);
    if (prof) {
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = A.n;
        ctx->prof_kind = 4;
        ctx->prof_n++;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// out[row] = ln sum_k exp(z_k) - norm for the rows ids[r] (ids == NULL: row0 + r), r < n, where z_k * log2(e) = tile constant of component k +
// <Y[row], tile row k> with Y [n_emb, ldy] float32 rows of D2 dimensions (segk_fbbatch.hip).  Internal, not ABI.
int segk_launch_score_lse(segk_ctx *ctx, const float *Y, int64_t ldy, int D2, const int32_t *ids, int64_t row0, int64_t n,
                          const float *tiles, int n_tiles, double norm, double *out, void *stream)
{
    if (n <= 0) return SEGK_OK;
    ScoreArgs A{};
    memset(&A, 0, sizeof(A));
    A.X32 = Y; A.ld32 = ldy; A.ids = ids; A.row0 = row0; A.n = n;
    A.tiles = tiles; A.n_tiles = n_tiles; A.tile_stride = segk_tile_stride(D2);
    A.G = segk_G(D2); A.D = D2;
    A.lse_out = out; A.lse_norm = norm;
    hipStream_t st = (hipStream_t)stream;
    switch (segk_gmax(D2)) {
#define SEGK_CASE(g) \
    case g: return launch_score_lse<g>(ctx, A, st);
        SEGK_CASE(1) SEGK_CASE(2) SEGK_CASE(4) SEGK_CASE(6) SEGK_CASE(8) SEGK_CASE(10) SEGK_CASE(13) SEGK_CASE(16)
        SEGK_CASE(20) SEGK_CASE(25) SEGK_CASE(26) SEGK_CASE(28) SEGK_CASE(32) SEGK_CASE(33) SEGK_CASE(34) SEGK_CASE(40)
        SEGK_CASE(50) SEGK_CASE(64) SEGK_CASE(75) SEGK_CASE(100)
#undef SEGK_CASE
        default: break;
    }
    segk_set_error("segk_launch_score_lse: 2D=%d > 400 is not supported by the register-resident score kernel", D2);
    return SEGK_ERR_UNSUPPORTED;
}

// Exact duplicates among the rows of `means` (clean_components leaves the moved component's old row behind,
// inactive rows hold copies): a duplicate with the HIGHER index can never be np.argmax -- its score is the
// lower one's bit for bit and the first maximum wins -- but it makes every row near the pair a tie that only
// the full scan resolves (2 100 of the 2 200 queued rows of a 1 250-utterance shard had exactly these two
// contenders).  One workgroup: value hashes of all rows (8 lanes per row), then every row looks for an
// earlier row with its hash, verifies equality element by element, and if it finds one writes the
// "absent" constant (-3e38, what the padding components carry) over its accumulator seed in both tile
// images.  The full scan does not read those constants, so its first-maximum rule is untouched.
template <typename XT>
__global__ __launch_bounds__(1024) void k_kmeans_mark_dups(const XT *means, int K_max, int D, float *tiles, int stride32, int G,
                                                           float *tiles_sp, int stride_sp, int sp_const_off, int32_t *n_marked,
                                                           const unsigned long long *row_hash)
{
    // open-addressing table in LDS: key = row hash, value = the lowest row index with that hash
    constexpr int TB = 4096;                                      // slots (K_max <= 2048: load factor <= 1/2; 48 KiB)
    __shared__ unsigned long long keys[TB];
    __shared__ int32_t first[TB];
    const int tid = threadIdx.x, sub = tid & 7;
    for (int i = tid; i < TB; i += blockDim.x) { keys[i] = 0ull; first[i] = 0x7fffffff; }
    __syncthreads();
    for (int k = tid; k < K_max; k += blockDim.x) {
        const unsigned long long h = row_hash[k];
        for (unsigned slot = (unsigned)(h >> 20) & (TB - 1);; slot = (slot + 1) & (TB - 1)) {
            const unsigned long long prev = atomicCAS(&keys[slot], 0ull, h);
            if (prev == 0ull || prev == h) { atomicMin(&first[slot], k); break; }
        }
    }
    __syncthreads();
    // every row: the first row with its hash; if that is an earlier one, verify element by element (8 lanes per
    // row, all loads of a lane in flight together) and mark
    int marked = 0;
    for (int k0 = 0; k0 < K_max; k0 += 128) {
        const int k = k0 + (tid >> 3);
        int i = -1;
        if (k < K_max) {
            const unsigned long long h = row_hash[k];
            unsigned slot = (unsigned)(h >> 20) & (TB - 1);
            while (keys[slot] != h) slot = (slot + 1) & (TB - 1);
            i = first[slot];
            if (i >= k) i = -1;
        }
        int eq = i >= 0;
        if (i >= 0) {
#pragma unroll 4
            for (int d = sub; d < D; d += 8) eq &= means[(int64_t)i * D + d] == means[(int64_t)k * D + d];
        }
        eq &= __shfl_xor(eq, 1);
        eq &= __shfl_xor(eq, 2);
        eq &= __shfl_xor(eq, 4);
        if (sub == 0 && i >= 0 && eq) {
            tiles[(int64_t)(k >> 5) * stride32 + G * 128 + (k & 31)] = -3.0e38f;
            if (tiles_sp) tiles_sp[1024 + (int64_t)(k >> 5) * stride_sp + sp_const_off + (k & 31)] = -3.0e38f;
            marked++;
        }
    }
    if (n_marked && marked) atomicAdd(n_marked, marked);
}

__global__ void k_zero_two(int32_t *a, int32_t *b)
{
    *a = 0;
    *b = 0;
}

extern "C" {

int64_t segk_kmeans_tiles_floats(int32_t K_max, int32_t D)
{
    return (int64_t)segk_n_tiles(K_max) * segk_tile_stride(D);
}

int32_t segk_corpus_prepare(segk_ctx *ctx, const segk_corpus *c, float *X32_out, float *xnorm_out, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(xnorm_out != nullptr, "xnorm_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    int64_t nblk = (c->n_emb + 3) / 4;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_corpus_prepare<XT>, dim3((unsigned)nblk), dim3(256), 0, st,
                                       (const XT *)c->X, c->ldx, c->n_emb, c->D, c->ld32, X32_out, xnorm_out););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int kmeans_prepare_impl(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream, bool mnorm_zeroed);

int32_t segk_kmeans_prepare(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream)
{
    return kmeans_prepare_impl(ctx, c, m, stream, false);
}

static int kmeans_prepare_impl(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream, bool mnorm_zeroed)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && m->mnorm_max, "kmeans tiles/mnorm_max");
    hipStream_t st = (hipStream_t)stream;
    // value hashes of the rows, for segk_kmeans_mark_duplicates (context-owned, K_max <= 2048 only)
    unsigned long long *row_hash = nullptr;
    if (ctx && m->K_max <= 2048) {
        if (!ctx->row_hash) SEGK_CHECK_HIP(hipMalloc((void **)&ctx->row_hash, 2048 * sizeof(unsigned long long)));
        row_hash = ctx->row_hash;
        ctx->row_hash_means = m->means;
    }
    // mnorm_max = max_k |m_k|^2, maintained by atomicMax on the bit pattern (non-negative doubles)
    if (!mnorm_zeroed) SEGK_CHECK_HIP(hipMemsetAsync(m->mnorm_max, 0, sizeof(double), st));
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_prepare<XT>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st,
                                       (const XT *)m->means, m->K_max, c->D, m->tiles,
                                       (unsigned long long *)m->mnorm_max, m->tiles_b3 ? (unsigned int *)m->tiles_b3 + 1 : nullptr,
                                       row_hash););
    if (m->tiles_b3 && c->Xb3 && c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128) {
        if (c->sp_pieces == 2)
            hipLaunchKernelGGL(k_kmeans_prepare_sp<2>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st, (const float *)m->means,
                               m->K_max, c->D, m->tiles_b3, m->mnorm_max, (const unsigned char *)c->Xb3, (const double *)nullptr);
        else if (c->sp_pieces == 3)
            hipLaunchKernelGGL(k_kmeans_prepare_sp<3>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st, (const float *)m->means,
                               m->K_max, c->D, m->tiles_b3, m->mnorm_max, (const unsigned char *)c->Xb3, (const double *)nullptr);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_mark_duplicates(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int32_t *n_marked, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && m->means, "kmeans tiles / means");
    // needs the row hashes of the segk_kmeans_prepare that built these images (same context, same means buffer)
    if (!ctx || m->K_max > 2048 || !ctx->row_hash || ctx->row_hash_means != m->means) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool sp = m->tiles_b3 && c->Xb3 && c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128 && (c->sp_pieces == 2 || c->sp_pieces == 3);
    const int kp = segk_b3_kp(c->D);
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_mark_dups<XT>, dim3(1), dim3(1024), 0, st,
                                       (const XT *)m->means, m->K_max, c->D, m->tiles, segk_tile_stride(c->D), segk_gmax(c->D),
                                       sp ? m->tiles_b3 : (float *)nullptr, sp ? segk_sp_tile_stride(c->D, c->sp_pieces) : 0,
                                       sp ? (kp / 16) * c->sp_pieces * 256 : 0, n_marked, ctx->row_hash););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int64_t segk_kmeans_tiles_b3_floats(int32_t K_max, int32_t D)
{
    return 1024 + (int64_t)segk_n_tiles(K_max) * segk_sp_tile_stride(D, 3);     // sized for either piece count
}

int64_t segk_corpus_b3_bytes(int64_t n_emb, int32_t D)
{
    return SEGK_SP_HEADER + n_emb * 3 * (int64_t)segk_b3_kp(D) * 2;             // sized for either piece count
}

int32_t segk_corpus_prepare_b3(segk_ctx *ctx, const segk_corpus *c, void *Xb3_out, int32_t pieces, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(Xb3_out != nullptr, "Xb3_out is NULL");
    SEGK_REQUIRE(pieces == 2 || pieces == 3, "pieces must be 2 (fp16x2) or 3 (bf16x3)");
    SEGK_REQUIRE(c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128, "the split images exist for float32 data with 8 <= D <= 128");
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(Xb3_out, 0, SEGK_SP_HEADER, st));
    const int64_t tot = c->n_emb * segk_b3_kp(c->D);
    if (pieces == 2) {
        const int64_t nx = c->n_emb * c->D;
        const int64_t blocks = (nx + 255) / 256;
        hipLaunchKernelGGL(k_corpus_maxabs, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned int *)Xb3_out);
        hipLaunchKernelGGL(k_corpus_split_sp<2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
        hipLaunchKernelGGL(k_corpus_resid_sp, dim3((unsigned)((c->n_emb + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
    } else {
        hipLaunchKernelGGL(k_corpus_split_sp<3>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int score_checks(const segk_corpus *c, const segk_kmeans *m, const int32_t *ids, int64_t row0, int64_t n,
                        const segk_cand *cand)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && cand && cand->k && cand->f && cand->s && cand->queue && cand->count && c->X32,
                 "score operands");
    SEGK_REQUIRE(ids != nullptr || (row0 >= 0 && row0 + n <= c->n_emb), "row range");
    return SEGK_OK;
}

int32_t segk_kmeans_clear_queue(segk_ctx *ctx, const segk_cand *cand, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(cand && cand->count, "cand");
    SEGK_CHECK_HIP(hipMemsetAsync(cand->count, 0, sizeof(int32_t), (hipStream_t)stream));
    return SEGK_OK;
}

int32_t segk_kmeans_filter(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                           int64_t row0, int64_t n, const segk_cand *cand, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = score_checks(c, m, ids, row0, n, cand);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    ScoreArgs A{};
    A.X32 = c->X32; A.ld32 = c->ld32; A.ids = ids; A.row0 = row0; A.n = n;
    A.tiles = m->tiles; A.n_tiles = segk_n_tiles(m->K_max); A.tile_stride = segk_tile_stride(c->D);
    A.G = segk_G(c->D); A.D = c->D;
    A.fuse_exact = (c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128) ? 1 : 0;
    A.is_f64 = c->x_dtype == SEGK_F64;
    A.dbg = getenv("SEGK_SCORE_DBG") ? atoi(getenv("SEGK_SCORE_DBG")) : 0;
    A.xnorm = c->xnorm; A.mnorm2 = m->mnorm_max; A.cand = *cand; A.amb_cap = (int)c->n_emb;
    A.n_chunks = 0; A.tiles_per_split = 0; A.part_k = nullptr; A.part_f = nullptr;
    // 4-wave workgroups, two per CU: the two waves sharing a SIMD belong to DIFFERENT workgroups
    // and drift apart, covering each other's barrier/staging gaps.  (Measured: an 8-wave
    // workgroup, one barrier for all eight waves, locks the SIMD partners in step and is 15 %
    // slower although it halves the staging instructions per wave.)
    if (segk_use_b3(c, m)) {
        A.xrows32 = c->X32;
        A.X32 = (const float *)c->Xb3;
        A.tiles = m->tiles_b3;
        A.tile_stride = segk_sp_tile_stride(c->D, c->sp_pieces);
        A.means32 = (const float *)m->means;
        A.fuse_exact = (c->D % 4 == 0) ? 1 : 0;
        A.K_max = m->K_max;
        A.xerr = (const float *)((const unsigned char *)c->Xb3 + SEGK_SP_HEADER + c->n_emb * 2 * (int64_t)segk_b3_kp(c->D) * 2);
        // one-product pre-filter in front (two-piece images, D % 4 == 0).  Its three extra launches -- and the
        // second stage's fixed cost, one workgroup's pass over every tile with all three products (~45 us)
        // -- pay once the split-precision kernel alone would need more than four rounds of the chip
        // (estimated break-even near 130 k rows; 1 M rows: 0.58 ms against 0.77 ms).
        // SEGK_SCORE_PRE=0 disables it, =1 forces it at every size (tests).
        const char *pre_env = getenv("SEGK_SCORE_PRE");
        const int pre_mode = pre_env ? atoi(pre_env) : -1;
        if (c->sp_pieces == 2 && A.fuse_exact && pre_mode != 0 && (pre_mode == 1 || n > 1024 * (int64_t)ctx->n_cu) &&
            n < (int64_t)1 << 30)
            return dispatch_score_pre(ctx, A, segk_b3_kp(c->D) / 16, st);
        return c->sp_pieces == 2 ? dispatch_score_sp<2>(ctx, A, segk_b3_kp(c->D) / 16, st)
                                 : dispatch_score_sp<3>(ctx, A, segk_b3_kp(c->D) / 16, st);
    }
    // Rows per wave: one 32-row MFMA column block per wave (108 VGPRs, four workgroups per CU) beat
    // two blocks sharing every LDS tile fetch (194 VGPRs, two per CU) at every row count measured
    // (D = 100: 77 % vs 73 % of the fp32 matrix peak); SEGK_SCORE_NB=2 selects the latter.
    const char *nb_env = getenv("SEGK_SCORE_NB");
    const bool one_block = nb_env ? atoi(nb_env) == 1 : true;   // measured faster at every size for D = 100
    switch (segk_gmax(c->D)) {
#define SEGK_CASE(g, nb) \
    case g: return (nb == 2 && !one_block) ? launch_score<g, nb, 4>(ctx, c, m, A, st) : launch_score<g, 1, 4>(ctx, c, m, A, st);
        SEGK_CASE(1, 2) SEGK_CASE(2, 2) SEGK_CASE(4, 2) SEGK_CASE(6, 2) SEGK_CASE(8, 2) SEGK_CASE(10, 2)
        SEGK_CASE(13, 2) SEGK_CASE(16, 2) SEGK_CASE(20, 2) SEGK_CASE(25, 2) SEGK_CASE(26, 2) SEGK_CASE(28, 2)
        SEGK_CASE(32, 2) SEGK_CASE(33, 2) SEGK_CASE(34, 2) SEGK_CASE(40, 1) SEGK_CASE(50, 1) SEGK_CASE(64, 1)
        SEGK_CASE(75, 1) SEGK_CASE(100, 1)
#undef SEGK_CASE
        default: break;
    }
    segk_set_error("segk_kmeans_filter: D=%d > 400 is not supported by the register-resident score kernel", c->D);
    return SEGK_ERR_UNSUPPORTED;
}

static int resolve_on(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                      int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream);

int32_t segk_kmeans_resolve(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                            int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream)
{
    return resolve_on(ctx, c, m, ids, row0, n, cand, status, stream);
}

static int resolve_on(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                      int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = score_checks(c, m, ids, row0, n, cand);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool fused = (c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128);
    if (!fused || (segk_use_b3(c, m) && c->D % 4 != 0))       // the split-precision epilogue is fused for D % 4 == 0
        DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_exact_fill<XT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                           *c, *m, ids, row0, n, *cand););
    const int nt = 256;
    if (fused) {         // float32 data, 8 <= D <= 128: SEGK_BR queued rows per workgroup, components in slices
        const size_t lds = (size_t)SEGK_BR * ((c->D + 3) & ~3) * sizeof(float) + nt * (sizeof(float) + sizeof(int32_t));
        // grid for the worst case the host can see (every row queued), capped; the kernel reads the queue
        // length and slices the components over whatever the grid leaves per row group (a queue of 200 rows
        // on 1024 workgroups: four slices instead of 25 busy workgroups; alone: 15 us + 13 us per 1000 rows)
        const int64_t groups = (n + SEGK_BR - 1) / SEGK_BR;
        int max_split = (m->K_max + nt - 1) / nt;
        if (max_split > 8) max_split = 8;
        if (max_split < 1) max_split = 1;
        int64_t grid = groups * max_split;
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(k_kmeans_brute_rows, dim3((unsigned)grid), dim3(nt), lds, st, *c, *m, *cand,
                           (int)c->n_emb, status ? status + 1 : nullptr, (int)grid, max_split, ctx->ws_u64, SEGK_WS_ENTRIES);
        if (max_split > 1)
            hipLaunchKernelGGL(k_brute_finish, dim3(32), dim3(256), 0, st, *cand, (int)c->n_emb, ctx->ws_u64, SEGK_WS_ENTRIES,
                               (int)grid, max_split);
    } else {
        size_t xsz = (c->x_dtype == SEGK_F32 ? 4 : 8) * (size_t)((c->D + 1) & ~1);
        size_t lds = nt * sizeof(double) + xsz + nt * sizeof(int32_t);
        int64_t grid = n < 1024 ? n : 1024;
        DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_brute<XT>, dim3((unsigned)grid), dim3(nt), lds, st, *c, *m, *cand,
                                           (int)c->n_emb, status ? status + 1 : nullptr););
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_score(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                          int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream)
{
    int rc;
    if (ctx && ctx->pre_queue && cand && cand->count) {
        // one tiny kernel instead of two 4-byte memsets: the caller's queue length and the pre-filter's
        hipLaunchKernelGGL(k_zero_two, dim3(1), dim3(1), 0, (hipStream_t)stream, cand->count, ctx->pre_queue);
        ctx->pre_zeroed = 1;
        rc = SEGK_OK;
    } else {
        rc = segk_kmeans_clear_queue(ctx, cand, stream);
    }
    if (rc) return rc;
    // SEGK_SCORE_OVERLAP=0: everything on the caller's stream
    const char *ov = getenv("SEGK_SCORE_OVERLAP");
    ctx->overlap_req = (ov && atoi(ov) == 0) ? 0 : 1;
    rc = segk_kmeans_filter(ctx, c, m, ids, row0, n, cand, stream);
    ctx->overlap_req = 0;
    ctx->pre_zeroed = 0;
    if (rc) return rc;
    if (!ctx->aux_busy) return resolve_on(ctx, c, m, ids, row0, n, cand, status, stream);
    rc = resolve_on(ctx, c, m, ids, row0, n, cand, status, (void *)ctx->aux);
    ctx->aux_busy = 0;
    SEGK_CHECK_HIP(hipEventRecord(ctx->ev_join, ctx->aux));
    SEGK_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ctx->ev_join, 0));
    return rc;
}

int32_t segk_kmeans_exact_max(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                              int64_t n, const segk_cand *cand, double *out_max, int32_t *out_arg, void *stream)
{
    (void)ctx;
    (void)m;
    int rc = check_corpus(c);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    hipLaunchKernelGGL(k_kmeans_gather_cand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       *cand, ids, n, out_max, out_arg);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_neg_sqrd_norm(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int64_t row,
                                  void *out, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(row >= 0 && row < c->n_emb, "row out of range");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_neg_sqrd_norm<XT>, dim3((m->K_max + 255) / 256), dim3(256), 0, st,
                                       *c, *m, row, (XT *)out););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_segment(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *utts,
                            int32_t utt0, int32_t n_utts, int32_t n_slices_min, int32_t n_slices_max, double wip,
                            const segk_cand *cand, uint8_t *boundaries, int32_t *old_tok, int32_t *new_tok,
                            int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag, double *out_total,
                            int32_t *status, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1,
                 "n_slices_min must be 0 or 1 (>= 2 crashes in the reference, SURVEY 8(c))");
    SEGK_REQUIRE(n_slices_max >= 0, "n_slices_max");
    SEGK_REQUIRE(utts != nullptr || (utt0 >= 0 && utt0 + n_utts <= c->n_utt), "utterance range");
    if (n_utts <= 0) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    const int W = (n_slices_max > 0 && n_slices_max < c->N_max) ? n_slices_max : c->N_max;
    const int band_cap = c->N_max * W;
    size_t wave_bytes = (size_t)(band_cap + c->N_max + 1) * sizeof(double)
                        + (size_t)(2 * band_cap + 3 * c->N_max + 4) * sizeof(int32_t) + (size_t)c->N_max;
    wave_bytes = (wave_bytes + 15) & ~(size_t)15;
    int waves = 4;
    while (waves > 1 && waves * wave_bytes > 64 * 1024) waves >>= 1;
    size_t lds = waves * wave_bytes;
    if (lds > 160 * 1024) {
        segk_set_error("segk_kmeans_segment: band of %d x %d spans needs %zu B of LDS (> 160 KiB); "
                       "set n_slices_max", c->N_max, W, lds);
        return SEGK_ERR_UNSUPPORTED;
    }
    if (lds > 48 * 1024)
        SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_kmeans_segment, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds));
    const bool w8_ok = n_slices_max >= 1 && n_slices_max <= 8 && c->N_max <= 64 && !(getenv("SEGK_SEGMENT_GENERIC") && atoi(getenv("SEGK_SEGMENT_GENERIC")));
    // two utterances per wave once one-per-wave would not fit the chip in a single round (8 waves per SIMD);
    // SEGK_SEGMENT_X2=0 / 1: never / always
    const char *x2e = getenv("SEGK_SEGMENT_X2");
    const int n_cu_ = ctx ? ctx->n_cu : 256;
    if (w8_ok && c->N_max <= 32 && 2 * waves * wave_bytes <= 48 * 1024 && (x2e ? atoi(x2e) != 0 : n_utts > 28 * n_cu_)) {
        const int per_block = 2 * waves;
        hipLaunchKernelGGL(k_kmeans_segment_w8x2, dim3((n_utts + per_block - 1) / per_block), dim3(64 * waves), 2 * lds, st, *c, *m, utts,
                           utt0, n_utts, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old, n_new, n_flag,
                           out_total, status, band_cap, (int)wave_bytes);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    if (n_slices_max >= 1 && n_slices_max <= 8 && c->N_max <= 64 && !(getenv("SEGK_SEGMENT_GENERIC") && atoi(getenv("SEGK_SEGMENT_GENERIC")))) {
        hipLaunchKernelGGL(k_kmeans_segment_w8, dim3((n_utts + waves - 1) / waves), dim3(64 * waves), lds, st, *c, *m, utts, utt0,
                           n_utts, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old, n_new, n_flag,
                           out_total, status, band_cap, (int)wave_bytes);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    hipLaunchKernelGGL(k_kmeans_segment, dim3((n_utts + waves - 1) / waves), dim3(64 * waves), lds, st, *c, *m, utts,
                       utt0, n_utts, n_slices_min, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old,
                       n_new, n_flag, out_total, status, band_cap, (int)wave_bytes);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

static int launch_update(const segk_corpus *c, segk_kmeans *m, int op, int utt, int64_t item, int k_item,
                         const int32_t *old_tok, const int32_t *new_tok, const int32_t *new_k,
                         const int32_t *n_old, const int32_t *n_new, int32_t *status, hipStream_t st)
{
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_update<XT>, dim3(1), dim3(256), 0, st, *c, *m, op, utt, item, k_item,
                                       old_tok, new_tok, new_k, n_old, n_new, status););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_update_utt(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t utt,
                               const int32_t *old_tok, const int32_t *new_tok, const int32_t *new_k,
                               const int32_t *n_old, const int32_t *n_new, int32_t *status, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(utt >= 0 && utt < c->n_utt, "utt out of range");
    rc = launch_update(c, m, 0, utt, 0, 0, old_tok, new_tok, new_k, n_old, n_new, status, (hipStream_t)stream);
    if (rc) return rc;
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_kmeans_add_item(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int64_t i, int32_t k,
                             int32_t *status, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(i >= 0 && i < c->n_emb, "item out of range");
    SEGK_REQUIRE(k >= 0, "k");
    rc = launch_update(c, m, 1, 0, i, k, nullptr, nullptr, nullptr, nullptr, nullptr, status, (hipStream_t)stream);
    if (rc) return rc;
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_kmeans_del_item(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int64_t i, int32_t *status,
                             void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(i >= 0 && i < c->n_emb, "item out of range");
    rc = launch_update(c, m, 2, 0, i, 0, nullptr, nullptr, nullptr, nullptr, nullptr, status, (hipStream_t)stream);
    if (rc) return rc;
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_kmeans_clean_components(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t *status,
                                     void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    rc = launch_update(c, m, 3, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, status, (hipStream_t)stream);
    if (rc) return rc;
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_kmeans_del_component(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t k,
                                  int32_t *status, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(k >= 0 && k < m->K_max, "k out of range");
    rc = launch_update(c, m, 4, 0, 0, k, nullptr, nullptr, nullptr, nullptr, nullptr, status, (hipStream_t)stream);
    if (rc) return rc;
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_kmeans_batch_collect(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t utt_lo,
                                  int32_t utt_hi, const int32_t *new_k, const int32_t *n_new, const int32_t *n_flag,
                                  int32_t *tok_off, int32_t *flag_buf, int32_t cap, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(0 <= utt_lo && utt_lo <= utt_hi && utt_hi <= c->n_utt, "utterance range");
    hipLaunchKernelGGL(k_batch_collect, dim3(1), dim3(1024), 0, (hipStream_t)stream, *c, *m, utt_lo, utt_hi, new_k,
                       n_new, n_flag, tok_off, flag_buf, cap);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_batch_assign(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t utt_lo,
                                 int32_t utt_hi, const int32_t *flag_all, int32_t n_ranks, int32_t my_rank,
                                 int32_t cap, const int32_t *new_tok, int32_t *new_k, const int32_t *n_new,
                                 const int32_t *tok_off, int32_t *ctok_id, int32_t *ctok_k, int32_t *status,
                                 void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_batch_resolve, dim3(1), dim3(64), 0, st, *m, flag_all, n_ranks, my_rank, cap, new_k, status);
    int64_t tot = (int64_t)(utt_hi - utt_lo) * c->N_max;
    if (tot > 0)
        hipLaunchKernelGGL(k_batch_compact, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, *c, utt_lo, utt_hi,
                           new_tok, new_k, n_new, tok_off, ctok_id, ctok_k);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_batch_partials(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                   const int32_t *blk_lo, int32_t n_blocks_local, int32_t utt_lo,
                                   const int32_t *tok_off, const int32_t *ctok_id, const int32_t *ctok_k,
                                   const double *out_total, double *part_sum, int64_t *part_cnt,
                                   double *part_tot, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    if (n_blocks_local <= 0) return SEGK_OK;
    int64_t grid = (int64_t)n_blocks_local * ((m->K_max + 7) / 8);
    DISPATCH_XT(c, hipLaunchKernelGGL(k_batch_partials<XT>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream,
                                       *c, *m, blk_lo, n_blocks_local, utt_lo, tok_off, ctok_id, ctok_k, out_total,
                                       part_sum, part_cnt, part_tot,
                                       getenv("SEGK_PART_DBG") ? atoi(getenv("SEGK_PART_DBG")) : 0););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_batch_finalize(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t utt_lo,
                                   int32_t utt_hi, const double *part_sum, const int64_t *part_cnt,
                                   const double *part_tot, int32_t n_blocks_total, int32_t n_blocks_per_rank,
                                   int64_t rank_stride, int32_t *new_k, const int32_t *n_new,
                                   int32_t *remap_scratch, double *out_scalars, int32_t *status, void *stream)
{
    (void)status;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(n_blocks_total >= 1 && n_blocks_total <= 64, "1 <= n_blocks_total <= 64");
    SEGK_REQUIRE(n_blocks_per_rank >= 1 && n_blocks_total % n_blocks_per_rank == 0, "blocks per rank");
    SEGK_REQUIRE(m->K_max <= 8192, "batch mode supports K_max <= 8192");
    hipStream_t st = (hipStream_t)stream;
    int64_t tot = (int64_t)m->K_max * c->D;
    DISPATCH_XT(c, {
        hipLaunchKernelGGL(k_batch_combine<XT>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, *c, *m,
                           part_sum, part_cnt, part_tot, n_blocks_total, n_blocks_per_rank, rank_stride, out_scalars);
        hipLaunchKernelGGL(k_batch_clean<XT>, dim3(1), dim3(256), 0, st, *c, *m, remap_scratch, out_scalars);
    });
    int64_t nslot = (int64_t)(utt_hi - utt_lo) * c->N_max;
    if (nslot > 0)
        hipLaunchKernelGGL(k_batch_relabel, dim3((unsigned)((nslot + 255) / 256)), dim3(256), 0, st, *c, utt_lo,
                           utt_hi, new_k, n_new, remap_scratch, (double *)m->mnorm_max);
    SEGK_LAUNCH_CHECK();
    rc = kmeans_prepare_impl(ctx, c, m, stream, /* mnorm_max already zero */ nslot > 0);
    if (rc) return rc;
    // clean_components leaves exact copies behind (the moved rows, the inactive rows): out of the filters' images,
    // or every embedding near such a pair is a tie for the full scan.  SEGK_MARK_DUPS=0: leave them in.
    const char *md = getenv("SEGK_MARK_DUPS");
    if (md && atoi(md) == 0) return SEGK_OK;
    return segk_kmeans_mark_duplicates(ctx, c, m, nullptr, stream);
}

int32_t segk_kmeans_assignments_from_tokens(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                                            int32_t utt_lo, int32_t utt_hi, const int32_t *new_tok,
                                            const int32_t *new_k, const int32_t *n_new, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(m->assignments, 0xff, sizeof(int32_t) * (size_t)c->n_emb, st));
    int64_t nslot = (int64_t)(utt_hi - utt_lo) * c->N_max;
    if (nslot > 0)
        hipLaunchKernelGGL(k_assign_from_tokens, dim3((unsigned)((nslot + 255) / 256)), dim3(256), 0, st, *c, *m,
                           utt_lo, utt_hi, new_tok, new_k, n_new);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_sum_neg_sqrd_norm(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, double *out,
                                      void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(double), st));
    int64_t grid = (c->n_emb + 3) / 4;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_sum_neg_sqrd_norm<XT>, dim3((unsigned)grid), dim3(256), 0, st, *c, *m,
                                       out););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_init_stats(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(m->K, 0, sizeof(int32_t), st));
    int64_t grid = ((int64_t)m->K_max + 3) / 4;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_init_stats<XT>, dim3((unsigned)grid), dim3(256), 0, st, *c, *m););
    SEGK_LAUNCH_CHECK();
    return segk_kmeans_prepare(ctx, c, m, stream);
}

int32_t segk_dp_tri(segk_ctx *ctx, int32_t kind, const double *vecs, const int32_t *Ns, const int64_t *offs,
                    int32_t n_prob, int32_t n_slices_min, int32_t n_slices_max, double log_p_continue,
                    double anneal_temp, const double *uniforms, int64_t u_stride, uint8_t *bounds,
                    int64_t b_stride, double *totals, int32_t *n_draws, int32_t *status, double *work,
                    int64_t w_stride, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(kind >= 0 && kind <= 2, "kind");
    SEGK_REQUIRE(n_slices_min == 0 || n_slices_min == 1, "n_slices_min must be 0 or 1");
    SEGK_REQUIRE(kind != 2 || uniforms != nullptr, "uniforms required for forward_backward");
    if (n_prob <= 0) return SEGK_OK;
    hipLaunchKernelGGL(k_dp_tri, dim3((n_prob + 63) / 64), dim3(64), 0, (hipStream_t)stream, kind, vecs, Ns, offs,
                       n_prob, n_slices_min, n_slices_max, log_p_continue, anneal_temp, uniforms, u_stride, bounds,
                       b_stride, totals, n_draws, status, work, w_stride);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

}  // extern "C"
