// segk_kmeans_dev.h -- device helpers, launch arguments and cross-unit declarations shared by the
// translation units of the k-means path (segk_prepare / segk_score_f32 / segk_score_sp / segk_score_h1 /
// segk_exact / segk_segment / segk_stats / segk_kmeans_api .hip).  Not part of the ABI.
#pragma once
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

// pw_base for float rows read 16 bytes at a time (8 <= n <= 128, n % 4 == 0, both pointers 16-byte aligned): the same
// eight strided accumulators, combine tree and sequential tail as numpy -- only the loads are wider
__device__ __forceinline__ float neg_sqd_exact_v4(const float *m, const float *x, int n)
{
    const int nfull = n & ~7;
    float r[8];
    {
        const float4 m0 = *reinterpret_cast<const float4 *>(m), m1 = *reinterpret_cast<const float4 *>(m + 4);
        const float4 x0 = *reinterpret_cast<const float4 *>(x), x1 = *reinterpret_cast<const float4 *>(x + 4);
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float delta = mv[q] - xv[q];
            r[q] = delta * delta;
        }
    }
    for (int i = 8; i < nfull; i += 8) {
        const float4 m0 = *reinterpret_cast<const float4 *>(m + i), m1 = *reinterpret_cast<const float4 *>(m + i + 4);
        const float4 x0 = *reinterpret_cast<const float4 *>(x + i), x1 = *reinterpret_cast<const float4 *>(x + i + 4);
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float delta = mv[q] - xv[q];
            r[q] += delta * delta;
        }
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (n & 4) {                                            // the sequential tail: four elements
        const float4 mt = *reinterpret_cast<const float4 *>(m + nfull), xt = *reinterpret_cast<const float4 *>(x + nfull);
        const float mv[4] = {mt.x, mt.y, mt.z, mt.w}, xv[4] = {xt.x, xt.y, xt.z, xt.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float delta = mv[q] - xv[q];
            res += delta * delta;
        }
    }
    return -res;
}

// The same with two elements per instruction (v_pk_add_f32 with the second operand negated, v_pk_mul_f32, v_pk_add_f32: each
// half is rounded like the scalar operation; -ffp-contract=off keeps multiply and add apart)
__device__ __forceinline__ float neg_sqd_exact_v4_pk(const float *m, const float *x, int n)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    auto pk_sub = [](f32x2_t a, f32x2_t b) -> f32x2_t {
        f32x2_t d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
    const int nfull = n & ~7;
    f32x2_t r01, r23, r45, r67;
    {
        const f32x4_t m0 = *reinterpret_cast<const f32x4_t *>(m), m1 = *reinterpret_cast<const f32x4_t *>(m + 4);
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t *>(x), x1 = *reinterpret_cast<const f32x4_t *>(x + 4);
        const f32x2_t d0 = pk_sub(m0.xy, x0.xy), d1 = pk_sub(m0.zw, x0.zw), d2 = pk_sub(m1.xy, x1.xy), d3 = pk_sub(m1.zw, x1.zw);
        r01 = d0 * d0; r23 = d1 * d1; r45 = d2 * d2; r67 = d3 * d3;
    }
    for (int i = 8; i < nfull; i += 8) {
        const f32x4_t m0 = *reinterpret_cast<const f32x4_t *>(m + i), m1 = *reinterpret_cast<const f32x4_t *>(m + i + 4);
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t *>(x + i), x1 = *reinterpret_cast<const f32x4_t *>(x + i + 4);
        const f32x2_t d0 = pk_sub(m0.xy, x0.xy), d1 = pk_sub(m0.zw, x0.zw), d2 = pk_sub(m1.xy, x1.xy), d3 = pk_sub(m1.zw, x1.zw);
        r01 += d0 * d0; r23 += d1 * d1; r45 += d2 * d2; r67 += d3 * d3;
    }
    float res = ((r01.x + r01.y) + (r23.x + r23.y)) + ((r45.x + r45.y) + (r67.x + r67.y));
    if (n & 4) {                                            // the sequential tail: four elements
        const f32x4_t mt = *reinterpret_cast<const f32x4_t *>(m + nfull), xt = *reinterpret_cast<const f32x4_t *>(x + nfull);
        const f32x2_t d0 = pk_sub(mt.xy, xt.xy), d1 = pk_sub(mt.zw, xt.zw);
        const f32x2_t t0 = d0 * d0, t1 = d1 * d1;
        res += t0.x;
        res += t0.y;
        res += t1.x;
        res += t1.y;
    }
    return -res;
}

// Two components against one row: the row's elements are read once (the persistent sequential chain's score phase is bound by
// LDS instructions: three per pair and eight elements instead of four).  Per pair the arithmetic of neg_sqd_exact_v4_pk.
__device__ __forceinline__ void neg_sqd_exact_v4_pk2(const float *ma, const float *mb, const float *x, int n, float *out_a, float *out_b)
{
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    auto pk_sub = [](f32x2_t a, f32x2_t b) -> f32x2_t {
        f32x2_t d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
    const int nfull = n & ~7;
    f32x2_t a01, a23, a45, a67, b01, b23, b45, b67;
    {
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t *>(x), x1 = *reinterpret_cast<const f32x4_t *>(x + 4);
        const f32x4_t p0 = *reinterpret_cast<const f32x4_t *>(ma), p1 = *reinterpret_cast<const f32x4_t *>(ma + 4);
        const f32x4_t q0 = *reinterpret_cast<const f32x4_t *>(mb), q1 = *reinterpret_cast<const f32x4_t *>(mb + 4);
        const f32x2_t d0 = pk_sub(p0.xy, x0.xy), d1 = pk_sub(p0.zw, x0.zw), d2 = pk_sub(p1.xy, x1.xy), d3 = pk_sub(p1.zw, x1.zw);
        const f32x2_t e0 = pk_sub(q0.xy, x0.xy), e1 = pk_sub(q0.zw, x0.zw), e2 = pk_sub(q1.xy, x1.xy), e3 = pk_sub(q1.zw, x1.zw);
        a01 = d0 * d0; a23 = d1 * d1; a45 = d2 * d2; a67 = d3 * d3;
        b01 = e0 * e0; b23 = e1 * e1; b45 = e2 * e2; b67 = e3 * e3;
    }
    for (int i = 8; i < nfull; i += 8) {
        const f32x4_t x0 = *reinterpret_cast<const f32x4_t *>(x + i), x1 = *reinterpret_cast<const f32x4_t *>(x + i + 4);
        const f32x4_t p0 = *reinterpret_cast<const f32x4_t *>(ma + i), p1 = *reinterpret_cast<const f32x4_t *>(ma + i + 4);
        const f32x4_t q0 = *reinterpret_cast<const f32x4_t *>(mb + i), q1 = *reinterpret_cast<const f32x4_t *>(mb + i + 4);
        const f32x2_t d0 = pk_sub(p0.xy, x0.xy), d1 = pk_sub(p0.zw, x0.zw), d2 = pk_sub(p1.xy, x1.xy), d3 = pk_sub(p1.zw, x1.zw);
        const f32x2_t e0 = pk_sub(q0.xy, x0.xy), e1 = pk_sub(q0.zw, x0.zw), e2 = pk_sub(q1.xy, x1.xy), e3 = pk_sub(q1.zw, x1.zw);
        a01 += d0 * d0; a23 += d1 * d1; a45 += d2 * d2; a67 += d3 * d3;
        b01 += e0 * e0; b23 += e1 * e1; b45 += e2 * e2; b67 += e3 * e3;
    }
    float ra = ((a01.x + a01.y) + (a23.x + a23.y)) + ((a45.x + a45.y) + (a67.x + a67.y));
    float rb = ((b01.x + b01.y) + (b23.x + b23.y)) + ((b45.x + b45.y) + (b67.x + b67.y));
    if (n & 4) {                                            // the sequential tail: four elements
        const f32x4_t xt = *reinterpret_cast<const f32x4_t *>(x + nfull);
        const f32x4_t pt = *reinterpret_cast<const f32x4_t *>(ma + nfull), qt = *reinterpret_cast<const f32x4_t *>(mb + nfull);
        const f32x2_t d0 = pk_sub(pt.xy, xt.xy), d1 = pk_sub(pt.zw, xt.zw), e0 = pk_sub(qt.xy, xt.xy), e1 = pk_sub(qt.zw, xt.zw);
        const f32x2_t t0 = d0 * d0, t1 = d1 * d1, s0 = e0 * e0, s1 = e1 * e1;
        ra += t0.x; ra += t0.y; ra += t1.x; ra += t1.y;
        rb += s0.x; rb += s0.y; rb += s1.x; rb += s1.y;
    }
    *out_a = -ra;
    *out_b = -rb;
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

// value hash of one element of a row of `means` (k_kmeans_mark_dups): -0 and +0 hash alike, the per-element
// terms add up commutatively, so lanes can hash strided parts of a row and sum
__device__ __forceinline__ unsigned long long segk_elem_hash(double v, int d)
{
    unsigned long long z = (unsigned long long)__double_as_longlong(v + 0.0) + 0x9E3779B97F4A7C15ull * (unsigned long long)(d + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// single-instruction max (fmaxf() makes hipcc add a canonicalising v_max on MFMA outputs)
__device__ __forceinline__ float vmax_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

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
    int32_t n_dev_off;           /* rows of the queue an earlier launch of the same stage covers */
    int32_t *pre_queue, *pre_count;
    int pre_cap, K_max;
    const float *xerr;           /* pre-filter: |x - x1| per row (k_corpus_resid_sp) */
    const int32_t *n_tiles_dev;  /* optional: the number of leading tiles that hold components, on the device (the batch sampler
                                    packs the occupied slots into the first tiles); the kernels walk min(n_tiles, *n_tiles_dev) */
    unsigned long long *stamp;   /* -DSEGK_STAMP development builds: s_memtime at phase boundaries, 8 per workgroup */
};

// development (-DSEGK_STAMP builds only): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of the sweep's tail kernels,
// [kernel][workgroup % 1024][8] in a device buffer whose address comes through the environment (tools/diag_tail_stamps.py)
#ifdef SEGK_STAMP
static __device__ unsigned long long segk_tstamp_buf;           // the address as an integer: stores go through a global (address_space(1)) pointer
#define SEGK_TSTAMP_AT(kern, ph) ((__attribute__((address_space(1))) unsigned long long *)segk_tstamp_buf + (((kern) * 1024 + (blockIdx.x & 1023)) * 8) + (ph))
#define SEGK_TSTAMP(kern, ph)                                                                           \
    do {                                                                                                \
        if (segk_tstamp_buf && threadIdx.x == 0) *SEGK_TSTAMP_AT(kern, ph) = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define SEGK_TSTAMP_MAX(kern, ph)                                                                       \
    do {                                                                                                \
        if (segk_tstamp_buf && (threadIdx.x & 63) == 0) *SEGK_TSTAMP_AT(kern, ph) = __builtin_amdgcn_s_memrealtime(); /* the last wave to finish writes last */ \
    } while (0)
static inline void segk_tstamp_bind()
{
    static unsigned long long bound = ~0ull;       // (hipMemcpyToSymbol synchronises: only when the address changes)
    const char *e = getenv("SEGK_TSTAMP_PTR");
    unsigned long long p = e ? strtoull(e, nullptr, 0) : 0ull;
    if (p != bound) (void)hipMemcpyToSymbol(HIP_SYMBOL(segk_tstamp_buf), &p, sizeof(p));
    bound = p;
}
#else
#define SEGK_TSTAMP(kern, ph) do { } while (0)
#define SEGK_TSTAMP_MAX(kern, ph) do { } while (0)
static inline void segk_tstamp_bind() {}
#endif

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

__device__ __forceinline__ float filter_tau_h1(float xn, float M, int D, float ex, float Em)
{
    // operand rounding: |sum x1 m1 - sum x m| <= |x1| |m1 - m| + |x1 - x| |m| <= (|x| + e_x) E_m + e_x M with the
    // residual norms of THIS row and the worst component (k_corpus_resid_sp / k_kmeans_prepare_sp), never more
    // than the a-priori 1.01 * 2^-10 |x| M
    const float meas = (xn + ex) * Em + ex * M;
    const float apriori = 1.01f * 9.765625e-4f * xn * M;
    return filter_tau_sp(xn, M, D, 2) + 2.5f * 1.00001f * fminf(meas, apriori);
}

// A handful of left-over rows (fewer than SEGK_TAIL_QUEUE): not worth three more launches -- they
// are appended to the ambiguity queue and take the full reference-arithmetic scan.
#define SEGK_TAIL_QUEUE 2048
static __global__ void k_score_queue_rows(ScoreArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n) return;
    const int32_t id = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
    if (id < 0) return;
    const int q = atomicAdd(A.cand.count, 1);
    if (q < A.amb_cap) A.cand.queue[q] = id;
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

static inline int check_corpus(const segk_corpus *c)
{
    SEGK_REQUIRE(c != nullptr, "corpus is NULL");
    SEGK_REQUIRE(c->x_dtype == SEGK_F32 || c->x_dtype == SEGK_F64, "x_dtype");
    SEGK_REQUIRE(c->D > 0 && c->n_emb > 0, "empty corpus");
    SEGK_REQUIRE(c->ld32 % 4 == 0 && c->ld32 >= c->D, "ld32 must be D rounded up to a multiple of 4");
    return SEGK_OK;
}

static inline bool segk_use_b3(const segk_corpus *c, const segk_kmeans *m)
{
    const char *e = getenv("SEGK_SCORE_B3");
    if (e && atoi(e) == 0) return false;
    return c->Xb3 && (c->sp_pieces == 2 || c->sp_pieces == 3) && m->tiles_b3 && c->x_dtype == SEGK_F32 && c->D >= 8 &&
           c->D <= 128;
}

static inline int score_checks(const segk_corpus *c, const segk_kmeans *m, const int32_t *ids, int64_t row0, int64_t n,
                        const segk_cand *cand)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && cand && cand->k && cand->f && cand->s && cand->queue && cand->count && c->X32,
                 "score operands");
    SEGK_REQUIRE(ids != nullptr || (row0 >= 0 && row0 + n <= c->n_emb), "row range");
    // the queues of undecided rows (cand->queue, the context's pre-filter queue) hold n_emb entries: a longer
    // id list (repeated rows) could overflow them and leave rows with filter-stage values
    SEGK_REQUIRE(n <= c->n_emb, "at most n_emb rows per score call (split longer id lists)");
    return SEGK_OK;
}

// ======================================================================================
// means -> fp32 MFMA operand image of ONE tile of 32 components (layout: segk_internal.h): called by a whole
// workgroup of 256 threads (k_kmeans_prepare, and the batch sweep's finalize kernel for the rows it has just
// written).  mnorm2_bits: running max_k |m_k|^2 (atomicMax on the bit pattern of non-negative doubles).
// ======================================================================================
template <typename XT>
__device__ __forceinline__ void dev_prepare_tile(const XT *means, int K_max, int D, float *tiles, unsigned long long *mnorm2_bits,
                                                 unsigned int *zero_slot, unsigned long long *row_hash, const int tile)
{
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

// tiles image: [header 1024 floats: int32 exponent b at [0]] then per tile [s][p][lane][8] pieces + 32 constants
// consts == NULL: the k-means constants -|m|^2/2; otherwise consts[k] (< -1e37: component absent) -- the
// log-sum-exp use of the kernel (segk_fbbatch.hip), whose rows are not means.
template <int P>
__device__ __forceinline__ void dev_prepare_sp_tile(const float *means, int K_max, int D, float *tiles, const double *mnorm2,
                                                    const unsigned char *ximg, const double *consts, const int tile)
{
    typedef typename SegkPiece<P>::T T;
    const int KS = segk_b3_kp(D) / 16;
    const int stride = segk_sp_tile_stride(D, P);
    // max |m_d| <= sqrt(max |m|^2): every block derives the same exponent
    const int eb = P == 2 ? sp_exponent((float)(sqrt(*mnorm2) * (1.0 + 1e-6))) : 0;
    const int ea = ((const int *)ximg)[1];
    if (tile == 0 && threadIdx.x == 0) ((int *)tiles)[0] = eb;
    float *Tt = tiles + 1024 + (int64_t)tile * stride;
    T *Tb = (T *)Tt;
    __shared__ double nrm[32];
    // the tile's 32 rows (contiguous in `means`) into LDS first: coalesced loads, all in flight -- read in place, the image
    // loop below fetched one element per lane from 32 different rows per instruction, 14 dependent rounds (8 of the post
    // kernel's 14 us)
    constexpr int PITCH = 129;                                       // odd pitch: the 32 rows of a lane group on 32 banks
    __shared__ float rowsl[32 * PITCH];
    const bool staged = D <= 128;                                    // (wider rows -- the FBGMM span-score images -- are read in place)
    const float *src = means + (int64_t)tile * 32 * D;
    if (staged) {
        const int nrow = K_max - tile * 32 < 32 ? K_max - tile * 32 : 32;
        for (int idx = threadIdx.x; idx < 32 * D; idx += blockDim.x) {
            const int ci = idx / D, d = idx - ci * D;
            rowsl[ci * PITCH + d] = ci < nrow ? src[idx] : 0.f;
        }
    }
    __syncthreads();
    SEGK_TSTAMP(4, 5);
    auto elem = [&](int ci, int d) -> float { return staged ? rowsl[ci * PITCH + d] : src[(int64_t)ci * D + d]; };   // live rows only
    {
        const int ci = threadIdx.x >> 3, sub = threadIdx.x & 7;      // 256 threads = 32 x 8
        const int comp = tile * 32 + ci;
        double s = 0.0, rs = 0.0;
        if (comp < K_max)
            for (int d = sub; d < D; d += 8) {
                const float mv = elem(ci, d);
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
    SEGK_TSTAMP(4, 6);
    // one (k-step, lane) fragment per thread and trip: its eight elements split into pieces, one 16-byte store per piece
    for (int pr = threadIdx.x; pr < KS * 64; pr += blockDim.x) {
        const int sidx = pr >> 6, lane = pr & 63;
        const int ci = lane & 31;
        const bool live = tile * 32 + ci < K_max;
        typename SegkPiece<P>::V8 out[P];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int d = segk_b3_dim(16 * sidx + 8 * (lane >> 5) + i);
            const float v = (live && d < D) ? ldexpf(elem(ci, d), eb) : 0.f;
            T pc[P];
            split_sp<P>(v, pc);
#pragma unroll
            for (int q = 0; q < P; q++) out[q][i] = pc[q];
        }
#pragma unroll
        for (int q = 0; q < P; q++) *reinterpret_cast<typename SegkPiece<P>::V8 *>(Tb + ((sidx * P + q) * 64 + lane) * 8) = out[q];
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

// ======================================================================================
// Exact duplicates among the rows of `means` (see k_kmeans_mark_dups in segk_prepare.hip): pieces shared with
// the batch sweep's post kernel, which marks the duplicates of ONE tile per workgroup.
//   dev_dup_table: open-addressing table in LDS (SEGK_DUP_TB slots): key = row hash, value = the lowest row
//                  index with that hash; all threads of the workgroup, K_max <= SEGK_DUP_TB / 2.
//   dev_dup_first: the first row whose hash equals row k's (k itself when there is no earlier one).
// ======================================================================================
#define SEGK_DUP_TB 4096
__device__ __forceinline__ void dev_dup_table(unsigned long long *keys, int32_t *first, const unsigned long long *row_hash, int K_max)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < SEGK_DUP_TB; i += blockDim.x) { keys[i] = 0ull; first[i] = 0x7fffffff; }
    __syncthreads();
    for (int k = tid; k < K_max; k += blockDim.x) {
        const unsigned long long h = row_hash[k];
        for (unsigned slot = (unsigned)(h >> 20) & (SEGK_DUP_TB - 1);; slot = (slot + 1) & (SEGK_DUP_TB - 1)) {
            const unsigned long long prev = atomicCAS(&keys[slot], 0ull, h);
            if (prev == 0ull || prev == h) { atomicMin(&first[slot], k); break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ int dev_dup_first(const unsigned long long *keys, const int32_t *first, unsigned long long h)
{
    unsigned slot = (unsigned)(h >> 20) & (SEGK_DUP_TB - 1);
    while (keys[slot] != h) slot = (slot + 1) & (SEGK_DUP_TB - 1);
    return first[slot];
}

// queue lengths of a score call: the caller's ambiguity queue and the pre-filter's per-chunk counters
static __global__ void k_zero_two(int32_t *a, int32_t *b)
{
    if (threadIdx.x == 0) *a = 0;
    if (threadIdx.x < 16) b[threadIdx.x] = 0;
}
// segk_kmeans_score defers that clearing to the first kernel of the path its filter takes (ctx->defer_zero)
static inline void segk_flush_deferred_zero(segk_ctx *ctx, hipStream_t st)
{
    if (ctx && ctx->defer_zero) {
        hipLaunchKernelGGL(k_zero_two, dim3(1), dim3(64), 0, st, ctx->defer_zero, ctx->pre_queue);
        ctx->defer_zero = nullptr;
    }
}

// ---- cross-unit entry points (host side; each lives in the unit named) --------------------------------
// segk_prepare.hip
int segk_kmeans_prepare_impl(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream, bool mnorm_zeroed);
// segk_score_f32.hip: the fp32-MFMA filter (float64 data, D outside 8..128, SEGK_SCORE_B3=0)
int segk_dispatch_score_f32(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const ScoreArgs &A, hipStream_t st);
// segk_score_sp.hip: the split-precision filter (pieces = 2 fp16x2, 3 bf16x3) and the pre-filter's second stage
int segk_dispatch_score_sp(segk_ctx *ctx, const ScoreArgs &A, int ks, int pieces, hipStream_t st);
int segk_launch_sp_second(segk_ctx *ctx, const ScoreArgs &B, int ks, hipStream_t st);
int segk_launch_clean(const segk_corpus *c, segk_kmeans *m, int32_t *status, hipStream_t st, int32_t *relog = nullptr);
int segk_launch_seq_chain(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, const int32_t *order, int n_order, int n_slices_max,
                          double wip, uint8_t *boundaries, int32_t *old_tok, int32_t *new_tok, int32_t *new_k, int32_t *n_old,
                          int32_t *n_new, int32_t *n_flag, double *out_total, int32_t *status, hipStream_t st);
// segk_score_h1.hip: one-product pre-filter + exact pair stage + second stage
int segk_dispatch_score_pre(segk_ctx *ctx, const ScoreArgs &A, int ks, hipStream_t st);
// segk_score_hint.hip: value-only top-2 on the matrix cores + exact stage that verifies a hint per row (cand.k on entry)
int segk_dispatch_score_hint(segk_ctx *ctx, const ScoreArgs &A, const int32_t *remap, int64_t n_emb, int ks, hipStream_t st);
// segk_score_band.hip: the hinted path's undecided rows -- candidates inside the band of the filter's maximum, exact scores
bool segk_band_applies(const ScoreArgs &A);
int segk_launch_band(segk_ctx *ctx, const ScoreArgs &A, const float *thr, int64_t call_rows, int ks, hipStream_t st);
// segk_exact.hip / segk_stats.hip: the pieces of the sequential (reference-chain) sweep
int segk_launch_seq_score(const segk_corpus *c, const segk_kmeans *m, int utt, const segk_cand *cand, unsigned long long *keys,
                          hipStream_t st);
int segk_launch_update_utt(const segk_corpus *c, segk_kmeans *m, int utt, const int32_t *old_tok, const int32_t *new_tok,
                           const int32_t *new_k, const int32_t *n_old, const int32_t *n_new, int32_t *status, hipStream_t st);
// segk_exact.hip: full scan of the ambiguity queue
int segk_resolve_on(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids, int64_t row0, int64_t n,
                    const segk_cand *cand, int32_t *status, void *stream);
