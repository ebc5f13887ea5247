// segk_score_h1.hip -- A1 one-product fp16 pre-filter, the exact stage of its decided rows (k_kmeans_exact_pair), launch plan
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"

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
// ABL: timing-only ablations for development (SEGK_H1_ABL; results are wrong): 1 no top-2 drain (one running maximum per
// block), 2 drain without the pair index, 3 drain without the MFMAs
template <int KS, int NBLK, int ABL = 0>
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
        const T *xp = (const T *)((const unsigned char *)A.X32 + SEGK_SP_HEADER) + (int64_t)(rowid[b] >= 0 ? rowid[b] : 0) * KP + 8 * h;      // plane 0
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
        if constexpr (ABL == 1) {                                                     \
            if ((pi) == 0) m1[O_] = vmax_f32(m1[O_], ACC[0]);                         \
        } else if constexpr (ABL == 2) {                                              \
            const float tmp_ = __builtin_amdgcn_fmed3f(m1[O_], ACC[2 * (pi)], ACC[2 * (pi) + 1]); \
            float nm_;                                                                \
            asm volatile("v_max_f32 %1, %1, %2\n\t"                                   \
                         "v_max3_f32 %0, %3, %4, %5"                                  \
                         : "=&v"(nm_), "+v"(m2[O_])                                   \
                         : "v"(tmp_), "v"(m1[O_]), "v"(ACC[2 * (pi)]), "v"(ACC[2 * (pi) + 1])); \
            m1[O_] = nm_;                                                             \
        } else {                                                                      \
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
        }                                                                             \
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
            if constexpr (ABL != 3) acc[(N_) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], xb[N_][s], acc[(N_) & 1], 0, 0, 0); \
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

// Exact stage of the pre-filter's decided rows: cand.k = (c | SEGK_PAIR_PENDING) names the pair (c, c + 1);
// both members are scored in the reference's float32 arithmetic (sp_exact_score_x) and the larger wins
// (the lower index on a tie, as np.argmax).  A kernel of its own because inside the MFMA kernel these
// reads -- 16 bytes per lane from 64 different rows per instruction, eight waves per CU, one row block
// after the other -- took 60 % of a workgroup's lifetime (s_memtime stamps, profiles/README.md r01_h).
// The arithmetic wants a lane to own whole strided accumulators of one (row, member), the memory system
// wants whole lines.  Two forms are left (the first two -- members staged through LDS lane by lane, members
// gathered into registers -- were retired in round 3; DESIGN.md section 2 and `git log` have them):
#define SEGK_PAIR_ROWS 16

// k_kmeans_exact_pair3 (tables that need more than eight LDS ranges).  Per-lane gathers of the means (16 bytes per lane, two
// lanes per 400-byte row and instruction) touch 32 cache lines per load instruction: 49 M tag lookups in the vector L1 for 1.3 GB,
// one per clock and CU -- the stage ran at the L1's lookup rate (137 us alone), not at the memory system's
// (profiles/README.md, r02_t).  Here the member rows are loaded the way the x rows are -- whole rows, RPI per
// instruction, 7 lines each -- and transposed through LDS: 0.2 M lookups per CU instead of 0.75 M.  Everything a step
// waits on is one step ahead in flight (rows and means of step s + 1 in registers while step s is summed from LDS).
// LDS row pitch `ps` 16-byte slots, ps = 2 (mod 4): the 16 lanes of one ds_read_b128 pass fall on 16 distinct slots.
template <int KS, int V>
__global__ __launch_bounds__(64) void k_kmeans_exact_pair3(ScoreArgs A)
{
    // D = 16 KS - 4 V is a compile-time constant: the block loop has no branches and all of a step's LDS reads are in
    // flight together (with D at run time every 32-byte block waited for its own two reads: ~2000 cycles of LDS
    // latency per step)
    constexpr int D = 16 * KS - 4 * V, D4 = D >> 2;
    constexpr int ps = D4 + ((2 - D4) & 3);
    constexpr int R = SEGK_PAIR_ROWS;
    constexpr int C4 = KS * 4 + 2;                                 // lanes per row in a load instruction
    constexpr int RPI = 64 / C4;                                   // rows per load instruction
    constexpr int NLX = (R + RPI - 1) / RPI, NLM = (2 * R + RPI - 1) / RPI;
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [R][ps * 4] x rows, [2 R][ps * 4] member rows
    const int lane = threadIdx.x;
    constexpr int LD = ps * 4;
    float *xs = lds, *ms = lds + R * LD;
    const int64_t n_steps = (A.n + R - 1) / R;
    const int sub = lane / C4, c4 = lane - sub * C4;
    const int sub_c = sub < RPI ? sub : RPI - 1;
    const unsigned off = 16u * (unsigned)(c4 < D4 ? c4 : D4 - 1);
    const bool wr = sub < RPI && c4 < ps;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f32x4_t *gptr_t;

    auto fetch_rid = [&](int64_t step) -> int32_t {
        const int64_t r = step * R + lane;
        int32_t rid = -1;
        if (step < n_steps && lane < R && r < A.n) rid = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
        return rid;
    };
    auto fetch_k = [&](int32_t rid) -> int32_t { return rid >= 0 ? A.cand.k[rid] : 0; };
    auto decode = [&](int32_t &rid, int32_t k) -> int32_t {       // pair base, or -1 (and rid = -1) when not pending
        if (rid >= 0 && k >= 0 && (k & SEGK_PAIR_PENDING) && (k & ~SEGK_PAIR_PENDING) < A.K_max) return k & ~SEGK_PAIR_PENDING;
        rid = -1;
        return -1;
    };
    f32x4_t vx[NLX], vm[NLM];
    // the loads of one step: lane l < R holds the step's row l (rid_l, -1: none) and its pair base c_l
    auto issue = [&](int32_t rid_l, int32_t c_l, int64_t step_) {
        int64_t r_any = rid_l;                                  // some valid row for the lanes without one
        if (rid_l < 0) {
            r_any = A.ids ? 0 : A.row0 + step_ * R + (lane < R ? lane : 0);
            if (A.ids || r_any >= A.row0 + A.n) r_any = A.ids ? 0 : A.row0;
        }
        // member rows: item i = 2 row + member on lane i, clamped into the table (the result of a member beyond it is dropped)
        const int32_t c_it = __shfl(c_l, lane >> 1), rid_it = __shfl(rid_l, lane >> 1);
        int cm = rid_it >= 0 ? c_it + (lane & 1) : 0;
        if (cm >= A.K_max) cm = A.K_max - 1;
        // byte offsets worked out once per row, not once per load instruction
        const unsigned moff = __umul24((unsigned)cm, (unsigned)(D * 4));                 // K_max < 2^24, K_max * D * 4 < 2^32
        const uintptr_t xaddr = (uintptr_t)(A.xrows32 + r_any * A.ld32);
        const unsigned xlo = (unsigned)xaddr, xhi = (unsigned)(xaddr >> 32);
#pragma unroll
        for (int t = 0; t < NLX; t++) {
            const int src = (RPI * t + sub_c) & (R - 1);
            const uintptr_t base = ((uintptr_t)(unsigned)__shfl((int)xhi, src) << 32) | (unsigned)__shfl((int)xlo, src);
            vx[t] = *reinterpret_cast<gptr_t>(base + off);
        }
#pragma unroll
        for (int t = 0; t < NLM; t++) {
            const unsigned mo = (unsigned)__shfl((int)moff, (RPI * t + sub_c) & (2 * R - 1));
            vm[t] = *reinterpret_cast<gptr_t>((uintptr_t)A.means32 + mo + off);
        }
    };

    const int item = lane >> 1, h = lane & 1, row = item >> 1;
    constexpr int nfull = D & ~7, nblk = nfull >> 3, rem = D & 7;
    const float *xrow = xs + row * LD + 4 * h, *mrow = ms + item * LD + 4 * h;

    int64_t step = blockIdx.x;
    int32_t rid0 = fetch_rid(step);
    int32_t c0 = decode(rid0, fetch_k(rid0));
    issue(rid0, c0, step);
    int32_t rid1 = fetch_rid(step + gridDim.x);
    int32_t k1 = fetch_k(rid1);
    int32_t rid2 = fetch_rid(step + 2 * (int64_t)gridDim.x);
    for (; step < n_steps; step += gridDim.x) {
        const int32_t rid = __shfl(rid0, row), c = __shfl(c0, row);
        // this step's rows into LDS, the next step's into flight
#pragma unroll
        for (int t = 0; t < NLX; t++)
            if (wr && RPI * t + sub < R) *reinterpret_cast<f32x4_t *>(xs + (RPI * t + sub) * LD + 4 * c4) = vx[t];
#pragma unroll
        for (int t = 0; t < NLM; t++)
            if (wr && RPI * t + sub < 2 * R) *reinterpret_cast<f32x4_t *>(ms + (RPI * t + sub) * LD + 4 * c4) = vm[t];
        const int32_t c1 = decode(rid1, k1);
        issue(rid1, c1, step + gridDim.x);
        rid0 = rid1; c0 = c1;
        rid1 = rid2;
        k1 = fetch_k(rid1);
        rid2 = fetch_rid(step + 3 * (int64_t)gridDim.x);
        // the reference's float32 -(deltas*deltas).sum() in numpy's pairwise order: this lane owns the strided
        // accumulators r_{4h..4h+3}
        f32x4_t xv[nblk + 1], mv[nblk + 1];
#pragma unroll
        for (int b = 0; b < nblk; b++) {
            xv[b] = *reinterpret_cast<const f32x4_t *>(xrow + 8 * b);
            mv[b] = *reinterpret_cast<const f32x4_t *>(mrow + 8 * b);
        }
        if (rem) {                                                 // the four tail elements, the same on both lanes of the item
            xv[nblk] = *reinterpret_cast<const f32x4_t *>(xrow - 4 * h + nfull);
            mv[nblk] = *reinterpret_cast<const f32x4_t *>(mrow - 4 * h + nfull);
        }
        // two elements per instruction (v_pk_add_f32 / v_pk_mul_f32: each half rounded like the scalar operation)
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        auto pk_sub = [](f32x2_t a, f32x2_t b2) -> f32x2_t {       // a - b, both halves in one instruction
            f32x2_t d;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b2));
            return d;
        };
        f32x2_t rl = {0.f, 0.f}, rh = {0.f, 0.f};
        float tt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < nblk; b++) {
            const f32x2_t dl = pk_sub(mv[b].xy, xv[b].xy), dh = pk_sub(mv[b].zw, xv[b].zw);
            const f32x2_t tl = dl * dl, th = dh * dh;
            rl = b == 0 ? tl : rl + tl;
            rh = b == 0 ? th : rh + th;
        }
        if (rem) {
            const f32x2_t dl = pk_sub(mv[nblk].xy, xv[nblk].xy), dh = pk_sub(mv[nblk].zw, xv[nblk].zw);
            const f32x2_t tl = dl * dl, th = dh * dh;
            tt[0] = tl.x; tt[1] = tl.y; tt[2] = th.x; tt[3] = th.y;
        }
        float res = (rl.x + rl.y) + (rh.x + rh.y);
        const float ro = __shfl_xor(res, 1);
        res = (h == 0) ? res + ro : ro + res;                      // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
        if (rem > 0) res += tt[0];
        if (rem > 1) res += tt[1];
        if (rem > 2) res += tt[2];
        if (rem > 3) res += tt[3];
        const float sc = -res;
        const float so = __shfl_xor(sc, 2);
        if ((lane & 3) == 0 && rid >= 0) {
            const bool second = c + 1 < A.K_max && so > sc;
            A.cand.k[rid] = second ? c + 1 : c;
            A.cand.s[rid] = (double)(second ? so : sc);
        }
    }
}

// Fourth form of the exact stage: the component table in LDS.  Forms two and three move 1.3 GB through the vector L1s
// (the rows, 0.42 GB from HBM, and two member rows per row, 0.84 GB of L2 hits) and run at what the L1s can have in
// flight -- ~34 GB/s per CU, 9 TB/s over the chip, 145 us -- whatever the instruction stream looks like (r02_t/u/v in
// profiles/README.md).  Here the table is split into P ranges of `cpp` components (+1: a pair may straddle), each
// workgroup keeps ONE range in LDS (pitch ps slots, see the third form) and walks a slice of the rows, taking only those
// whose pair base lies in its range: the members then cost LDS reads, and what goes through the L1s is the rows
// themselves plus P reads of cand.k.  One workgroup per CU, four waves; every wave works alone on its own rows: it scans
// 64 candidates at a time into a small ring (ballot + prefix count), takes 16 rows per step off the ring, and has the
// 13 loads per lane of the next step in flight while it sums the current one.
#define SEGK_PAIR4_RING 128
template <int KS, int V, int NW>
__global__ __launch_bounds__(64 * NW) void k_kmeans_exact_pair4(ScoreArgs A, int P, int cpp)
{
    constexpr int D = 16 * KS - 4 * V, D4 = D >> 2;
    constexpr int ps = D4 + ((2 - D4) & 3), LD = ps * 4;
    constexpr int nfull = D & ~7, nblk = nfull >> 3, rem = D & 7;
    constexpr int NX = nblk + (rem ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float lds[];    // [cpp + 1][LD] member rows, then the waves' rings
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) f32x4_t *gptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = blockIdx.x % P, chunk = blockIdx.x / P, n_chunk = gridDim.x / P;
    if (chunk >= n_chunk) return;
    const int c_lo = part * cpp;
    int c_n = A.K_max - c_lo;
    if (c_n > cpp + 1) c_n = cpp + 1;
    for (int i = tid; i < c_n * D4; i += 64 * NW) {
        const int r = i / D4, s4 = i - r * D4;
        *reinterpret_cast<f32x4_t *>(lds + r * LD + 4 * s4) = *reinterpret_cast<const f32x4_t *>(A.means32 + (int64_t)(c_lo + r) * D + 4 * s4);
    }
    __syncthreads();
    if (c_n <= 0) return;
    volatile int2 *ring = reinterpret_cast<volatile int2 *>(lds + (size_t)(cpp + 1) * LD) + wave * SEGK_PAIR4_RING;

    // this wave's rows: a multiple of 64 per wave
    const int64_t n_slots = (int64_t)n_chunk * NW;
    const int64_t per = ((A.n + n_slots - 1) / n_slots + 63) & ~(int64_t)63;
    int64_t pos = ((int64_t)chunk * NW + wave) * per;
    const int64_t r_end = pos + per < A.n ? pos + per : A.n;

    auto load_rid = [&](int64_t p_) -> int32_t {
        const int64_t r = p_ + lane;
        return r < r_end ? (A.ids ? A.ids[r] : (int32_t)(A.row0 + r)) : -1;
    };
    auto pk_sub = [](f32x2_t a, f32x2_t b2) -> f32x2_t {           // a - b, both halves in one instruction
        f32x2_t d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b2));
        return d;
    };
    const int row = lane >> 2, mem = (lane >> 1) & 1, h = lane & 1;

    int32_t rid_n = -1, k_n = 0;
    if (pos < r_end) {
        rid_n = load_rid(pos);
        k_n = rid_n >= 0 ? A.cand.k[rid_n] : 0;
    }
    int count = 0, head = 0;
    bool have_prev = false;
    f32x4_t xp[NX];
    int32_t p_rid = -1, p_base = 0;
    for (;;) {
        // candidates into the ring until a step's worth is there
        while (count < SEGK_PAIR_ROWS && pos < r_end) {
            const int32_t rid = rid_n, k = k_n;
            pos += 64;
            if (pos < r_end) {
                rid_n = load_rid(pos);
                k_n = rid_n >= 0 ? A.cand.k[rid_n] : 0;
            }
            int32_t base = -1;
            if (rid >= 0 && k >= 0 && (k & SEGK_PAIR_PENDING) && (k & ~SEGK_PAIR_PENDING) < A.K_max) base = (k & ~SEGK_PAIR_PENDING) - c_lo;
            const bool sel = base >= 0 && base < cpp;
            const unsigned long long mask = __ballot(sel);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (sel) {
                const int at = (head + count + before) & (SEGK_PAIR4_RING - 1);
                ring[at].x = rid;
                ring[at].y = base;
            }
            count += __popcll(mask);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n = count < SEGK_PAIR_ROWS ? count : SEGK_PAIR_ROWS;
        // the next step: its rows off the ring, their loads into flight
        int32_t n_rid = -1, n_base = 0;
        if (row < n) {
            const int at = (head + row) & (SEGK_PAIR4_RING - 1);
            n_rid = ring[at].x;
            n_base = ring[at].y;
        }
        head = (head + n) & (SEGK_PAIR4_RING - 1);
        count -= n;
        f32x4_t xn[NX];
        {
            const int64_t r_any = n_rid >= 0 ? (int64_t)n_rid : (A.ids ? 0 : A.row0);
            const uintptr_t xa = (uintptr_t)(A.xrows32 + r_any * A.ld32);
#pragma unroll
            for (int b = 0; b < nblk; b++) xn[b] = *reinterpret_cast<gptr_t>(xa + 16u * h + 32u * b);
            if (rem) xn[nblk] = *reinterpret_cast<gptr_t>(xa + 4u * nfull);
        }
        if (have_prev) {
            // the reference's float32 -(deltas*deltas).sum() in numpy's pairwise order: this lane owns the strided
            // accumulators r_{4h..4h+3} of member `mem` of its row
            int mi = p_base + mem;
            if (mi >= c_n) mi = c_n - 1;                           // beyond the table: the result is dropped
            const float *mrow = lds + mi * LD;
            f32x4_t mv[NX];
#pragma unroll
            for (int b = 0; b < nblk; b++) mv[b] = *reinterpret_cast<const f32x4_t *>(mrow + 4 * h + 8 * b);
            if (rem) mv[nblk] = *reinterpret_cast<const f32x4_t *>(mrow + nfull);
            f32x2_t rl = {0.f, 0.f}, rh = {0.f, 0.f};
#pragma unroll
            for (int b = 0; b < nblk; b++) {
                const f32x2_t dl = pk_sub(mv[b].xy, xp[b].xy), dh = pk_sub(mv[b].zw, xp[b].zw);
                const f32x2_t tl = dl * dl, th = dh * dh;
                rl = b == 0 ? tl : rl + tl;
                rh = b == 0 ? th : rh + th;
            }
            float res = (rl.x + rl.y) + (rh.x + rh.y);
            const float ro = __shfl_xor(res, 1);
            res = (h == 0) ? res + ro : ro + res;                  // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
            if (rem) {
                const f32x2_t dl = pk_sub(mv[nblk].xy, xp[nblk].xy), dh = pk_sub(mv[nblk].zw, xp[nblk].zw);
                const f32x2_t tl = dl * dl, th = dh * dh;
                res += tl.x;
                if (rem > 1) res += tl.y;
                if (rem > 2) res += th.x;
                if (rem > 3) res += th.y;
            }
            const float sc = -res;
            const float so = __shfl_xor(sc, 2);
            if ((lane & 3) == 0 && p_rid >= 0) {
                const int c = p_base + c_lo;
                const bool second = c + 1 < A.K_max && so > sc;
                A.cand.k[p_rid] = second ? c + 1 : c;
                A.cand.s[p_rid] = (double)(second ? so : sc);
            }
        }
        if (n == 0) break;
#pragma unroll
        for (int b = 0; b < NX; b++) xp[b] = xn[b];
        p_rid = n_rid;
        p_base = n_base;
        have_prev = true;
    }
}

// start-up kernel of the pre-filter path, ONE workgroup: the queue lengths of the call cleared (the caller's, when
// segk_kmeans_score deferred it, and the per-chunk counters), then the rows the big launches do not cover (fewer than
// SEGK_TAIL_QUEUE) put straight into the first chunk's queue (the order inside the queue is immaterial)
__global__ __launch_bounds__(1024) void k_pre_begin(ScoreArgs A, int32_t *zero_cnt, int32_t *hdr)
{
    if (threadIdx.x == 0 && zero_cnt) *zero_cnt = 0;
    if (threadIdx.x < 16) hdr[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t r = threadIdx.x; r < A.n; r += blockDim.x) {
        const int32_t id = A.ids ? A.ids[r] : (int32_t)(A.row0 + r);
        if (id < 0) continue;
        const int q = atomicAdd(A.pre_count, 1);
        if (q < A.pre_cap) A.pre_queue[q] = id;
        else {
            const int q2 = atomicAdd(A.cand.count, 1);
            if (q2 < A.amb_cap) A.cand.queue[q2] = id;
        }
    }
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
    // queue lengths: segk_kmeans_score leaves the clearing (its caller's queue length and this path's counters) to the
    // start-up kernel below; a bare segk_kmeans_filter call clears the counters here
    int32_t *zero_cnt = ctx->defer_zero;
    ctx->defer_zero = nullptr;
    ctx->pre_zeroed = 0;
    bool zero_pending = true;

    constexpr size_t lds = 2 * (size_t)(KS + 1) * 256 * sizeof(float);
    const int64_t slots = 2 * (int64_t)ctx->n_cu;      // two 4-wave workgroups per CU (launch bounds)
    // whole rounds of 512-row workgroups (four row blocks per wave), the remainder in 256-row workgroups
    const int64_t round4 = slots * 512;
    const int64_t n4 = (A.n / round4) * round4;
    const bool prof = segk_prof_now(ctx);
    const int slot = ctx->prof_n % SEGK_PROF_SLOTS;
    // the timed interval (segk_profile_*): the 512-row-workgroup launch when there is one, else the 256-row one
    auto prof_end = [&](int64_t rows, int launches) -> int {
        if (!prof) return SEGK_OK;
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = rows;
        ctx->prof_kind = 1;
        ctx->prof_launches = launches;
        ctx->prof_n++;
        return SEGK_OK;
    };
    // Everything on the caller's stream, one launch per stage.  Retired in round 3 after losing on MI355X
    // (profiles/README.md r02_c, r02_z; `git log` has the code): a pipeline over chunks of rounds with the exact stage and
    // the second stage of chunk i on two more streams beside the pre-filter of chunk i + 1 (-13 %: the kernels compete
    // for the same CUs and the same power budget), and the second stage + full scan on a second stream beside the exact
    // stage (no gain once that stage ran at HBM speed: it stretches every latency chain beside it).
    // a short remainder is only queued: first, so that nothing small sits between the big launches
    const int64_t rem = A.n - n4;
    const bool rem_queued = rem > 0 && rem < SEGK_TAIL_QUEUE && n4 > 0;
    ScoreArgs T = A;
    T.n = rem;
    T.row0 = A.row0 + n4;
    T.ids = A.ids ? A.ids + n4 : nullptr;
    T.pre_cap = (int)cap2;
    if (rem_queued) {                                  // one start-up kernel: clear the counters, queue the remainder
        hipLaunchKernelGGL(k_pre_begin, dim3(1), dim3(1024), 0, st, T, zero_cnt, ctx->pre_queue);
        zero_pending = false;
    }
    if (zero_pending) {
        if (zero_cnt) hipLaunchKernelGGL(k_zero_two, dim3(1), dim3(64), 0, st, zero_cnt, ctx->pre_queue);
        else SEGK_CHECK_HIP(hipMemsetAsync(ctx->pre_queue, 0, 16 * sizeof(int32_t), st));
    }
    // The exact stage: the component table in LDS, split into at most 8 ranges, eight waves, one workgroup per CU
    // (k_kmeans_exact_pair4); larger tables: whole-row loads transposed through LDS, pipelined (k_kmeans_exact_pair3; 32-bit
    // offsets into the means: segk_kmeans_filter takes this path for tables under 4 GB only)
    const int pitch4 = ((A.D >> 2) + ((2 - (A.D >> 2)) & 3)) * 16;
    constexpr int nw4 = 8;
    const int64_t lds4_budget = 158 * 1024 - nw4 * SEGK_PAIR4_RING * (int64_t)sizeof(int2);
    const int64_t cpp_max = lds4_budget / pitch4 - 1;
    const int parts4 = cpp_max > 0 ? (int)((A.K_max + cpp_max - 1) / cpp_max) : 99;
    const int cpp4 = parts4 > 0 ? (A.K_max + parts4 - 1) / parts4 : 0;
    const int pair_v = (parts4 <= 8 && ctx->n_cu >= parts4) ? 4 : 3;
    const int d4 = A.D >> 2, pitch_slots = d4 + ((2 - d4) & 3);
    const size_t lds_p = 3 * (size_t)SEGK_PAIR_ROWS * pitch_slots * 16;
    const int max_waves = 6;                           // pair3: waves per CU (12: 1 555, 8: 1 598, 6: 1 626 sweeps/s; r02_d)
    auto launch_pair = [&]() {
        const ScoreArgs &P = A;
        const int64_t steps = (A.n + SEGK_PAIR_ROWS - 1) / SEGK_PAIR_ROWS;
        int64_t waves = (int64_t)ctx->n_cu * (int64_t)((160 * 1024) / lds_p);
        if (waves > max_waves * (int64_t)ctx->n_cu) waves = max_waves * (int64_t)ctx->n_cu;
        if (waves > steps) waves = steps;
        if (pair_v == 4) {
            const size_t lds4 = (size_t)(cpp4 + 1) * pitch4 + nw4 * SEGK_PAIR4_RING * sizeof(int2);
            const unsigned grid4 = (unsigned)((ctx->n_cu / parts4) * parts4);
#define SEGK_PAIR4_LAUNCH(VV)                                                                                                      \
    do {                                                                                                                           \
        (void)hipFuncSetAttribute((const void *)k_kmeans_exact_pair4<KS, VV, nw4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); \
        hipLaunchKernelGGL((k_kmeans_exact_pair4<KS, VV, nw4>), dim3(grid4), dim3(64 * nw4), lds4, st, P, parts4, cpp4);            \
    } while (0)
            switch ((16 * KS - A.D) / 4) {
                case 0: SEGK_PAIR4_LAUNCH(0); break;
                case 1: SEGK_PAIR4_LAUNCH(1); break;
                case 2: SEGK_PAIR4_LAUNCH(2); break;
                default: SEGK_PAIR4_LAUNCH(3); break;
            }
#undef SEGK_PAIR4_LAUNCH
        } else {
            switch ((16 * KS - A.D) / 4) {
                case 0: hipLaunchKernelGGL((k_kmeans_exact_pair3<KS, 0>), dim3((unsigned)waves), dim3(64), lds_p, st, P); break;
                case 1: hipLaunchKernelGGL((k_kmeans_exact_pair3<KS, 1>), dim3((unsigned)waves), dim3(64), lds_p, st, P); break;
                case 2: hipLaunchKernelGGL((k_kmeans_exact_pair3<KS, 2>), dim3((unsigned)waves), dim3(64), lds_p, st, P); break;
                default: hipLaunchKernelGGL((k_kmeans_exact_pair3<KS, 3>), dim3((unsigned)waves), dim3(64), lds_p, st, P); break;
            }
        }
    };
    if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));
    if (n4 > 0) {
        ScoreArgs M = A;
        M.n = n4;
        const int ab = segk_dev_env("SEGK_H1_ABL");                  // development (-DSEGK_DEV builds), timing only
        if (ab == 1) hipLaunchKernelGGL((k_kmeans_score_h1<KS, 4, 1>), dim3((unsigned)(n4 / 512)), dim3(256), lds, st, M);
        else if (ab == 2) hipLaunchKernelGGL((k_kmeans_score_h1<KS, 4, 2>), dim3((unsigned)(n4 / 512)), dim3(256), lds, st, M);
        else if (ab == 3) hipLaunchKernelGGL((k_kmeans_score_h1<KS, 4, 3>), dim3((unsigned)(n4 / 512)), dim3(256), lds, st, M);
        else hipLaunchKernelGGL((k_kmeans_score_h1<KS, 4>), dim3((unsigned)(n4 / 512)), dim3(256), lds, st, M);
        if (int rc = prof_end(n4, 1)) return rc;
    }
    if (rem > 0 && !rem_queued) {
        hipLaunchKernelGGL((k_kmeans_score_h1<KS, 2>), dim3((unsigned)((rem + 255) / 256)), dim3(256), lds, st, T);
        if (n4 == 0)
            if (int rc = prof_end(rem, 1)) return rc;
    }
    // the decided rows' exact stage, then the undecided rows' second stage (all three products; its row count is read on
    // the device); the full scan of what that leaves follows in segk_kmeans_score
    launch_pair();
    {
        ScoreArgs B = A;
        B.ids = A.pre_queue;
        B.row0 = 0;
        B.n = cap2;
        B.n_dev = ctx->pre_queue;
        if (int rc = segk_launch_sp_second(ctx, B, KS, st)) return rc;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int segk_dispatch_score_pre(segk_ctx *ctx, const ScoreArgs &A, int ks, hipStream_t st)
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
