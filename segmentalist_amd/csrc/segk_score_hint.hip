// segk_score_hint.hip -- A1 with a HINT per row (round 3): the dense one-product fp16 contraction with a value-only top-2
// drain, and an exact stage that verifies the hinted component against it.
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
//
// KMeansComponents.argmax_neg_sqrd_norm_i (kmeans_components.py:225-232) is evaluated for every row in every sweep, and from
// one sweep to the next almost every row keeps its component (measured on the headline corpus: 6-17 % of the rows change
// in sweeps 2-10, 0.1 % once the chain has settled; tools/diag_hint_stability.py).  The one-product pre-filter
// (segk_score_h1.hip) spends as many vector-ALU issue cycles on its running top-2 -- five operations per two values, two
// of them only to remember WHICH pair won -- as its matrix pipe spends on the products, and pays a staging barrier per
// tile: 41 % matrix-pipe occupancy.  With a hint the index does not have to be tracked at all:
//
//   K1  k_kmeans_top2_rs   every (row, component) product on the matrix cores exactly as before (all K_max slots, nothing
//       skipped), but the drain keeps only the two largest VALUES per row: m1' = max3(m1, a, b), m2' = max(m2, med3(m1, a, b))
//       -- three operations per two values.  Range-stationary: a workgroup (8 waves, one per CU) copies the fp16 tile
//       images of ONE range of components into LDS once (16 tiles = 114 KB for the headline model) and its waves stream
//       row blocks past them without a single barrier; the constants -|m|^2/2 enter as the C operand of each block's
//       first MFMA.  Output: (m1, m2) per (row, range).
//   K2  k_kmeans_hint_exact   the component table in LDS as float32 (ranges, like k_kmeans_exact_pair4): for each row the
//       hinted component h is scored in the reference's arithmetic, s = -|x - m_h|^2, together with |x|^2 in the same
//       summation order, so that f_h = (s + |x|^2) / 2 = x.m_h - |m_h|^2/2 is known to within the reference's own
//       rounding.  With F the filter values (|F_k - f_k| <= E for every k), top1 >= top2 their two largest and
//       tau >= 2 E + E2 the pre-filter's margin (filter_tau_h1):
//           top1 - top2 > tau            =>  the filter's argmax k1 is the reference's argmax      (as in the pre-filter)
//           f_h >= top1 - tau + E + dl   =>  h = k1: any other k has F_k <= top2, so f_k <= top2 + E < top1 - tau + E
//       (dl bounds the error of the computed f_h).  Both hold -> cand.k = h, cand.s = s: the reference's bits.  Otherwise
//       the row is queued for the existing second stage (k_kmeans_score_sp, all three products) and, from there, the full
//       scan -- exactly the rows the pre-filter would have queued plus the rows whose hint was wrong.  A hint that names a
//       component the filters' images carry as "absent" (an exact duplicate of a lower row, segk_kmeans_mark_duplicates)
//       is no hint: its F is not a bound on anything.
//
// Results are those of segk_kmeans_score whatever the hints are (a wrong hint costs time, never correctness); the
// full-size parity tests run this path against the C oracle row by row.
#include "segk_kmeans_dev.h"

// cand.k in K2 (marked by k_hint_map): (previous label | SEGK_HINT_BIT) = a hint to be translated by the map; -1 = none.  K2's
// workgroups (one table range each) all scan every row: the mark tells a row that still waits for its range's workgroup from
// one that workgroup has already given its final label.
#define SEGK_HINT_BIT 0x20000000
#define SEGK_HINT_MAX_TPR 32         /* tiles per LDS range of K1 at most (its fill: one thread per float4 of the constants) */

// development, timing only (-DSEGK_K1_ABL=n, results wrong): 1 no drain in the tile loop, 2 no operand refill from LDS
#ifndef SEGK_K1_ABL
#define SEGK_K1_ABL 0
#endif

struct HintArgs {
    const unsigned char *ximg;      // fp16x2 row image (segk_corpus.Xb3): header, then plane 0 [n_emb][KP]
    const int32_t *ids;
    int64_t row0, n;
    const float *tiles;             // first tile of the fp16x2 tile image (tiles_b3 + 1024)
    int n_tiles, tpr, n_ranges;     // tiles per range, ranges
    float2 *part;                   // [n_ranges][n] (m1, m2) in the scaled domain of the images
    int K_max;
    int dbg;                        // development (SEGK_HINT_DBG, results wrong): 1 no result stores, 2 no hint loads / marks
    unsigned long long *stamp;      // development (-DSEGK_STAMP builds): per wave {cycles in the row waits, in the tile loops, total, groups}
    float *fb_w;                    // [3][8] share of the row groups each XCD took in the launches L - 1, L, L + 1 (slot = launch % 3)
    unsigned int *fb_t;             // [3][8] how long its waves lived (s_memrealtime ticks, maximum); NULL: equal shares
    int fb_cur;                     // this launch's slot
    const int64_t *fb_split;        // [9] this launch's first group per XCD (k_hint_map; unused with `own`)
    // own != 0: the kernel does k_hint_map's work itself (the label map straight into LDS, the XCDs' shares by every wave, the
    // queue counters by its first thread): one launch and one kernel boundary (~5 us) less per score call
    int own;
    const int32_t *remap;           // previous label -> current label (NULL: identity)
    int32_t *zero_cnt, *pre_hdr;    // the caller's ambiguity-queue length (may be NULL) and the undecided-row queue's header [16]
    int64_t total_groups;
    int64_t k1_groups;              // groups this kernel multiplies (the few behind the last whole round of all waves go to the second stage)
    // the hint waves (waves NW .. 2 NW - 1 of every workgroup): the hinted component of every row scored in reference arithmetic
    const float *xrows32;           // float32 rows [n_emb][ld32]
    int64_t ld32;
    const float *means32;           // float32 `means` [K_max][D]
    const int32_t *cand_k;          // per row: the label the previous call left (the hint, in that call's labelling)
    const int32_t *map;             // [K_max] previous label -> current label, or -1 (k_hint_map)
    const float *nxx;               // -|x|^2 per row in the reference's summation order (k_corpus_resid_sp)
    float4 *hint_out;               // [n] by position in the launch: {s = -|x - m_h|^2, f_h = x.m_h - |m_h|^2/2, bits of h (-1: no hint), 0}
};

// ---- K1 ---------------------------------------------------------------------------------------------------------------
// NW waves per workgroup, one workgroup per CU.  NW = 4: ONE wave per SIMD with the whole register file (512 per lane) --
// the rows of the wave's next group are prefetched into a second register set while the current group is multiplied, and
// nothing but the wave's own instruction stream decides whether the matrix pipe idles: per 32-cycle MFMA slot the MFMA's
// issue (8 cycles) and 3-4 vector operations of the other block's drain.  NW = 8: two waves per SIMD with 256 registers
// each (no prefetch), which fill each other's gaps at group boundaries but compete for the SIMD's vector issue inside the
// tile loop.
//
// Drain of a block's 16 values per lane, four at a time, value-only (no index): with x1 = max3(m1, a, b),
// t1 = med3(m1, a, b) [the second largest of m1, a, b], u = med3(x1, c, d) [the second largest of x1, c, d]:
//     m1' = max3(x1, c, d)     m2' = max3(m2, t1, u)
// (the second largest of {m1, a, b, c, d} is max(t1, u): t1 <= x1, and whichever of x1, c, d is largest, u is the runner-up
// among them) -- five operations per four values.  Operation 0 of a quad is a compiler-visible builtin, so that the hazard
// recogniser sees the first read of the MFMA's result; the others are single-instruction asm (fmaxf() would add a
// canonicalising v_max per MFMA output, and the scheduler may not reorder asm volatile).
// One quad = one asm block (the compiler pads every asm statement that a vector instruction of its own follows with an
// s_nop 0, four cycles of issue each: with one statement per operation the nops were a fifth of the loop's issue slots, and a
// lone wave has none to spare).  FIRST: the quad that opens a block's drain -- its first operation, the first read of the
// MFMA's result, stays a compiler-visible builtin so that the hazard recogniser places the wait states the matrix pipe needs.
#define SEGK_RS_DRAIN_QUAD(O_, AO, q_, FIRST)                                                                               \
    do {                                                                                                                     \
        float x1_, u_;                                                                                                       \
        if (FIRST) {                                                                                                         \
            const float t1_ = __builtin_amdgcn_fmed3f(m1[O_], AO[4 * (q_)], AO[4 * (q_) + 1]);                              \
            asm volatile("v_max3_f32 %2, %0, %4, %5\n\t"                                                                     \
                         "v_med3_f32 %3, %2, %6, %7\n\t"                                                                     \
                         "v_max3_f32 %0, %2, %6, %7\n\t"                                                                     \
                         "v_max3_f32 %1, %1, %8, %3"                                                                          \
                         : "+v"(m1[O_]), "+v"(m2[O_]), "=&v"(x1_), "=&v"(u_)                                                 \
                         : "v"(AO[4 * (q_)]), "v"(AO[4 * (q_) + 1]), "v"(AO[4 * (q_) + 2]), "v"(AO[4 * (q_) + 3]), "v"(t1_)); \
        } else {                                                                                                             \
            float t1_;                                                                                                       \
            asm volatile("v_med3_f32 %4, %0, %5, %6\n\t"                                                                     \
                         "v_max3_f32 %2, %0, %5, %6\n\t"                                                                     \
                         "v_med3_f32 %3, %2, %7, %8\n\t"                                                                     \
                         "v_max3_f32 %0, %2, %7, %8\n\t"                                                                     \
                         "v_max3_f32 %1, %1, %4, %3"                                                                          \
                         : "+v"(m1[O_]), "+v"(m2[O_]), "=&v"(x1_), "=&v"(u_), "=&v"(t1_)                                     \
                         : "v"(AO[4 * (q_)]), "v"(AO[4 * (q_) + 1]), "v"(AO[4 * (q_) + 2]), "v"(AO[4 * (q_) + 3]));         \
        }                                                                                                                    \
    } while (0)

// ---- the hint waves of K1 -----------------------------------------------------------------------------------------------
// K1's four matrix waves leave the chip's memory system nearly idle (224 bytes per row and 200 us: 1.2 TB/s) and every SIMD a
// second wave slot.  Four more waves per workgroup use both: they stream the float32 rows ONCE, contiguously (whole 128-byte
// lines: a step is 32 consecutive rows), fetch each row's hinted component from the float32 table (400 KB: L2-resident) and
// evaluate the reference's -|x - m_h|^2 in numpy's pairwise order -- TWO lanes per row (h = 0, 1: the lane halves of the eight
// strided accumulators), packed fp32 arithmetic with every half rounded like the scalar operation.  Nothing here depends on the
// matrix waves' results: the certificate (top-2 of ALL ranges against f_h) is taken afterwards by k_hint_merge, one pass over
// 48 bytes per row.  Round 3 did this as a kernel of its own behind K1 (k_kmeans_hint_exact: 129 us, 572 MB fetched because
// its range workgroups picked scattered rows); fused, the float32 corpus crosses HBM once per sweep, under the matrix work.
template <int KS, int V>
__device__ __forceinline__ void hint_wave_rows(const HintArgs &H, const int32_t *map /* LDS */, int64_t w, int64_t n_w)
{
    constexpr int D = 16 * KS - 4 * V;
    constexpr int nfull = D & ~7, nblk = nfull >> 3, rem = D & 7;
    constexpr int NX = nblk + (rem ? 1 : 0);
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) f32x4_t *gptr_t;
    const int lane = threadIdx.x & 63, row = lane >> 1, h = lane & 1;
    auto pk_sub = [](f32x2_t a, f32x2_t b2) -> f32x2_t {           // a - b, both halves in one instruction
        f32x2_t d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b2));
        return d;
    };
    const int64_t n_steps = (H.n + 31) >> 5;
    if (H.dbg & 8) return;                                         // development (make DEV=1): timing without the hint waves' work
    auto rid_of = [&](int64_t s_) -> int32_t {
        const int64_t p_ = s_ * 32 + row;
        return p_ < H.n ? (H.ids ? H.ids[p_] : (int32_t)(H.row0 + p_)) : -1;
    };
    int64_t s = w;
    // row ids two steps ahead, previous labels one step ahead: no load of a step waits for another load of the same step
    int32_t rid = -1, rid_n = -1, kprev = -1;
    if (s < n_steps) {
        rid = rid_of(s);
        if (s + n_w < n_steps) rid_n = rid_of(s + n_w);
        kprev = rid >= 0 ? H.cand_k[rid] : -1;
    }
    for (; s < n_steps; s += n_w) {
        const int32_t hint = (rid >= 0 && kprev >= 0 && kprev < H.K_max) ? map[kprev] : -1;
        const int32_t rid_c = rid;
        f32x4_t xv[NX], mv[NX];
        {
            int64_t r_any = rid_c >= 0 ? (int64_t)rid_c : (H.ids ? 0 : H.row0);
            if (H.dbg & 16) r_any &= 1023;                          // development: the rows from a cache-resident corner (timing)
            const uintptr_t xa = (uintptr_t)(H.xrows32 + r_any * H.ld32);
            const uintptr_t ma = (uintptr_t)(H.means32 + (int64_t)(hint >= 0 ? hint : 0) * D);
#pragma unroll
            for (int b = 0; b < nblk; b++) xv[b] = *reinterpret_cast<gptr_t>(xa + 16u * h + 32u * b);
            if constexpr (rem != 0) xv[nblk] = *reinterpret_cast<gptr_t>(xa + 4u * nfull);
#pragma unroll
            for (int b = 0; b < nblk; b++) mv[b] = *reinterpret_cast<gptr_t>(ma + 16u * h + 32u * b);
            if constexpr (rem != 0) mv[nblk] = *reinterpret_cast<gptr_t>(ma + 4u * nfull);
        }
        const float nx = rid_c >= 0 ? H.nxx[rid_c] : 0.f;
        if (H.dbg & 64) __builtin_amdgcn_s_sleep(64);               // development: paced hint waves
        // the next step's previous labels and the row ids of the step after it travel under this step's arithmetic
        {
            rid = rid_n;
            kprev = rid >= 0 ? H.cand_k[rid] : -1;
            const int64_t s2 = s + 2 * n_w;
            rid_n = s2 < n_steps ? rid_of(s2) : -1;
        }
        // the reference's float32 -(deltas*deltas).sum() in numpy's pairwise order: this lane owns the strided accumulators
        // r_{4h..4h+3}
        f32x2_t rl = {0.f, 0.f}, rh = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < ((H.dbg & 32) ? 1 : nblk); b++) {      // (dbg 32, development: one block of the arithmetic only)
            const f32x2_t dl = pk_sub(mv[b].xy, xv[b].xy), dh = pk_sub(mv[b].zw, xv[b].zw);
            const f32x2_t tl = dl * dl, th = dh * dh;
            rl = b == 0 ? tl : rl + tl;
            rh = b == 0 ? th : rh + th;
        }
        float res = (rl.x + rl.y) + (rh.x + rh.y);
        const float ro = __shfl_xor(res, 1);
        res = (h == 0) ? res + ro : ro + res;                      // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
        if constexpr (rem != 0) {
            const f32x2_t dl = pk_sub(mv[nblk].xy, xv[nblk].xy), dh = pk_sub(mv[nblk].zw, xv[nblk].zw);
            const f32x2_t tl = dl * dl, th = dh * dh;
            res += tl.x;
            if (rem > 1) res += tl.y;
            if (rem > 2) res += th.x;
            if (rem > 3) res += th.y;
        }
        const float sc = -res;                                     // -|x - m_h|^2
        const int64_t p = s * 32 + row;
        if (h == 0 && p < H.n) H.hint_out[p] = make_float4(sc, 0.5f * (sc - nx), __int_as_float(rid_c >= 0 ? hint : -1), 0.f);
    }
}

// (launch bounds "two waves per SIMD" for both: 256 registers per lane, all of them vector registers.  Given 512 the compiler
// keeps the accumulators in the accumulator file and copies every value out for the drain, 16 v_accvgpr_read per block)
template <int KS, int V, int NW>
__global__ __launch_bounds__(128 * NW, 1) void k_kmeans_top2_rs(HintArgs H)
{
    static_assert(NW == 4, "four matrix waves (one per SIMD, rows prefetched) + four hint waves per workgroup");
    typedef _Float16 T;
    typedef SegkPiece<2>::V8 V8;
    // (only 256 of a lone wave's 512 registers are addressable by vector instructions, the rest is the accumulator file: a
    // second 4-block row set lands there and is copied back and forth -- 5 000 v_accvgpr moves.  So: two blocks per group with
    // prefetch for NW = 4, four without for NW = 8)
    constexpr int P = 2, KP = KS * 16, NBLK = NW == 4 ? 2 : 4;
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;      // floats per tile of the global image
    constexpr int TL = KS * 256 + 32;                                     // floats per tile in LDS: KS piece-0 blocks + constants
    constexpr bool PREFETCH = NW == 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int R = H.n_ranges;
    const bool is_hint = wave >= NW;                // waves NW .. 2 NW - 1: hint_wave_rows
    if (!is_hint) __builtin_amdgcn_s_setprio(2);    // the matrix waves first wherever the two kinds meet at an issue port
#ifdef SEGK_STAMP
    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // workgroup -> (range, slot).  Workgroups b and b + 8 share an XCD (round-robin placement, speed only): the R
    // workgroups that stream the same rows sit on one XCD when the grid allows, so that the rows cross HBM once
    int range, wgr, n_wgr;
    const bool xcd_aware = (gridDim.x & 7) == 0 && ((gridDim.x >> 3) % R) == 0;
    if (xcd_aware) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        range = idx % R;
        wgr = (idx / R) * 8 + xcd;
        n_wgr = gridDim.x / R;
    } else {
        n_wgr = gridDim.x / R;
        range = blockIdx.x % R;
        wgr = blockIdx.x / R;
    }
    const int t_lo = range * H.tpr;
    int nt = H.n_tiles - t_lo;
    if (nt > H.tpr) nt = H.tpr;
    // (a workgroup without matrix work -- beyond the last whole set of ranges, or a range without tiles -- still runs its hint waves)
    const bool mm_on = wgr < n_wgr && nt > 0;
    if (nt < 1) nt = 1;
    const T *plane0 = (const T *)(H.ximg + SEGK_SP_HEADER);
    // this wave's row groups: g_first, g_first + n_slots, ... below n_groups
    const int64_t total_groups = (H.n + 32 * NBLK - 1) / (32 * NBLK);
    int64_t n_groups = H.k1_groups < total_groups ? H.k1_groups : total_groups, n_slots = (int64_t)n_wgr * NW, g_first = (int64_t)wgr * NW + wave;
    // Under this kernel the chip runs into its power limit, and the eight XCDs then hold DIFFERENT clocks (1.65-1.77 GHz
    // measured, the same XCDs slow launch after launch): with equal shares the fast ones idle for the last 10-20 us of 200.
    // So an XCD takes a contiguous share of the groups in proportion to the rate it showed in the previous launch (its share
    // then / the lifetime of its waves, half-way blended; k_hint_map, the small launch in front, does the arithmetic): nothing is
    // exchanged during the launch, and the results do not depend on who computes which rows.
    const unsigned long long fb_t0 = __builtin_amdgcn_s_memrealtime();
    const bool fb = xcd_aware && H.fb_t != nullptr;
    int64_t own_lo = 0, own_hi = 0;
    if (H.own) {
        if (blockIdx.x == 0 && tid == 0) {
            if (H.zero_cnt) *H.zero_cnt = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) H.pre_hdr[i] = 0;
        }
        // the XCDs' shares for this launch (k_hint_map's arithmetic, by every wave: one load round trip, sums and prefix by
        // shuffles; lane x & 7 = XCD x): share of the previous launch / lifetime of its waves = the rate an XCD showed; new
        // share = half the old one, half the rate's
        const int x = lane & 7;
        const int cur = H.fb_cur, prev = (cur + 2) % 3, next = (cur + 1) % 3;
        const float wp = H.fb_w[prev * 8 + x];
        const unsigned int tp = H.fb_t ? H.fb_t[prev * 8 + x] : 0u;
        const bool ok = __all(tp > 0u && wp > 0.f);
        const float rate = ok ? wp / (float)tp : 0.f;
        float rsum = rate;
        rsum += __shfl_xor(rsum, 1);
        rsum += __shfl_xor(rsum, 2);
        rsum += __shfl_xor(rsum, 4);
        float w = !(wp > 0.f) ? 0.125f : ok ? 0.5f * wp + 0.5f * (rate / rsum) : wp;
        w = fminf(fmaxf(w, 0.0625f), 0.25f);
        float wsum = w;
        wsum += __shfl_xor(wsum, 1);
        wsum += __shfl_xor(wsum, 2);
        wsum += __shfl_xor(wsum, 4);
        double cum = 0.0;                                        // exclusive prefix in XCD order, every lane the same additions
        for (int y = 0; y < 8; y++) {
            const float wy = __shfl(w, y);
            if (y < x) cum += (double)wy;
        }
        const int64_t split = (int64_t)(cum * ((double)H.total_groups / (double)wsum));
        const int xcd = blockIdx.x & 7;
        own_lo = __shfl(split, xcd);
        own_hi = xcd == 7 ? H.total_groups : __shfl(split, (xcd + 1) & 7);
        if (blockIdx.x == 0 && wave == 0 && lane < 8) {
            H.fb_w[cur * 8 + x] = w / wsum;
            if (H.fb_t) H.fb_t[next * 8 + x] = 0u;
        }
    }
    if (fb) {
        const int xcd = blockIdx.x & 7;
        const int64_t lo = H.own ? own_lo : H.fb_split[xcd], hi = H.own ? own_hi : H.fb_split[xcd + 1];
        n_slots = (int64_t)(n_wgr >> 3) * NW;
        g_first = lo + (int64_t)(wgr >> 3) * NW + wave;
        n_groups = hi;
    }

    // the rows of group g into a register set
#define SEGK_RS_LOAD(g_, XB, HROW, HK)                                                                          \
    do {                                                                                                         \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                                       \
            const int64_t r = (g_) * (32 * NBLK) + 32 * b + j;                                                   \
            int64_t rowid = -1;                                                                                  \
            if (r < H.n) rowid = H.ids ? (int64_t)H.ids[r] : H.row0 + r;                                         \
            HROW[b] = -1;                                                                                        \
            if (rowid < 0) rowid = 0;          /* a skipped entry of the id list: some valid row, result unused */ \
            const T *xp = plane0 + rowid * KP + 8 * h;                                                           \
            _Pragma("unroll") for (int s = 0; s < KS; s++) XB[b][s] = *reinterpret_cast<const V8 *>(xp + 16 * s); \
        }                                                                                                        \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) HK[b] = -1;                                             \
    } while (0)

    // MFMAs of block N_ (accumulator AN) over the drain of block O_'s values (accumulator AO).  The twenty operations of the
    // drain are spread over the MFMAs 1 .. KS-1: block O_'s last MFMA was issued just before this unit's first one, and a
    // drain right behind that one would wait out the matrix pipe's latency.
    // The MFMA intrinsic has no side effects, so neither volatile asm nor sched_barrier orders it (instruction selection sinks
    // it towards its use, behind the drain): two empty asm statements pin it by DATA dependence -- its A operand passes
    // through the first, its result through the second.
#define SEGK_RS_UNIT(XB, N_, AN, O_, AO, REFILL)                                                                     \
    do {                                                                                                              \
        _Pragma("unroll") for (int s = 0; s < KS; s++) {                                                              \
            asm volatile("" : "+v"(a[s]));                                                                            \
            AN = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], XB[N_][s], s == 0 ? cs : AN, 0, 0, 0);                  \
            asm volatile("" : "+v"(AN));                                                                              \
            if (REFILL && !(SEGK_K1_ABL & 2)) {   /* this tile is done with a[s] (and, after its first MFMA, with cs) */ \
                load_a(tn, s);                                                                                        \
                if (s == 0) load_cs(tn);                                                                              \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            /* quads q_lo .. q_hi-1 of block O_ behind this MFMA: the four quads over the slots 1 .. KS-1 */          \
            constexpr int SL = KS > 1 ? KS - 1 : 1;                                                                   \
            const int q_lo = KS > 1 ? ((s - 1) * 4 + SL - 1) / SL : 0, q_hi = KS > 1 ? (s * 4 + SL - 1) / SL : 4;     \
            if ((KS == 1 || s >= 1) && !(SEGK_K1_ABL & 1)) {                                                          \
                if (0 >= q_lo && 0 < q_hi) SEGK_RS_DRAIN_QUAD(O_, AO, 0, true);                                       \
                if (1 >= q_lo && 1 < q_hi) SEGK_RS_DRAIN_QUAD(O_, AO, 1, false);                                      \
                if (2 >= q_lo && 2 < q_hi) SEGK_RS_DRAIN_QUAD(O_, AO, 2, false);                                      \
                if (3 >= q_lo && 3 < q_hi) SEGK_RS_DRAIN_QUAD(O_, AO, 3, false);                                      \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    } while (0)

    // (two tiles per trip: with one, the accumulator of block 3 and the constants of the next tile change registers
    // across the back edge -- 24 moves and an s_nop 11 per tile)
#define SEGK_RS_TILE(XB, t_)                                        \
    do {                                                            \
        const int tn = (t_) + 1 < nt ? (t_) + 1 : 0;     /* the last tile refills tile 0's operands: the next group's */ \
        if constexpr (NBLK == 4) {                                  \
            SEGK_RS_UNIT(XB, 0, acc0, 3, acc1, false);              \
            SEGK_RS_UNIT(XB, 1, acc1, 0, acc0, false);              \
            SEGK_RS_UNIT(XB, 2, acc0, 1, acc1, false);              \
            SEGK_RS_UNIT(XB, 3, acc1, 2, acc0, true);               \
        } else {                                                    \
            SEGK_RS_UNIT(XB, 0, acc0, 1, acc1, false);              \
            SEGK_RS_UNIT(XB, 1, acc1, 0, acc0, true);               \
        }                                                           \
    } while (0)

    // one group: all the range's tiles against the rows in XB, then the (m1, m2) of its rows and the marks of their hints
#define SEGK_RS_GROUP(g_, XB, HROW, HK)                                                                                   \
    do {                                                                                                                   \
        float m1[NBLK], m2[NBLK];                                                                                          \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) { m1[b] = NEG_INF_F; m2[b] = NEG_INF_F; }                         \
        f32x16 acc0, acc1;                                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 16; q++) acc1[q] = NEG_INF_F;        /* "last block of tile -1": drains to nothing */ \
        int t = 0;                                                                                                         \
        for (; t + 1 < nt; t += 2) {                                                                                       \
            SEGK_RS_TILE(XB, t);                                                                                           \
            SEGK_RS_TILE(XB, t + 1);                                                                                       \
        }                                                                                                                  \
        if (t < nt) SEGK_RS_TILE(XB, t);                                                                                   \
        SEGK_RS_DRAIN_QUAD(NBLK - 1, acc1, 0, true);                                                                       \
        SEGK_RS_DRAIN_QUAD(NBLK - 1, acc1, 1, false);                                                                      \
        SEGK_RS_DRAIN_QUAD(NBLK - 1, acc1, 2, false);                                                                      \
        SEGK_RS_DRAIN_QUAD(NBLK - 1, acc1, 3, false);                                                                      \
        /* the two lane halves of a row hold 16 components of every tile each: merge; stored at the start of the next */   \
        /* group (SEGK_RS_STORE), behind that group's prefetch: a store issued here would sit in front of the loads in */  \
        /* the in-order vmcnt queue and every wait for rows would wait out its write acknowledge as well              */  \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                                                 \
            const float o1 = __shfl_xor(m1[b], 32), o2 = __shfl_xor(m2[b], 32);                                            \
            pend1[b] = fmaxf(m1[b], o1);                                                                                   \
            pend2[b] = fmaxf(fminf(m1[b], o1), fmaxf(m2[b], o2));                                                          \
            pend_row[b] = HROW[b];                                                                                         \
            pend_k[b] = HK[b];                                                                                             \
        }                                                                                                                  \
        pend_g = (g_);                                                                                                     \
    } while (0)

    // results of the group before: (m1, m2) of its rows (lane half 0)
#define SEGK_RS_STORE()                                                                                                    \
    do {                                                                                                                   \
        if (pend_g >= 0) {                                                                                                 \
            _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                                             \
                const int64_t r = pend_g * (32 * NBLK) + 32 * b + j;                                                       \
                if (h == 0 && r < H.n && !(H.dbg & 1)) H.part[(int64_t)range * H.n + r] = make_float2(pend1[b], pend2[b]); \
                (void)pend_row[b]; (void)pend_k[b];                                                                        \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)

    V8 a[KS];
    f32x16 cs;
    auto load_a = [&](int t, int s) { a[s] = *reinterpret_cast<const V8 *>((const T *)(lds + t * TL) + (s * 64 + lane) * 8); };
    auto load_cs = [&](int t) {
        const float *cv = lds + t * TL + KS * 256 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 c4 = *reinterpret_cast<const float4 *>(cv + 8 * q);
            cs[4 * q + 0] = c4.x; cs[4 * q + 1] = c4.y; cs[4 * q + 2] = c4.z; cs[4 * q + 3] = c4.w;
        }
    };
    int64_t g = g_first;
    float pend1[NBLK], pend2[NBLK];
    int32_t pend_row[NBLK], pend_k[NBLK];
    int64_t pend_g = -1;
    V8 xa[NBLK][KS];
    int32_t hrow_a[NBLK], hk_a[NBLK];
    if (!is_hint && mm_on && g < n_groups && !(H.dbg & 4)) SEGK_RS_LOAD(g, xa, hrow_a, hk_a);      // the first rows travel while the tile images are copied
#ifdef SEGK_STAMP
    const unsigned long long st_k1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- the range's tile images into LDS, once.  The unit of the copy is one 1 KiB piece-0 block (a k-step of a tile:
    // 64 lanes x 16 bytes, contiguous in both images): wave w takes the blocks w, w + NW, ...; source and destination of a
    // block are wave-uniform (scalar arithmetic), and ALL of a wave's loads -- 28 for the headline model -- are in flight
    // together: one round trip for the 114 KB.  (Element-wise with a division per 16 bytes and 16 loads in flight per thread
    // the fill took 21 700 cycles, 12 us of a 200 us kernel -- and a third of a 1 250-utterance shard's.)
    {
        constexpr int MAXT = (160 * 1024 / (TL * 4)) < SEGK_HINT_MAX_TPR ? (160 * 1024 / (TL * 4)) : SEGK_HINT_MAX_TPR;
        constexpr int MAXB = MAXT * KS;                                 // blocks of the largest range (LDS, SEGK_HINT_MAX_TPR)
        constexpr int NWF = 2 * NW;                                     // all eight waves copy
        constexpr int PER_W = (MAXB + NWF - 1) / NWF;
        const int n_blk = nt * KS;                                      // (without matrix work: tile 0's blocks are loaded and dropped)
        const int t_ld = mm_on ? t_lo : 0;
        // every workgroup of a range copies the same bytes at the same moment: started at the same block they all queue on
        // the same L2 channel (5.8 bytes per cycle and CU measured).  Each starts somewhere else in the range instead.
        const int rot = (int)(((unsigned)wgr * 2654435761u) >> 8) % n_blk;
        float4 v[PER_W];
#pragma unroll
        for (int u = 0; u < PER_W; u++) {
            int c = wave + u * NWF;
            if (c >= n_blk) c = n_blk - 1;                              // clamped, unconditional load
            c += rot;
            if (c >= n_blk) c -= n_blk;
            const int t = c / KS, ks = c - t * KS;
            v[u] = *reinterpret_cast<const float4 *>(H.tiles + (int64_t)(t_ld + t) * STRIDE + ks * (P * 256) + lane * 4);
        }
        float4 cv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool has_c = mm_on && tid < nt * 8;                       // the 32 constants of every tile: one float4 per thread
        if (has_c) cv4 = *reinterpret_cast<const float4 *>(H.tiles + (int64_t)(t_lo + (tid >> 3)) * STRIDE + KS * P * 256 + (tid & 7) * 4);
        // the label map of the hint waves behind the tile images
        int32_t *map_l = reinterpret_cast<int32_t *>(lds + H.tpr * TL);
        if (H.own) {
            // the label a hint of the previous call stands for now (the relabelling of clean_components), or -1 when the
            // filters' images carry that component as absent (a marked duplicate: such a hint proves nothing)
            for (int k = tid; k < H.K_max; k += 128 * NW) {
                int v = H.remap ? H.remap[k] : k;
                if (v < 0 || v >= H.K_max) v = -1;
                else if (H.tiles[(int64_t)(v >> 5) * STRIDE + KS * P * 256 + (v & 31)] < -1.0e37f) v = -1;
                map_l[k] = v;
            }
        } else {
            for (int k = tid; k < H.K_max; k += 128 * NW) map_l[k] = H.map[k];
        }
#pragma unroll
        for (int u = 0; u < PER_W; u++) {
            int c = wave + u * NWF;
            if (mm_on && c < n_blk) {
                c += rot;
                if (c >= n_blk) c -= n_blk;
                const int t = c / KS, ks = c - t * KS;
                *reinterpret_cast<float4 *>(lds + t * TL + ks * 256 + lane * 4) = v[u];
            }
        }
        if (has_c) *reinterpret_cast<float4 *>(lds + (tid >> 3) * TL + KS * 256 + (tid & 7) * 4) = cv4;
        static_assert(128 * NW >= MAXT * 8, "one thread per float4 of the constants");
    }
    __syncthreads();
    if (is_hint) {
        // every workgroup's hint waves take steps of 32 rows, strided over the whole grid: the chip walks the corpus front to back
        hint_wave_rows<KS, V>(H, reinterpret_cast<const int32_t *>(lds + H.tpr * TL), (int64_t)blockIdx.x * NW + (wave - NW),
                              (int64_t)gridDim.x * NW);
        return;
    }
#ifdef SEGK_STAMP
    const unsigned long long st_k2 = __builtin_amdgcn_s_memtime();
#endif
    if (!mm_on || g >= n_groups || (H.dbg & 128)) return;       // (dbg 128, make DEV=1: timing of the hint waves alone)
    if (H.dbg & 4) SEGK_RS_LOAD(g, xa, hrow_a, hk_a);
#pragma unroll
    for (int s = 0; s < KS; s++) load_a(0, s);          // tile 0's operands for the first group; every group's last tile reloads them
    load_cs(0);
    if constexpr (PREFETCH) {
        V8 xb[NBLK][KS];
        int32_t hrow_b[NBLK], hk_b[NBLK];
        // The explicit waits (the builtin, which the compiler's wait-count pass understands; an asm wait it would not) tell
        // it that the current set has landed BEFORE the other set's loads are issued: left to itself it waits for the
        // current set inside the tile loop with counted vmcnt, which -- the counter being in issue order -- waits out the
        // prefetch too.  What is outstanding at such a wait was issued a whole group earlier (the rows, the stores of the
        // group before).
#ifdef SEGK_STAMP
        unsigned long long st_wait = 0, st_loop = 0, st_groups = 0;
        const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#define SEGK_ST(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define SEGK_ST(var) do { } while (0)
#endif
        for (;;) {
            SEGK_ST(s0);
            __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): the rows of group g are in xa
            SEGK_ST(s1);
            const int64_t g1 = g + n_slots;
            if (g1 < n_groups) SEGK_RS_LOAD(g1, xb, hrow_b, hk_b);          // in flight under this group's tile loop
            SEGK_RS_STORE();
            SEGK_ST(s2);
            SEGK_RS_GROUP(g, xa, hrow_a, hk_a);
            SEGK_ST(s3);
#ifdef SEGK_STAMP
            st_wait += s1 - s0; st_loop += s3 - s2; st_groups++;
#endif
            if (g1 >= n_groups) break;
            SEGK_ST(s4);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            SEGK_ST(s5);
            g = g1 + n_slots;
            if (g < n_groups) SEGK_RS_LOAD(g, xa, hrow_a, hk_a);
            SEGK_RS_STORE();
            SEGK_ST(s6);
            SEGK_RS_GROUP(g1, xb, hrow_b, hk_b);
            SEGK_ST(s7);
#ifdef SEGK_STAMP
            st_wait += s5 - s4; st_loop += s7 - s6; st_groups++;
#endif
            if (g >= n_groups) break;
        }
#ifdef SEGK_STAMP
        if (H.stamp && lane == 0) {
            unsigned long long *o = H.stamp + ((int64_t)blockIdx.x * NW + wave) * 8;
            o[0] = st_wait; o[1] = st_loop; o[2] = __builtin_amdgcn_s_memtime() - st_begin; o[3] = st_groups;
            o[4] = st_begin - st_k0; o[5] = st_r0; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = ((st_k1 - st_k0) << 32) | (st_k2 - st_k1);
        }
#endif
#undef SEGK_ST
    } else {
        for (;;) {
            SEGK_RS_GROUP(g, xa, hrow_a, hk_a);
            const int64_t gp = pend_g;
            g += n_slots;
            if (g < n_groups) SEGK_RS_LOAD(g, xa, hrow_a, hk_a);            // the next rows first, the stores behind them
            pend_g = gp;
            SEGK_RS_STORE();
            pend_g = -1;
            if (g >= n_groups) break;
        }
    }
    SEGK_RS_STORE();
    if (fb && lane == 0) atomicMax(&H.fb_t[H.fb_cur * 8 + (blockIdx.x & 7)], (unsigned int)(__builtin_amdgcn_s_memrealtime() - fb_t0));
#undef SEGK_RS_STORE
#undef SEGK_RS_GROUP
#undef SEGK_RS_TILE
#undef SEGK_RS_UNIT
#undef SEGK_RS_LOAD
}
#undef SEGK_RS_DRAIN_QUAD

// One small launch in front of K1:
//   map[k] = the label a hint k of the previous call stands for now (remap, identity when NULL), or -1 when that component
//            is carried as "absent" by the filters' images (seed constant <= -1e37: a marked duplicate) -- such a hint
//            proves nothing;
//   the queue lengths of the call cleared: the caller's ambiguity queue (when segk_kmeans_score_hinted deferred it) and the
//   second stage's counters; the XCDs' shares of K1's row groups; (m1, m2) = (0, 0) for the rows K1 leaves out.
__global__ void k_hint_map(const int32_t *remap, const float *tiles_sp /* first tile */, int K_max, int stride, int const_off, int32_t *map,
                           int64_t n, int32_t *zero_cnt, int32_t *pre_hdr,
                           float *fb_w, unsigned int *fb_t, int fb_cur, int64_t *fb_split, int64_t total_groups, int64_t first_skipped,
                           float2 *part, int n_ranges)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0 && zero_cnt) *zero_cnt = 0;
        if (threadIdx.x < 16) pre_hdr[threadIdx.x] = 0;
        // the XCDs' shares of the matrix kernel's row groups for the launch behind this one (see k_kmeans_top2_rs): share of
        // the previous launch / lifetime of its waves = the rate an XCD showed; new share = half the old one, half the rate's
        if (threadIdx.x < 64 && fb_split) {                     // lane x < 8 = XCD x (one load round trip, sums and prefix by shuffles)
            const int x = threadIdx.x & 7;
            const bool mine = threadIdx.x < 8;
            const int cur = fb_cur, prev = (cur + 2) % 3, next = (cur + 1) % 3;
            const float wp = fb_w[prev * 8 + x];
            const unsigned int tp = fb_t ? fb_t[prev * 8 + x] : 0u;
            const bool ok = __all(tp > 0u && wp > 0.f);           // (lanes 8.. repeat lanes 0..7)
            const float rate = ok ? wp / (float)tp : 0.f;
            float rsum = rate;
            rsum += __shfl_xor(rsum, 1);
            rsum += __shfl_xor(rsum, 2);
            rsum += __shfl_xor(rsum, 4);
            float w = !(wp > 0.f) ? 0.125f : ok ? 0.5f * wp + 0.5f * (rate / rsum) : wp;
            w = fminf(fmaxf(w, 0.0625f), 0.25f);
            float wsum = w;
            wsum += __shfl_xor(wsum, 1);
            wsum += __shfl_xor(wsum, 2);
            wsum += __shfl_xor(wsum, 4);
            // exclusive prefix over the eight lanes, added in XCD order (every lane the same sequence of additions)
            double cum = 0.0;
            for (int y = 0; y < 8; y++) {
                const float wy = __shfl(w, y);
                if (y < x) cum += (double)wy;
            }
            if (mine) {
                fb_split[x] = (int64_t)(cum * ((double)total_groups / (double)wsum));
                fb_w[cur * 8 + x] = w / wsum;
                if (fb_t) fb_t[next * 8 + x] = 0u;
                if (x == 0) fb_split[8] = total_groups;
            }
        }
    }
    // rows the matrix kernel leaves out (the groups behind the last whole round of all its waves, when they are few): (m1, m2) =
    // (0, 0) in every range reads as "undecided" to the merge, which queues them for the second stage
    if (first_skipped + i < n)
        for (int rg = 0; rg < n_ranges; rg++) part[(int64_t)rg * n + first_skipped + i] = make_float2(0.f, 0.f);
    if (i < K_max) {
        int v = remap ? remap[i] : (int)i;
        if (v < 0 || v >= K_max) v = -1;
        else if (tiles_sp[(int64_t)(v >> 5) * stride + const_off + (v & 31)] < -1.0e37f) v = -1;
        map[i] = v;
    }
}

struct HintMergeArgs {
    const float2 *part;             // K1's matrix waves: (m1, m2) per (range, position)
    const float4 *hint_out;         // K1's hint waves: {s, f_h, bits of h, 0} per position
    int n_ranges;
    const float *tiles_hdr;         // tiles_b3: [0] exponent b, [1] E_m
    const unsigned char *ximg;      // row image header: [1] exponent a
};

// K2 (round 4): the certificate.  Per row the filter's top-2 merged over the ranges, the hinted component's exact score s and its
// filter-domain value f_h (see the head of the file):
//     top1 - top2 > tau   and   f_h >= top1 - tau + E + dl      =>   cand.k = h, cand.s = s  (the reference's bits)
// anything else -- no hint, a wrong hint, a near-tie -- is queued for the second stage (one reservation per wave).  One thread
// per row, 48 bytes read and 12 written: the whole exact stage of round 3 (k_kmeans_hint_exact, 129 us) shrunk to this pass,
// its arithmetic moved under K1's matrix work.
#define SEGK_MERGE_ROWS 6144        /* rows per workgroup at most (its list of undecided rows in LDS) */
#define SEGK_MERGE_THREADS 1024
__global__ __launch_bounds__(SEGK_MERGE_THREADS) void k_hint_merge(ScoreArgs A, HintMergeArgs H, int KP, int64_t per, float *pre_thr, int64_t first_skipped)
{
    // ONE queue reservation per workgroup, one workgroup per CU: returning atomics on one address are served one after the other,
    // ~11 ns each (a first version with one per wave -- 15 600 of them -- took 185 us for 55 MB of traffic, 1 024 workgroups
    // still 25 us); the workgroup's undecided rows wait in LDS
    __shared__ int32_t ulist[SEGK_MERGE_ROWS];
    __shared__ float uthr[SEGK_MERGE_ROWS];         // per undecided row: top1 - tau in the scaled domain (the band stage's threshold)
    __shared__ int32_t ucnt, ubase;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) ucnt = 0;
    __syncthreads();
    const int64_t p_lo = (int64_t)blockIdx.x * per, p_hi = p_lo + per < A.n ? p_lo + per : A.n;
    const int e_ab = ((const int *)H.ximg)[1] + ((const int *)H.tiles_hdr)[0];
    const float unscale = ldexpf(1.f, -e_ab), scale = ldexpf(1.f, e_ab);
    const float M = (float)(sqrt(*A.mnorm2) * (1.0 + 1e-6)) + 1e-30f;
    const float Em = H.tiles_hdr[1];
    // five rows per thread and trip, their loads in flight together (one row per trip was four dependent round trips per
    // workgroup: 25 us for 55 MB; with four the headline corpus -- 4 102 rows per workgroup -- took a second trip for six rows)
    constexpr int U = 5;
    for (int64_t p0 = p_lo; p0 < p_hi; p0 += SEGK_MERGE_THREADS * U) {
        int32_t rid[U];
        float4 ho[U];
        float t1[U], t2[U], xnb[U], xer[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int64_t p = p0 + j * SEGK_MERGE_THREADS + tid;
            rid[j] = -1;
            if (p < p_hi) rid[j] = A.ids ? A.ids[p] : (int32_t)(A.row0 + p);
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int64_t p = p0 + j * SEGK_MERGE_THREADS + tid;
            ho[j] = make_float4(0.f, 0.f, __int_as_float(-1), 0.f);
            t1[j] = NEG_INF_F; t2[j] = NEG_INF_F; xnb[j] = 0.f; xer[j] = 0.f;
            if (rid[j] >= 0) {
                ho[j] = H.hint_out[p];
                for (int r = 0; r < H.n_ranges; r++) {
                    const float2 pv = H.part[(int64_t)r * A.n + p];
                    const float n1 = fmaxf(t1[j], pv.x);
                    t2[j] = fmaxf(fminf(t1[j], pv.x), fmaxf(t2[j], pv.y));
                    t1[j] = n1;
                }
                xnb[j] = A.xnorm[rid[j]];
                xer[j] = A.xerr[rid[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            bool und = false;
            float thr = 0.f;
            if (rid[j] >= 0 && p0 + j * SEGK_MERGE_THREADS + tid >= first_skipped) {
                // a row K1 left out (the few groups behind the last whole round of its waves): no filter values, full scan
                const int q2 = atomicAdd(A.cand.count, 1);
                if (q2 < A.amb_cap) A.cand.queue[q2] = rid[j];
            } else if (rid[j] >= 0) {
                const int32_t hint = __float_as_int(ho[j].z);
                und = true;
                const float tau = filter_tau_h1(xnb[j], M, A.D, xer[j], Em);
                // the band the reference's argmax lies in: F >= top1 - tau (scaled domain; the subtraction's own rounding and a
                // little more taken off)
                thr = t1[j] - tau * scale * 1.000001f;
                thr -= 4e-7f * fabsf(t1[j]);
                if (hint >= 0) {
                    const float top1 = t1[j] * unscale, top2 = t2[j] * unscale;          // powers of two: exact
                    const float u = 5.9604645e-8f;
                    // E: bound of |F_k - f_k| (accumulation + operand rounding, the terms of tau); dl: of the computed f_h
                    const float e1 = (1.02f * (float)(KP + 16) + 16.f) * u * (xnb[j] * M + 0.5f * M * M);
                    const float rnd = 1.00001f * fminf((xnb[j] + xer[j]) * Em + xer[j] * M, 1.01f * 9.765625e-4f * xnb[j] * M);
                    const float s2 = xnb[j] + M;
                    const float dl = ((float)(A.D / 8 + 13) + 4.f) * u * s2 * s2;
                    const bool ok = (top1 - top2 > tau) && (ho[j].y >= top1 - tau + (e1 + rnd + dl) * 1.0001f);
                    if (ok) {
                        A.cand.k[rid[j]] = hint;
                        A.cand.s[rid[j]] = (double)ho[j].x;
                        und = false;
                    }
                }
            }
            const unsigned long long mask = __ballot(und);
            if (mask != 0ull) {
                const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                int base = 0;
                if (lane == 0) base = atomicAdd(&ucnt, __popcll(mask));           // LDS
                base = __shfl(base, 0);
                if (und) {
                    ulist[base + before] = rid[j];
                    uthr[base + before] = thr;
                }
            }
        }
    }
    __syncthreads();
    const int cnt = ucnt;
    if (cnt == 0) return;
    if (tid == 0) ubase = atomicAdd(A.pre_count, cnt);
    __syncthreads();
    const int base = ubase;
    for (int i = tid; i < cnt; i += SEGK_MERGE_THREADS) {
        const int q = base + i;
        const int32_t rid = ulist[i];
        if (q < A.pre_cap) {
            A.pre_queue[q] = rid;
            if (pre_thr) pre_thr[q] = uthr[i];
        } else {                                               // beyond the second stage's launch: full scan
            const int q2 = atomicAdd(A.cand.count, 1);
            if (q2 < A.amb_cap) A.cand.queue[q2] = rid;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
static int launch_score_hint(segk_ctx *ctx, ScoreArgs A, const int32_t *remap, int64_t n_emb, hipStream_t st)
{
    const int n_cu = ctx->n_cu;
    // ---- workspaces: the second stage's queue (as the pre-filter path), K1's partial top-2, the hint map
    if (ctx->pre_cap < A.n) {
        SEGK_REQUIRE(!ctx->capturing, "workspaces must exist before a graph capture (run the sequence once first)");
        if (ctx->pre_queue) SEGK_CHECK_HIP(hipFree(ctx->pre_queue));
        ctx->pre_queue = nullptr;
        ctx->pre_cap = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->pre_queue, sizeof(int32_t) * (size_t)(A.n + 16)));
        ctx->pre_cap = A.n;
    }
    A.pre_queue = ctx->pre_queue + 16;
    A.pre_count = ctx->pre_queue;
    A.pre_cap = (int)A.n;
    const bool band = segk_band_applies(A);
    if (band && ctx->pre_thr_cap < A.n) {
        SEGK_REQUIRE(!ctx->capturing, "workspaces must exist before a graph capture (run the sequence once first)");
        SEGK_CHECK_HIP(hipStreamSynchronize(st));
        if (ctx->pre_thr) (void)hipFree(ctx->pre_thr);
        ctx->pre_thr = nullptr;
        ctx->pre_thr_cap = 0;
        SEGK_CHECK_HIP(hipMalloc((void **)&ctx->pre_thr, sizeof(float) * (size_t)A.n));
        ctx->pre_thr_cap = A.n;
    }
    // K1's ranges: as many tiles as fit in LDS beside nothing else (one workgroup per CU)
    constexpr int TL = KS * 256 + 32;
    const size_t map_bytes = (size_t)((A.K_max + 3) & ~3) * sizeof(int32_t);          // the hint waves' label map behind the images
    SEGK_REQUIRE(map_bytes + TL * sizeof(float) <= 160 * 1024, "hinted score path: K_max too large for the label map in LDS");
    int max_tiles = (int)((160 * 1024 - map_bytes) / (TL * sizeof(float)));
    if (max_tiles > SEGK_HINT_MAX_TPR) max_tiles = SEGK_HINT_MAX_TPR;
    int n_ranges = (A.n_tiles + max_tiles - 1) / max_tiles;
    // two ranges at least when that halves the LDS fill per workgroup without starving the grid (the fill is per workgroup)
    if (n_ranges < 1) n_ranges = 1;
    const int tpr = (A.n_tiles + n_ranges - 1) / n_ranges;
    SEGK_REQUIRE(n_ranges <= 4, "hinted score path: K_max too large (more than four LDS ranges of tile images)");
    // [n_ranges][n] (m1, m2) of the matrix waves, then [n] {s, f_h, h, 0} of the hint waves
    const size_t part_bytes = ((size_t)n_ranges * (size_t)A.n * sizeof(float2) + 255) & ~(size_t)255;
    const size_t need_part = part_bytes + (size_t)A.n * sizeof(float4);
    if (ctx->hint_part_bytes < need_part || !ctx->hint_map || ctx->hint_map_k < A.K_max) {
        SEGK_REQUIRE(!ctx->capturing, "workspaces must exist before a graph capture (run the sequence once first)");
        SEGK_CHECK_HIP(hipStreamSynchronize(st));
        if (ctx->hint_part_bytes < need_part) {
            if (ctx->hint_part) (void)hipFree(ctx->hint_part);
            ctx->hint_part = nullptr;
            ctx->hint_part_bytes = 0;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->hint_part, need_part));
            ctx->hint_part_bytes = need_part;
        }
        if (!ctx->hint_map || ctx->hint_map_k < A.K_max) {
            if (ctx->hint_map) (void)hipFree(ctx->hint_map);
            ctx->hint_map = nullptr;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->hint_map, sizeof(int32_t) * (size_t)A.K_max));
            ctx->hint_map_k = A.K_max;
        }
    }
    // queue lengths of the call (the caller's ambiguity queue, deferred by segk_kmeans_score, and the second stage's)
    int32_t *zero_cnt = ctx->defer_zero;
    ctx->defer_zero = nullptr;
    const int stride_sp = segk_sp_tile_stride(A.D, 2);
    // per-XCD shares of the row groups (see the kernel): three slots of (shares, lifetimes), owned by the context
    if (!ctx->hint_fb) {
        SEGK_CHECK_HIP(hipMalloc(&ctx->hint_fb, 3 * 8 * (sizeof(float) + sizeof(unsigned int)) + 16 * sizeof(int64_t)));
        float init[3 * 8 + 3 * 8];
        for (int i = 0; i < 24; i++) init[i] = 0.125f;
        memset(init + 24, 0, 24 * sizeof(unsigned int));
        SEGK_CHECK_HIP(hipMemcpyAsync(ctx->hint_fb, init, sizeof(init), hipMemcpyHostToDevice, st));
        SEGK_CHECK_HIP(hipStreamSynchronize(st));                            // (`init` lives on this stack frame)
        ctx->hint_fb_launch = 0;
    }
    const bool fb_off = getenv("SEGK_HINT_BALANCE") && atoi(getenv("SEGK_HINT_BALANCE")) == 0;
    float *fb_w = (float *)ctx->hint_fb;
    // (only where a wave has a few dozen groups to shift: at shard sizes -- 4 to 8 groups of 5 us per wave -- shares other
    // than equal ones only make the last round ragged: 1 250 utterances 5 960 against 6 215 sweeps/s, 2 500: 4 878 against 4 948)
    const bool fb_big = (A.n + 63) / 64 >= 192 * 64;
    unsigned int *fb_t = (fb_off || !fb_big) ? nullptr : (unsigned int *)(fb_w + 24);
    int64_t *fb_split = (int64_t *)(fb_w + 48);                  // [9] (+ padding), rewritten by every launch of k_hint_map
    const int fb_cur = (int)(ctx->hint_fb_launch++ % 3u);
    const int64_t total_groups = (A.n + 63) / 64;                // k_kmeans_top2_rs<KS, 4>: two blocks of 32 rows per group
    // K1's grid: four waves per workgroup (one per SIMD with the next group's rows prefetched into registers; the eight-wave
    // instantiation -- two per SIMD, no prefetch: +6 % -- is still in the kernel's template, no longer launched)
    constexpr int nw1 = 4;
    int grid1 = (n_cu / n_ranges) * n_ranges;
    {   // no more workgroups than there are steps per range (each wave takes 64 rows at a time)
        const int64_t rows_ws = 64 * (int64_t)nw1;                          // rows a workgroup takes per step
        const int64_t steps = (A.n + rows_ws - 1) / rows_ws;
        if ((int64_t)grid1 / n_ranges > steps) grid1 = (int)steps * n_ranges;
    }
    // With equal shares every wave walks the groups slot, slot + slots, ...: when a handful of groups is left behind the last
    // whole round (a 1 250-utterance shard: 2 051 groups = 4 x 512 + 3) three waves would take a fifth group, 5 us, for all
    // the others to wait on.  Those few rows skip the filter: marked undecided (k_hint_map), they take the second stage.
    // (with the band stage the rows K1 leaves out would take the full scan, 9 us for a shard's 192 rows: no skipping then)
    int64_t k1_groups = total_groups;
    if (!fb_t && !band) {
        const int64_t slots = (int64_t)(grid1 / n_ranges) * nw1;
        const int64_t whole = slots > 0 ? (total_groups / slots) * slots : 0;
        if (whole > 0 && total_groups - whole <= 32) k1_groups = whole;
    }
    // k_hint_map's work is K1's own (H.own) unless rows are left out (their (m1, m2) must be initialised in front of K1)
    const bool own = k1_groups == total_groups;
    if (!own) {
        const int64_t skipped = A.n - k1_groups * 64 > 0 ? A.n - k1_groups * 64 : 0;
        const int64_t nthr = skipped > A.K_max ? skipped : A.K_max;
        hipLaunchKernelGGL(k_hint_map, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, remap, A.tiles + 1024, A.K_max, stride_sp,
                           KS * 2 * 256, ctx->hint_map, A.n, zero_cnt, ctx->pre_queue, fb_w, fb_t, fb_cur,
                           fb_split, total_groups, k1_groups * 64, (float2 *)ctx->hint_part, n_ranges);
    }

    // ---- K1: matrix waves (top-2 values per row and range) + hint waves (the hinted component in reference arithmetic)
    HintArgs H{};
    H.ximg = (const unsigned char *)A.X32;
    H.ids = A.ids; H.row0 = A.row0; H.n = A.n;
    H.tiles = A.tiles + 1024;
    H.n_tiles = A.n_tiles; H.tpr = tpr; H.n_ranges = n_ranges;
    H.part = (float2 *)ctx->hint_part;
    H.K_max = A.K_max;
    H.dbg = segk_dev_env("SEGK_HINT_DBG");
#ifdef SEGK_STAMP
    H.stamp = getenv("SEGK_STAMP_PTR") ? (unsigned long long *)strtoull(getenv("SEGK_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
    H.fb_w = fb_w;
    H.fb_t = fb_t;
    H.fb_cur = fb_cur;
    H.fb_split = fb_split;
    H.k1_groups = k1_groups;
    H.xrows32 = A.xrows32;
    H.ld32 = A.ld32;
    H.means32 = A.means32;
    H.cand_k = A.cand.k;
    H.map = ctx->hint_map;
    H.own = own ? 1 : 0;
    H.remap = remap;
    H.zero_cnt = zero_cnt;
    H.pre_hdr = ctx->pre_queue;
    H.total_groups = total_groups;
    H.nxx = A.xerr + n_emb;                                      // -|x|^2 per row, behind the residual norms
    H.hint_out = (float4 *)((unsigned char *)ctx->hint_part + part_bytes);
    const size_t lds1 = (size_t)tpr * TL * sizeof(float) + map_bytes;
    const bool prof = segk_prof_now(ctx);
    const int slot = ctx->prof_n % SEGK_PROF_SLOTS;
#define SEGK_K1_LAUNCH(VV)                                                                                                  \
    do {                                                                                                                     \
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_top2_rs<KS, VV, 4>, lds1));                                       \
        if (prof) SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][0], st));                                                 \
        hipLaunchKernelGGL((k_kmeans_top2_rs<KS, VV, 4>), dim3((unsigned)grid1), dim3(512), lds1, st, H);                    \
    } while (0)
    switch ((16 * KS - A.D) / 4) {
        case 0: SEGK_K1_LAUNCH(0); break;
        case 1: SEGK_K1_LAUNCH(1); break;
        case 2: SEGK_K1_LAUNCH(2); break;
        default: SEGK_K1_LAUNCH(3); break;
    }
#undef SEGK_K1_LAUNCH
    if (prof) {
        SEGK_CHECK_HIP(hipEventRecord(ctx->prof_ev[slot][1], st));
        ctx->prof_rows[slot] = A.n;
        ctx->prof_kind = 5;
        ctx->prof_launches = 1;
        ctx->prof_n++;
    }

    // ---- K2: the certificate, one thread per row
    HintMergeArgs E{};
    E.part = (const float2 *)ctx->hint_part;
    E.hint_out = H.hint_out;
    E.n_ranges = n_ranges;
    E.tiles_hdr = A.tiles;
    E.ximg = (const unsigned char *)A.X32;
    {
        // one workgroup per CU, each a contiguous run of at most SEGK_MERGE_ROWS rows
        int64_t grid2 = (int64_t)n_cu;
        if (grid2 * SEGK_MERGE_THREADS > A.n) grid2 = (A.n + SEGK_MERGE_THREADS - 1) / SEGK_MERGE_THREADS;
        if (grid2 * SEGK_MERGE_ROWS < A.n) grid2 = (A.n + SEGK_MERGE_ROWS - 1) / SEGK_MERGE_ROWS;
        const int64_t per = (A.n + grid2 - 1) / grid2;
        hipLaunchKernelGGL(k_hint_merge, dim3((unsigned)grid2), dim3(SEGK_MERGE_THREADS), 0, st, A, E, KS * 16, per, band ? ctx->pre_thr : nullptr,
                           k1_groups * 64);
    }
    // ---- the rows the certificate could not decide: candidates inside the band of the filter's maximum, scored in the
    // reference's arithmetic (segk_score_band.hip); tables beyond its reach keep round 3's three-product second stage
    if (band) {
        if (int rc = segk_launch_band(ctx, A, ctx->pre_thr, A.n, KS, st)) return rc;
        return SEGK_OK;
    }
    // ---- the rows K2 queued: all three products (the pre-filter's second stage); its own undecided rows go to cand.queue
    ScoreArgs B = A;
    B.ids = A.pre_queue;
    B.row0 = 0;
    B.n = A.n;
    B.n_dev = ctx->pre_queue;
    if (int rc = segk_launch_sp_second(ctx, B, KS, st)) return rc;
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int segk_dispatch_score_hint(segk_ctx *ctx, const ScoreArgs &A, const int32_t *remap, int64_t n_emb, int ks, hipStream_t st)
{
    switch (ks) {
        case 1: return launch_score_hint<1>(ctx, A, remap, n_emb, st);
        case 2: return launch_score_hint<2>(ctx, A, remap, n_emb, st);
        case 3: return launch_score_hint<3>(ctx, A, remap, n_emb, st);
        case 4: return launch_score_hint<4>(ctx, A, remap, n_emb, st);
        case 5: return launch_score_hint<5>(ctx, A, remap, n_emb, st);
        case 6: return launch_score_hint<6>(ctx, A, remap, n_emb, st);
        case 7: return launch_score_hint<7>(ctx, A, remap, n_emb, st);
        case 8: return launch_score_hint<8>(ctx, A, remap, n_emb, st);
        default: break;
    }
    segk_set_error("hinted score path: D out of range");
    return SEGK_ERR_UNSUPPORTED;
}
