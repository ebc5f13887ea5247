// segk_score_band.hip -- the rows the hint certificate could not decide (round 4): candidates by a BAND around the row's
// largest filter value, then the reference's arithmetic over the candidates.
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
//
// After K1 (segk_score_hint.hip) every row has the two largest values top1 >= top2 of the one-product filter F over all K_max
// component slots, and k_hint_merge has queued the rows it could not certify: a wrong or missing hint (top1 - top2 > tau, the
// winner's INDEX unknown: 6-16 % of the rows in the early sweeps of a chain) and the near-ties (top1 - top2 <= tau: 4.7 % of the
// rows once the chain has settled).  Rounds 2-3 re-scored all of them with the three-product fp16x2 kernel (index tracking,
// 61 us for 49 600 rows, three times that in the early sweeps) and sent what that could not decide through the full scan.
// But the filter's margin already says where the reference's argmax can be: with |F_k - f_k| <= E for every k and
// tau >= 2 E + E2 (filter_tau_h1; E2 = the rounding of the reference's own float32 evaluation),
//
//     k_ref = np.argmax(neg_sqrd_norm)   =>   F_kref >= top1 - tau
//
// (a component below the band loses to the filter's argmax in the reference's arithmetic as well).  So:
//
//   K3  k_kmeans_band_rs   the queued rows once more through the one-product contraction -- K1's range-stationary loop, the rows
//       gathered by queue entry -- and the drain only COMPARES: one bit per (row, component), F >= thr = top1 - tau, shifted
//       into a mask register with two instructions per value (v_cmp_ge_f32 + v_addc_co_u32 mask, mask, mask, vcc) and no
//       branch; 16 bytes of mask per (row, lane half, range of 8 tiles).  Nothing about the comparison depends on reproducing
//       K1's bits: any evaluation of F within E of f keeps k_ref inside the band.
//   K4  k_band_exact   one thread per mask word: every set bit is a component scored in the reference's arithmetic
//       (neg_sqd_exact's order), the row's candidates meet by a 64-bit maximum of (score, ~component) in LDS: first maximum,
//       as np.argmax.  A row without a candidate (cannot happen while the bounds hold) goes to the full scan.
//
// Typically two or three candidates per near-tie row, one per wrong hint: the second stage and almost all of the full scan are
// replaced by ~1.2 x K1's cost per queued row plus a few hundred bytes per row.
#include "segk_kmeans_dev.h"

#define SEGK_BAND_TPR 8            /* tiles per LDS range: 16 bits per tile and lane half -> four 32-bit mask words */
#define SEGK_BAND_MAX_RANGES 8     /* K_max <= 2048 */

struct BandArgs {
    const unsigned char *ximg;      // fp16x2 row image (segk_corpus.Xb3): header, then plane 0 [n_emb][KP]
    const int32_t *queue;           // row ids of the undecided rows
    const float *thr;               // per queue entry: top1 - tau in the scaled domain of the images
    const int32_t *n_dev;           // queue length (device)
    int cap;                        // queue capacity
    const float *tiles;             // first tile of the fp16x2 tile image (tiles_b3 + 1024)
    int n_tiles, n_ranges;
    int4 *mask;                     // [n_ranges][cap][2 lane halves] four words each
    unsigned long long *fb;         // host-mapped feedback word (NULL: none): (fb_seq << 32) | permille of fb_rows that were queued
    unsigned int fb_seq;
    int64_t fb_rows;
};

// four values: bit = (value >= thr), shifted in from the right (value q of a tile ends at bit 15 - q of its half word)
#define SEGK_BAND_QUAD(MK, AO, q_, THR, FIRST)                                                                       \
    do {                                                                                                              \
        if (FIRST) {   /* the first read of the MFMA's result stays visible to the compiler's hazard recogniser */    \
            MK = (MK << 1) | (AO[4 * (q_)] >= THR ? 1u : 0u);                                                         \
            asm volatile("v_cmp_ge_f32 vcc, %1, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"                        \
                         "v_cmp_ge_f32 vcc, %2, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"                        \
                         "v_cmp_ge_f32 vcc, %3, %4\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"                            \
                         : "+v"(MK) : "v"(AO[4 * (q_) + 1]), "v"(AO[4 * (q_) + 2]), "v"(AO[4 * (q_) + 3]), "v"(THR) : "vcc"); \
        } else {                                                                                                      \
            asm volatile("v_cmp_ge_f32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"                        \
                         "v_cmp_ge_f32 vcc, %2, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"                        \
                         "v_cmp_ge_f32 vcc, %3, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"                        \
                         "v_cmp_ge_f32 vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"                            \
                         : "+v"(MK) : "v"(AO[4 * (q_)]), "v"(AO[4 * (q_) + 1]), "v"(AO[4 * (q_) + 2]), "v"(AO[4 * (q_) + 3]), "v"(THR) : "vcc"); \
        }                                                                                                             \
    } while (0)

// K3.  Four waves per workgroup, one per SIMD (256 registers: two row sets, the next group's rows prefetched), a (range, slot)
// pair per workgroup as in k_kmeans_top2_rs; groups of 64 queue entries (two blocks of 32).  The range always runs
// SEGK_BAND_TPR tiles: tiles beyond the table are filled with zero operands and "absent" constants (-3e38: never inside a band).
template <int KS>
__global__ __launch_bounds__(256, 2) void k_kmeans_band_rs(BandArgs B)
{
    typedef _Float16 T;
    typedef SegkPiece<2>::V8 V8;
    constexpr int P = 2, KP = KS * 16, NW = 4, NBLK = 2, TPR = SEGK_BAND_TPR;
    constexpr int STRIDE = (KS * P * 256 + 32 + 1023) / 1024 * 1024;      // floats per tile of the global image
    constexpr int TL = KS * 256 + 32;                                     // floats per tile in LDS: KS piece-0 blocks + constants
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int R = B.n_ranges;
    int count = *B.n_dev;
    if (count > B.cap) count = B.cap;
    const int64_t n_groups = ((int64_t)count + 32 * NBLK - 1) / (32 * NBLK);
    if (B.fb && blockIdx.x == 0 && tid == 0) {
        const unsigned long long pm = B.fb_rows > 0 ? (unsigned long long)((int64_t)count * 1000 / B.fb_rows) : 0ull;
        __hip_atomic_store(B.fb, ((unsigned long long)B.fb_seq << 32) | pm, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // workgroup -> (range, slot); the R workgroups that stream the same queue entries share an XCD when the grid allows
    int range, wgr, n_wgr;
    if ((gridDim.x & 7) == 0 && ((gridDim.x >> 3) % R) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        range = idx % R;
        wgr = (idx / R) * 8 + xcd;
        n_wgr = gridDim.x / R;
    } else {
        n_wgr = gridDim.x / R;
        range = blockIdx.x % R;
        wgr = blockIdx.x / R;
        if (wgr >= n_wgr) return;
    }
    if ((int64_t)wgr * NW >= n_groups) return;                // (workgroup-uniform) nothing queued for this slot: no LDS fill either
    const int t_lo = range * TPR;
    int nt = B.n_tiles - t_lo;
    if (nt > TPR) nt = TPR;
    if (nt < 0) nt = 0;
    const T *plane0 = (const T *)(B.ximg + SEGK_SP_HEADER);
    const int64_t n_slots = (int64_t)n_wgr * NW;
    int64_t g = (int64_t)wgr * NW + wave;

    // queue entries of a group: row ids (-1 beyond the queue)
#define SEGK_BD_IDS(g_, RID)                                                                 \
    do {                                                                                      \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                    \
            const int64_t r = (g_) * (32 * NBLK) + 32 * b + j;                                \
            RID[b] = r < count ? B.queue[r] : -1;                                             \
        }                                                                                     \
    } while (0)
    // the rows of a group (ids already in registers) and their thresholds
#define SEGK_BD_LOAD(g_, XB, THR, RID)                                                                           \
    do {                                                                                                          \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                                        \
            const int64_t r = (g_) * (32 * NBLK) + 32 * b + j;                                                    \
            const int64_t rowid = RID[b] >= 0 ? (int64_t)RID[b] : 0;       /* some valid row, result unused */     \
            const T *xp = plane0 + rowid * KP + 8 * h;                                                            \
            _Pragma("unroll") for (int s = 0; s < KS; s++) XB[b][s] = *reinterpret_cast<const V8 *>(xp + 16 * s); \
            THR[b] = (r < count && RID[b] >= 0) ? B.thr[r] : __builtin_huge_valf();                               \
        }                                                                                                         \
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

    // MFMAs of block N_ (accumulator AN) with the compare of block O_'s previous values (accumulator AO, mask word MK) spread
    // over the MFMAs 1 .. KS-1; DRAIN = false: no values to compare yet (block 1 of "tile -1")
#define SEGK_BD_UNIT(XB, N_, AN, AO, MK, THR, DRAIN, REFILL, tn_)                                                     \
    do {                                                                                                              \
        _Pragma("unroll") for (int s = 0; s < KS; s++) {                                                              \
            asm volatile("" : "+v"(a[s]));                                                                            \
            AN = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], XB[N_][s], s == 0 ? cs : AN, 0, 0, 0);                  \
            asm volatile("" : "+v"(AN), "+v"(THR));    /* (the threshold too: the visible compare below stays behind this MFMA) */ \
            if (REFILL) {                                                                                             \
                load_a(tn_, s);                                                                                       \
                if (s == 0) load_cs(tn_);                                                                             \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            constexpr int SL = KS > 1 ? KS - 1 : 1;                                                                   \
            const int q_lo = KS > 1 ? ((s - 1) * 4 + SL - 1) / SL : 0, q_hi = KS > 1 ? (s * 4 + SL - 1) / SL : 4;     \
            if (DRAIN && (KS == 1 || s >= 1)) {                                                                       \
                if (0 >= q_lo && 0 < q_hi) SEGK_BAND_QUAD(MK, AO, 0, THR, true);                                      \
                if (1 >= q_lo && 1 < q_hi) SEGK_BAND_QUAD(MK, AO, 1, THR, false);                                     \
                if (2 >= q_lo && 2 < q_hi) SEGK_BAND_QUAD(MK, AO, 2, THR, false);                                     \
                if (3 >= q_lo && 3 < q_hi) SEGK_BAND_QUAD(MK, AO, 3, THR, false);                                     \
            }                                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
        }                                                                                                             \
    } while (0)

    // one group: the range's TPR tiles against the rows in XB.  Block 0's values of tile t are compared under block 1's MFMAs
    // of tile t, block 1's under block 0's MFMAs of tile t + 1 (the last ones behind the loop); the loop is unrolled, so the
    // mask words are registers with constant indices: tile t of a block -> word t / 2, upper half first
#define SEGK_BD_GROUP(g_, XB, THR)                                                                                     \
    do {                                                                                                               \
        unsigned int mk[NBLK][TPR / 2];                                                                                \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++)                                                               \
            _Pragma("unroll") for (int w = 0; w < TPR / 2; w++) mk[b][w] = 0u;                                         \
        f32x16 acc0, acc1;                                                                                             \
        _Pragma("unroll") for (int t = 0; t < TPR; t++) {                                                              \
            const int tn = t + 1 < TPR ? t + 1 : 0;       /* the last tile refills tile 0's operands: the next group's */ \
            if (t == 0) SEGK_BD_UNIT(XB, 0, acc0, acc1, mk[1][0], THR[1], false, false, tn);                           \
            else        SEGK_BD_UNIT(XB, 0, acc0, acc1, mk[1][(t - 1) >> 1], THR[1], true, false, tn);                 \
            SEGK_BD_UNIT(XB, 1, acc1, acc0, mk[0][t >> 1], THR[0], true, true, tn);                                    \
        }                                                                                                              \
        SEGK_BAND_QUAD(mk[1][(TPR - 1) >> 1], acc1, 0, THR[1], true);                                                  \
        SEGK_BAND_QUAD(mk[1][(TPR - 1) >> 1], acc1, 1, THR[1], false);                                                 \
        SEGK_BAND_QUAD(mk[1][(TPR - 1) >> 1], acc1, 2, THR[1], false);                                                 \
        SEGK_BAND_QUAD(mk[1][(TPR - 1) >> 1], acc1, 3, THR[1], false);                                                 \
        _Pragma("unroll") for (int b = 0; b < NBLK; b++)                                                               \
            _Pragma("unroll") for (int w = 0; w < TPR / 2; w++) pend[b][w] = mk[b][w];                                 \
        pend_g = (g_);                                                                                                 \
    } while (0)

    // the masks of the group before (stored behind the next prefetch: a store in front of the loads would make every wait
    // for rows wait out its acknowledgement as well)
#define SEGK_BD_STORE()                                                                                                \
    do {                                                                                                               \
        if (pend_g >= 0) {                                                                                             \
            _Pragma("unroll") for (int b = 0; b < NBLK; b++) {                                                         \
                const int64_t r = pend_g * (32 * NBLK) + 32 * b + j;                                                   \
                if (r < count)                                                                                         \
                    B.mask[((int64_t)range * B.cap + r) * 2 + h] = make_int4((int)pend[b][0], (int)pend[b][1], (int)pend[b][2], (int)pend[b][3]); \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)
    static_assert(TPR == 8, "four mask words per (row, lane half, range)");

    unsigned int pend[NBLK][TPR / 2];
    int64_t pend_g = -1;
    V8 xa[NBLK][KS], xb[NBLK][KS];
    float thr_a[NBLK], thr_b[NBLK];
    int32_t rid_a[NBLK], rid_b[NBLK];
    _Pragma("unroll") for (int b = 0; b < NBLK; b++) { rid_a[b] = -1; rid_b[b] = -1; thr_a[b] = 0.f; thr_b[b] = 0.f; }
    if (g < n_groups) SEGK_BD_IDS(g, rid_a);
    if (g + n_slots < n_groups) SEGK_BD_IDS(g + n_slots, rid_b);
    // ---- the range's tile images into LDS, once (see k_kmeans_top2_rs): 1 KiB piece-0 blocks, all of a wave's loads in flight
    {
        constexpr int MAXB = TPR * KS;
        constexpr int PER_W = (MAXB + NW - 1) / NW;
        const int n_blk = nt * KS;
        float4 v[PER_W];
#pragma unroll
        for (int u = 0; u < PER_W; u++) {
            int c = wave + u * NW;
            const bool live = c < n_blk;
            if (!live) c = 0;
            const int t = c / KS, ks = c - t * KS;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v[u] = *reinterpret_cast<const float4 *>(B.tiles + (int64_t)(t_lo + t) * STRIDE + ks * (P * 256) + lane * 4);
        }
        float4 cv4 = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);     // a tile beyond the table: absent
        const bool is_c = tid < TPR * 8;                                // the 32 constants of every tile: one float4 per thread
        if (is_c && (tid >> 3) < nt) cv4 = *reinterpret_cast<const float4 *>(B.tiles + (int64_t)(t_lo + (tid >> 3)) * STRIDE + KS * P * 256 + (tid & 7) * 4);
#pragma unroll
        for (int u = 0; u < PER_W; u++) {
            const int c = wave + u * NW;
            if (c < MAXB) {
                const int t = c / KS, ks = c - t * KS;
                *reinterpret_cast<float4 *>(lds + t * TL + ks * 256 + lane * 4) = v[u];
            }
        }
        if (is_c) *reinterpret_cast<float4 *>(lds + (tid >> 3) * TL + KS * 256 + (tid & 7) * 4) = cv4;
        static_assert(256 >= TPR * 8, "one thread per float4 of the constants");
    }
    if (g < n_groups) SEGK_BD_LOAD(g, xa, thr_a, rid_a);       // (the ids have landed: the copy above waited for its own loads behind them)
    __syncthreads();
    if (g >= n_groups) return;
#pragma unroll
    for (int s = 0; s < KS; s++) load_a(0, s);
    load_cs(0);
    for (;;) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): the rows of group g are in xa, the ids of g1 in rid_b
        const int64_t g1 = g + n_slots;
        if (g1 < n_groups) SEGK_BD_LOAD(g1, xb, thr_b, rid_b);          // in flight under this group's tile loop
        if (g1 + n_slots < n_groups) SEGK_BD_IDS(g1 + n_slots, rid_a);  // ids two groups ahead
        SEGK_BD_STORE();
        SEGK_BD_GROUP(g, xa, thr_a);
        if (g1 >= n_groups) break;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        g = g1 + n_slots;
        if (g < n_groups) SEGK_BD_LOAD(g, xa, thr_a, rid_a);
        if (g + n_slots < n_groups) SEGK_BD_IDS(g + n_slots, rid_b);
        SEGK_BD_STORE();
        SEGK_BD_GROUP(g1, xb, thr_b);
        if (g >= n_groups) break;
    }
    SEGK_BD_STORE();
#undef SEGK_BD_STORE
#undef SEGK_BD_GROUP
#undef SEGK_BD_UNIT
#undef SEGK_BD_LOAD
#undef SEGK_BD_IDS
}
#undef SEGK_BAND_QUAD

struct BandExactArgs {
    const int32_t *queue;
    const int32_t *n_dev;
    int cap;
    const int4 *mask;
    int n_ranges, K_max, D;
    const float *xrows32;
    int64_t ld32;
    const float *means32;
    segk_cand cand;
    int amb_cap;
};

// K4.  One thread per (queue entry, mask word): words per row = n_ranges x 2 lane halves x 4.  Bit pb of word ww of (range,
// half h): tile 8 range + 2 ww + (pb < 16), value q = 15 - (pb & 15) of that tile -> component 32 tile + 4 h + (q & 3) +
// 8 (q >> 2) (the accumulator layout of v_mfma_f32_32x32x16).  The row's candidates meet in LDS: 64-bit maximum of (orderable
// score bits, ~component) -- the largest score, among equal scores the lowest component: np.argmax's first maximum.
#define SEGK_BAND_CLIST 128        /* candidates of one wave's rows per trip at most (beyond: those rows take the full scan) */
// K4 proper.  A wave takes 64 / (words per row) queue entries per trip, one lane per mask word, and works alone (no barrier; two
// earlier forms -- a workgroup per eight rows with LDS maxima and three barriers, then a lane per candidate with the rows fetched
// in chunks -- were chains of four to eight dependent round trips per trip: 57-63 us for 49 600 rows).  The set bits of the
// wave's words are listed in LDS (prefix sum over the lanes), then EIGHT lanes score one candidate: lane a owns numpy's strided
// accumulator r_a (elements a, a + 8, ... in order), the eight meet by the fixed tree ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7))
// -- three xor-shuffles, every lane the same additions --, the tail elements are added one by one by lane 0.  All of a
// candidate's loads (two dozen 4-byte elements per lane, 32-byte segments per group) are in flight together: three dependent
// round trips per trip (queue length; words and row ids; elements).
template <int D>
__global__ __launch_bounds__(256) void k_band_exact(BandExactArgs E)
{
    __shared__ int2 clist[4][SEGK_BAND_CLIST];
    __shared__ unsigned long long best[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wpr = E.n_ranges * 8;                      // words per row: 8, 16, 24, ... 64
    const int rpw = 64 / wpr;                            // rows per wave and trip (8 at most)
    int count = *E.n_dev;
    if (count > E.cap) count = E.cap;
    const int row_l = lane / wpr, w = lane - row_l * wpr;
    const int range = w >> 3, h = (w >> 2) & 1, ww = w & 3;
    const bool worker = row_l < rpw;
    constexpr int nblk = D / 8, rem = D & 7;
    const int grp = lane >> 3, a = lane & 7;
    const int64_t wave_g = ((int64_t)blockIdx.x * 256 + tid) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t q0 = wave_g * rpw; q0 < count; q0 += n_waves * rpw) {
        const int64_t q = q0 + row_l;
        unsigned int word = 0u;
        int32_t rid = -1;
        if (worker && q < count) {
            word = ((const unsigned int *)E.mask)[(((int64_t)range * E.cap + q) * 2 + h) * 4 + ww];
            rid = E.queue[q];                                // (with the word: one round trip, not two)
        }
        // positions of the lanes' candidates in the wave's list: exclusive prefix sum of the bit counts
        const int pc = __popc(word);
        int pre = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(pre, off);
            if (lane >= off) pre += o;
        }
        const int total = __shfl(pre, 63);
        pre -= pc;
        if (lane < 8) best[wv][lane] = 0ull;
        const bool overflow = total > SEGK_BAND_CLIST;
        if (!overflow) {
            int at = pre;
            while (word != 0u) {
                const int pb = 31 - __clz((int)word);
                word &= ~(1u << pb);
                const int tile = range * SEGK_BAND_TPR + 2 * ww + (pb < 16 ? 1 : 0);
                const int qq = 15 - (pb & 15);
                const int comp = 32 * tile + 4 * h + (qq & 3) + 8 * (qq >> 2);
                clist[wv][at++] = make_int2(row_l, comp < E.K_max ? comp : -1);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!overflow) {
            for (int c0 = 0; c0 < total; c0 += 8) {
                const int c = c0 + grp;
                int2 cd = make_int2(0, -1);
                if (c < total) cd = clist[wv][c];
                const int32_t rid_c = __shfl(rid, cd.x * wpr);
                const bool live = cd.y >= 0 && rid_c >= 0;
                const float *mp = E.means32 + (int64_t)(live ? cd.y : 0) * D + a;
                const float *xp = E.xrows32 + (int64_t)(live ? rid_c : 0) * E.ld32 + a;
                float mv[nblk], xv[nblk], mt = 0.f, xt = 0.f;
#pragma unroll
                for (int b = 0; b < nblk; b++) mv[b] = mp[8 * b];
#pragma unroll
                for (int b = 0; b < nblk; b++) xv[b] = xp[8 * b];
                if constexpr (rem != 0) {
                    if (a < rem) { mt = mp[8 * nblk]; xt = xp[8 * nblk]; }
                }
                float r = 0.f;
#pragma unroll
                for (int b = 0; b < nblk; b++) {
                    const float dlt = mv[b] - xv[b];
                    const float t = dlt * dlt;
                    r = b == 0 ? t : r + t;
                }
                r = r + __shfl_xor(r, 1);                    // (r0+r1), (r2+r3), ...
                r = r + __shfl_xor(r, 2);                    // ((r0+r1)+(r2+r3)), ...
                r = r + __shfl_xor(r, 4);                    // the whole tree, the same on the eight lanes
                if constexpr (rem != 0) {
                    const float dlt = mt - xt;
                    const float t = dlt * dlt;
#pragma unroll
                    for (int i = 0; i < rem; i++) r += __shfl(t, (lane & ~7) + i);      // the sequential tail
                }
                if (live && a == 0) {
                    const float sc = -r;
                    const unsigned int bits = __float_as_uint(sc);
                    const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                    const unsigned long long pk = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)cd.y);
                    atomicMax(&best[wv][cd.x], pk);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (worker && q < count && w == 0) {
            const unsigned long long mine = overflow ? 0ull : best[wv][row_l];
            if (mine != 0ull) {
                const unsigned int ord = (unsigned int)(mine >> 32);
                const unsigned int bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
                E.cand.k[rid] = (int32_t)(0xffffffffu - (unsigned int)(mine & 0xffffffffu));
                E.cand.s[rid] = (double)__uint_as_float(bits);
            } else {                                       // no candidate inside the band (or too many): the full scan decides
                const int q2 = atomicAdd(E.cand.count, 1);
                if (q2 < E.amb_cap) E.cand.queue[q2] = rid;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int KS>
static int launch_band(segk_ctx *ctx, const ScoreArgs &A, const float *thr, int64_t call_rows, hipStream_t st)
{
    const int n_cu = ctx->n_cu;
    const int n_ranges = (A.n_tiles + SEGK_BAND_TPR - 1) / SEGK_BAND_TPR;
    SEGK_REQUIRE(n_ranges >= 1 && n_ranges <= SEGK_BAND_MAX_RANGES, "band stage: K_max out of range");
    const int cap = A.pre_cap;
    const size_t need = (size_t)n_ranges * (size_t)cap * 2 * sizeof(int4);
    if (ctx->band_mask_bytes < need) {
        SEGK_REQUIRE(!ctx->capturing, "workspaces must exist before a graph capture (run the sequence once first)");
        SEGK_CHECK_HIP(hipStreamSynchronize(st));
        if (ctx->band_mask) (void)hipFree(ctx->band_mask);
        ctx->band_mask = nullptr;
        ctx->band_mask_bytes = 0;
        SEGK_CHECK_HIP(hipMalloc(&ctx->band_mask, need));
        ctx->band_mask_bytes = need;
    }
    BandArgs B{};
    B.ximg = (const unsigned char *)A.X32;
    B.queue = A.pre_queue;
    B.thr = thr;
    B.n_dev = A.pre_count;
    B.cap = cap;
    B.tiles = A.tiles + 1024;
    B.n_tiles = A.n_tiles;
    B.n_ranges = n_ranges;
    B.mask = (int4 *)ctx->band_mask;
    B.fb = ctx->miss_dev;
    B.fb_seq = ++ctx->miss_seq;
    B.fb_rows = call_rows;
    constexpr int TL = KS * 256 + 32;
    const size_t lds = (size_t)SEGK_BAND_TPR * TL * sizeof(float);
    // two workgroups per CU (58 KB of LDS and 4 x 256 registers each): a (range, slot) pair per workgroup
    int grid = (2 * n_cu / n_ranges) * n_ranges;
    {
        const int64_t steps = ((int64_t)cap + 255) / 256;                 // a workgroup takes 4 x 64 queue entries per step
        if ((int64_t)grid / n_ranges > steps) grid = (int)steps * n_ranges;
    }
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_band_rs<KS>, lds));
    hipLaunchKernelGGL((k_kmeans_band_rs<KS>), dim3((unsigned)grid), dim3(256), lds, st, B);

    BandExactArgs E{};
    E.queue = A.pre_queue;
    E.n_dev = A.pre_count;
    E.cap = cap;
    E.mask = (const int4 *)ctx->band_mask;
    E.n_ranges = n_ranges;
    E.K_max = A.K_max;
    E.D = A.D;
    E.xrows32 = A.xrows32;
    E.ld32 = A.ld32;
    E.means32 = A.means32;
    E.cand = A.cand;
    E.amb_cap = A.amb_cap;
    // (the queue length is on the device: a grid for the queue lengths that occur -- 5 to 20 % of the rows --, waves stride)
    const int rpw = 64 / (n_ranges * 8);
    int64_t grid4 = ((int64_t)cap + 4 * rpw - 1) / (4 * rpw);
    if (grid4 > 16 * (int64_t)n_cu) grid4 = 16 * (int64_t)n_cu;
    switch ((16 * KS - A.D) / 4) {
        case 0: hipLaunchKernelGGL((k_band_exact<16 * KS>), dim3((unsigned)grid4), dim3(256), 0, st, E); break;
        case 1: hipLaunchKernelGGL((k_band_exact<16 * KS - 4>), dim3((unsigned)grid4), dim3(256), 0, st, E); break;
        case 2: hipLaunchKernelGGL((k_band_exact<16 * KS - 8>), dim3((unsigned)grid4), dim3(256), 0, st, E); break;
        default: hipLaunchKernelGGL((k_band_exact<(16 * KS - 12 >= 8 ? 16 * KS - 12 : 8)>), dim3((unsigned)grid4), dim3(256), 0, st, E); break;
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// does the band stage cover this table?  (otherwise the hinted path keeps the three-product second stage)
bool segk_band_applies(const ScoreArgs &A)
{
    const int n_ranges = (A.n_tiles + SEGK_BAND_TPR - 1) / SEGK_BAND_TPR;
    return n_ranges >= 1 && n_ranges <= SEGK_BAND_MAX_RANGES && A.D % 4 == 0 && A.D >= 8 && A.D <= 128;
}

int segk_launch_band(segk_ctx *ctx, const ScoreArgs &A, const float *thr, int64_t call_rows, int ks, hipStream_t st)
{
    switch (ks) {
        case 1: return launch_band<1>(ctx, A, thr, call_rows, st);
        case 2: return launch_band<2>(ctx, A, thr, call_rows, st);
        case 3: return launch_band<3>(ctx, A, thr, call_rows, st);
        case 4: return launch_band<4>(ctx, A, thr, call_rows, st);
        case 5: return launch_band<5>(ctx, A, thr, call_rows, st);
        case 6: return launch_band<6>(ctx, A, thr, call_rows, st);
        case 7: return launch_band<7>(ctx, A, thr, call_rows, st);
        case 8: return launch_band<8>(ctx, A, thr, call_rows, st);
        default: break;
    }
    segk_set_error("band stage: D out of range");
    return SEGK_ERR_UNSUPPORTED;
}
