// segk_score_f32.hip -- A1 filter on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), split-K tail, log-sum-exp twin
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"

// ======================================================================================
// A1 filter: fused fp32 MFMA contraction + running top-2/argmax.
//   workgroup = 4 waves; wave w owns NB blocks of 32 embeddings whose X32 rows live in
//   registers for the whole kernel as the MFMA B operand (lane (j,h): dims 4g+2h+{0,1});
//   the 32-component tiles of the means stream through a double-buffered LDS image and are
//   the A operand, so the 32x32 accumulator has the component on the register index and the
//   embedding on the lane: the running max over components is lane-local.
//   Accumulators start at -|m|^2/2, so acc = x.m - |m|^2/2 with no epilogue arithmetic.
// ======================================================================================
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
    int wg_per_cu = 1;
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score<GMAX, NB, WAVES, 1>, lds));
    SEGK_CHECK_HIP(segk_occupancy((const void *)k_kmeans_score<GMAX, NB, WAVES, 0>, 64 * WAVES, lds, &wg_per_cu));
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
        const bool prof = segk_prof_now(ctx);
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

template <int GMAX>
static int launch_score_lse(segk_ctx *ctx, const ScoreArgs &A, hipStream_t st)
{
    const size_t lds = 2 * (size_t)A.tile_stride * sizeof(float);
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_score<GMAX, 1, 4, 0, 1>, lds));
    const int64_t chunks = (A.n + 127) / 128;
    const bool prof = ctx && segk_prof_now(ctx);
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

int segk_dispatch_score_f32(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const ScoreArgs &A, hipStream_t st)
{
    // Rows per wave: one 32-row MFMA column block per wave (108 VGPRs, four workgroups per CU) beat
    // two blocks sharing every LDS tile fetch (194 VGPRs, two per CU) at every row count measured
    // (D = 100: 77 % vs 73 % of the fp32 matrix peak): the two-block instantiations were retired in round 3.
    switch (segk_gmax(c->D)) {
#define SEGK_CASE(g, nb) \
    case g: return launch_score<g, 1, 4>(ctx, c, m, A, st);
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
