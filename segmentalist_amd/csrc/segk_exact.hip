// segk_exact.hip -- exact stage: full scan of the ambiguity queue in the reference's arithmetic, A1 vector, candidate gather
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"

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
// BR = 4 (was 8: 234 VGPRs): when round 1 ran the scan on a second stream beside the exact pair
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
    __builtin_amdgcn_s_setprio(3);            // tail of the score stage's critical branch, beside the exact pair kernel's waves

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
    // all K_max slots: the reference's argmax runs over every row of `means`, and the slots beyond K hold random_means
    // (kmeans_components.py:149-166, 225-226) -- a token that lands there opens a new component
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
            // two elements per instruction (v_pk_add_f32 with the second operand negated, v_pk_mul_f32, v_pk_add_f32: every
            // half rounded like the scalar operation): the scan is bound by the vector ALU
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            auto pk_sub = [](f32x2_t a, f32x2_t b2) -> f32x2_t {
                f32x2_t d;
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b2));
                return d;
            };
            // the eight elements of block i as four adjacent pairs of the tile image (TileRow: pairs (d, d + 1), d even)
            auto load_m = [&](int i, f32x2_t *mp) {
                const float *b = mr.base + (i >> 2) * 128;
#pragma unroll
                for (int q = 0; q < 4; q++) mp[q] = *reinterpret_cast<const f32x2_t *>(b + (q >> 1) * 128 + (q & 1) * 64);
            };
            f32x2_t acc[SEGK_BR][4];
            f32x2_t mp[4];
            load_m(0, mp);
#pragma unroll
            for (int r = 0; r < SEGK_BR; r++) {
                const float4 x0 = *reinterpret_cast<const float4 *>(xs + r * DP), x1 = *reinterpret_cast<const float4 *>(xs + r * DP + 4);
                const f32x2_t xp[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x2_t d = pk_sub(mp[q], xp[q]);
                    acc[r][q] = d * d;
                }
            }
            int i;
            for (i = 8; i < nfull; i += 8) {
                load_m(i, mp);
#pragma unroll
                for (int r = 0; r < SEGK_BR; r++) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xs + r * DP + i), x1 = *reinterpret_cast<const float4 *>(xs + r * DP + i + 4);
                    const f32x2_t xp[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x2_t d = pk_sub(mp[q], xp[q]);
                        acc[r][q] += d * d;
                    }
                }
            }
            float res[SEGK_BR];
#pragma unroll
            for (int r = 0; r < SEGK_BR; r++)
                res[r] = ((acc[r][0].x + acc[r][0].y) + (acc[r][1].x + acc[r][1].y)) + ((acc[r][2].x + acc[r][2].y) + (acc[r][3].x + acc[r][3].y));
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

// The full scan with the COMPONENTS resident in LDS (round 3; float32 data, 8 <= D <= 128).  k_kmeans_brute_rows gives
// every group of four queued rows a workgroup of its own, which reads the whole table of means for them: 1 600 rows = 160 MB
// through the L2s and a thousand workgroups to dispatch.  Here a workgroup keeps a slice of BLS_TPS tiles (128 components:
// 51 KB as float32, the layout of the tile image) and walks a CHUNK of the queue with it, 32 rows at a time: the ids of the
// chunk are fetched once, the next 32 rows are in flight while the current 32 are scored.  Thread = (component of the slice,
// four of the 32 rows), sixteen waves (with four -- one per SIMD -- every dependent LDS read and packed operation was
// exposed); arithmetic and first-maximum rule as in k_kmeans_brute_rows (neg_sqd_exact's order, packed pairs); the slices of
// a row meet in ws[] by a 64-bit atomic maximum of (orderable score bits, ~component), one per wave, unpacked by
// the workgroup that finishes last.
#define BLS_TPS 4
#define BLS_THREADS 1024
#define BLS_ROWS (BLS_THREADS / 128 * 4)      /* rows per batch */
#define BLS_IDS 256
__global__ __launch_bounds__(BLS_THREADS) void k_kmeans_brute_ls(segk_corpus c, segk_kmeans m, segk_cand cand, int cap, int32_t *n_brute,
                                                                int n_slices, int n_chunks, unsigned long long *ws, unsigned int *ticket)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, D = c.D;
    const int DP = (D + 3) & ~3;                             // row pitch: float4 reads of the staged rows
    const int TP = segk_G(D) * 128;                          // floats of a tile that hold means
    const int tstride = segk_tile_stride(D);
    float *ts = (float *)smem;                               // [BLS_TPS][TP]
    float *xs = ts + BLS_TPS * TP;                           // [BLS_ROWS][DP]
    __shared__ int32_t ids[BLS_IDS];
    const float *X = (const float *)c.X;
    int nq = *cand.count;
    if (nq > cap) nq = cap;
    if (blockIdx.x == 0 && tid == 0 && n_brute && nq > 0) atomicAdd(n_brute, nq);
    const int slice = blockIdx.x % n_slices, chunk = blockIdx.x / n_slices;
    const int per = ((nq + n_chunks - 1) / n_chunks + BLS_ROWS - 1) & ~(BLS_ROWS - 1);
    const int q_lo = chunk * per, q_hi = q_lo + per < nq ? q_lo + per : nq;
    if (nq <= 0) return;                                     // (every workgroup alike) nothing queued: nothing to unpack either
    const bool has = q_lo < q_hi;                            // a workgroup without rows still takes its ticket below
    // ---- the slice's tiles (all K_max slots: the reference's argmax runs over every row of `means`)
    const int n_tiles = segk_n_tiles(m.K_max);
    const int tile0 = slice * BLS_TPS;
    const int nth = n_tiles - tile0 < BLS_TPS ? n_tiles - tile0 : BLS_TPS;
    for (int i = tid; has && i < nth * (TP / 4); i += BLS_THREADS) {
        const int t = i / (TP / 4), o = i - t * (TP / 4);
        reinterpret_cast<float4 *>(ts + t * TP)[o] = reinterpret_cast<const float4 *>(m.tiles + (int64_t)(tile0 + t) * tstride)[o];
    }
    const int comp_l = tid & 127, rh = tid >> 7;
    const int k = tile0 * 32 + comp_l;
    const bool live = comp_l < nth * 32 && k < m.K_max;
    const TileRow mr{ts + (live ? (comp_l >> 5) * TP + 2 * (comp_l & 31) : 0)};
    const int nfull = D - (D % 8);
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto pk_sub = [](f32x2_t a, f32x2_t b2) -> f32x2_t {
        f32x2_t d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b2));
        return d;
    };
    auto load_m = [&](int i, f32x2_t *mp) {                  // the eight elements of block i as four adjacent pairs
        const float *b = mr.base + (i >> 2) * 128;
#pragma unroll
        for (int q = 0; q < 4; q++) mp[q] = *reinterpret_cast<const f32x2_t *>(b + (q >> 1) * 128 + (q & 1) * 64);
    };
    // elements of a batch of rows this thread moves: j = tid + BLS_THREADS e -> (row j / D, dimension j % D)
    constexpr int NPF = 4;                                   // BLS_ROWS rows x D <= 128 floats / BLS_THREADS threads
    float pf[NPF];
    auto fetch = [&](int qb, int ib) {                       // rows qb .. of the queue (ids at ids[ib ..])
#pragma unroll
        for (int e = 0; e < NPF; e++) {
            const int j = tid + BLS_THREADS * e;
            const int r = j / D, d = j - r * D;
            const bool ok = r < BLS_ROWS && qb + r < q_hi;
            pf[e] = ok ? X[(int64_t)ids[ib + r] * c.ldx + d] : 0.f;
        }
    };
    for (int s0 = q_lo; s0 < q_hi; s0 += BLS_IDS) {
        const int ns = q_hi - s0 < BLS_IDS ? q_hi - s0 : BLS_IDS;
        __syncthreads();                                     // the previous super-chunk's ids are no longer read
        if (tid < ns) ids[tid] = cand.queue[s0 + tid];
        __syncthreads();
        fetch(s0, 0);
        for (int b0 = 0; b0 < ns; b0 += BLS_ROWS) {
            const int q0 = s0 + b0;
            const int nr = q_hi - q0 < BLS_ROWS ? q_hi - q0 : BLS_ROWS;
            __syncthreads();                                 // the previous batch's rows are no longer read (first trip: the tiles are staged)
#pragma unroll
            for (int e = 0; e < NPF; e++) {
                const int j = tid + BLS_THREADS * e;
                const int r = j / D, d = j - r * D;
                if (r < BLS_ROWS) xs[r * DP + d] = pf[e];
            }
            __syncthreads();
            if (b0 + BLS_ROWS < ns) fetch(q0 + BLS_ROWS, b0 + BLS_ROWS);        // in flight under the arithmetic
            const float *xr = xs + rh * 4 * DP;
            f32x2_t acc[4][4];
            f32x2_t mp[4];
            load_m(0, mp);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 x0 = *reinterpret_cast<const float4 *>(xr + r * DP), x1 = *reinterpret_cast<const float4 *>(xr + r * DP + 4);
                const f32x2_t xp[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const f32x2_t d = pk_sub(mp[q], xp[q]);
                    acc[r][q] = d * d;
                }
            }
            int i;
            for (i = 8; i < nfull; i += 8) {
                load_m(i, mp);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xr + r * DP + i), x1 = *reinterpret_cast<const float4 *>(xr + r * DP + i + 4);
                    const f32x2_t xp[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x2_t d = pk_sub(mp[q], xp[q]);
                        acc[r][q] += d * d;
                    }
                }
            }
            float res[4];
#pragma unroll
            for (int r = 0; r < 4; r++)
                res[r] = ((acc[r][0].x + acc[r][0].y) + (acc[r][1].x + acc[r][1].y)) + ((acc[r][2].x + acc[r][2].y) + (acc[r][3].x + acc[r][3].y));
            for (; i < D; i++) {
                const float mvi = mr[i];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float delta = mvi - xr[r * DP + i];
                    res[r] += delta * delta;
                }
            }
            // (orderable score bits, ~component): the largest score, the lowest component on ties; maximum over the wave's 64
            // components by butterfly, one atomic per wave and row
#pragma unroll
            for (int r = 0; r < 4; r++) {
                unsigned long long key = 0ull;
                if (live) {
                    const unsigned int bits = __float_as_uint(-res[r]);
                    const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                    key = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned)k);
                }
                for (int o = 32; o > 0; o >>= 1) {
                    const unsigned long long other = __shfl_xor(key, o);
                    key = other > key ? other : key;
                }
                if (lane == 0 && rh * 4 + r < nr && key != 0ull) atomicMax(&ws[q0 + rh * 4 + r], key);
            }
        }
    }
    // ---- the workgroup that finishes last unpacks the (score, component) pairs into the candidates and clears the workspace
    // (round 3 launched k_brute_finish_ls for this: one more kernel boundary, ~5 us of a sweep even when the queue is empty)
    // (EVERY wave's atomics must have been performed before the workgroup's ticket is taken -- a workgroup barrier orders
    // issue, not completion: each wave waits for the acknowledgements of its own (s_waitcnt vmcnt(0); they are device-scope
    // read-modify-writes, performed where the last workgroup's exchanges below read them).  A __threadfence() here instead
    // writes the L2 back once per wave: 4 096 of them made every call with a non-empty queue 72 us, 8 rows or 1 500.)
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (last) {
        for (int q = tid; q < nq; q += BLS_THREADS) {
            const unsigned long long pk = atomicExch(&ws[q], 0ull);      // (device scope: the maxima were formed by atomics of every XCD)
            const unsigned int ord = (unsigned int)(pk >> 32);
            const unsigned int bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
            const int32_t id = cand.queue[q];
            cand.k[id] = (int32_t)(0xffffffffu - (unsigned int)(pk & 0xffffffffu));
            cand.s[id] = (double)__uint_as_float(bits);
        }
        if (tid == 0) atomicExch(ticket, 0u);
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

// ======================================================================================
// Sequential (reference-chain) mode: A1 for the spans of ONE utterance, directly in the reference's arithmetic -- no
// filter, no operand images (which the per-utterance updates would have to refresh: two more launches).  The
// utterance's triangular span table holds <= N (N + 1) / 2 row ids; workgroup g scores them against components
// 32 g .. 32 g + 31 (their means and, chunk by chunk, the spans' rows in LDS), the 32 lanes of a wave half share a
// span, the best (score, lowest index) of the 32 goes to the span's 64-bit key by atomic maximum (32 workgroups per
// address; with 8 components per workgroup and the rows read from memory, lane by lane, the kernel took 32 us) -- key = ordered score bits << 32 | ~index, so that the maximum is
// np.argmax's first maximum -- and the LAST workgroup to finish unpacks the keys into cand.k / cand.s and clears them
// for the next utterance.  float32 data (a float64 score does not fit beside its index).
// keys [dev] uint64 [tri_max + 2]: [0..tri) the spans' keys, [tri_max] the arrival counter.
// ======================================================================================
#define SEQ_CPB 32          /* components per workgroup */
#define SEQ_JCH 128         /* valid spans per LDS chunk */
#define SEQ_MAXTRI 2048     /* span table entries the kernel compacts in one go (N_max <= 63) */
__global__ __launch_bounds__(1024) void k_seq_score(segk_corpus c, segk_kmeans m, int utt, int tri_max, segk_cand cand,
                                                    unsigned long long *keys)
{
    extern __shared__ __attribute__((aligned(16))) float s_buf[];        // [SEQ_CPB][D + 1] means, then [SEQ_JCH][LDX] rows
    __shared__ int s_last, s_nv;
    __shared__ short s_j[SEQ_MAXTRI];                                     // valid spans: position in the table
    __shared__ int32_t s_id[SEQ_MAXTRI];                                  // ... and embedding row
    const int tid = threadIdx.x, D = c.D;
    const bool vec4 = (D & 3) == 0 && (c.ldx & 3) == 0;
    // stride of the means in LDS: D itself when rows are read 16 bytes at a time (D / 4 odd or even, the 16 lanes of a span
    // fall on distinct 16-byte slots as long as D / 4 is odd; D + 4 otherwise), D + 1 for 4-byte reads
    const int LDM = vec4 ? (((D >> 2) & 1) ? D : D + 4) : D + 1;
    const int LDX = (D + 3) & ~3;
    float *s_means = s_buf, *s_x = s_buf + ((SEQ_CPB * LDM + 3) & ~3);
    const int N = c.lengths[utt], tri = N * (N + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)utt * ((int64_t)c.N_max * (c.N_max + 1) / 2);
    const float *X = (const float *)c.X;
    const float *means = (const float *)m.means;
    const int k0 = blockIdx.x * SEQ_CPB;
    if (tid == 0) s_nv = 0;
    __syncthreads();
    // the valid spans of the table, compacted (their order is immaterial); every load of a thread in flight together
    {
        int32_t idv[SEQ_MAXTRI / 1024];
#pragma unroll
        for (int q = 0; q < SEQ_MAXTRI / 1024; q++) {
            const int j = q * 1024 + tid;
            idv[q] = j < tri ? vid[j] : -1;
        }
#pragma unroll
        for (int q = 0; q < SEQ_MAXTRI / 1024; q++)
            if (idv[q] >= 0) {
                const int pos = atomicAdd(&s_nv, 1);
                s_j[pos] = (short)(q * 1024 + tid);
                s_id[pos] = idv[q];
            }
    }
    // this workgroup's means
    if (vec4) {
        const int D4 = D >> 2;
        for (int q = tid; q < SEQ_CPB * D4; q += blockDim.x) {
            const int cc = q / D4, d4 = q - cc * D4, kk = k0 + cc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < m.K_max) v = *reinterpret_cast<const float4 *>(means + (int64_t)kk * D + 4 * d4);
            *reinterpret_cast<float4 *>(s_means + cc * LDM + 4 * d4) = v;
        }
    } else {
        for (int q = tid; q < SEQ_CPB * D; q += blockDim.x) {
            const int cc = q / D, kk = k0 + cc;
            s_means[cc * LDM + (q - cc * D)] = kk < m.K_max ? means[(int64_t)kk * D + (q - cc * D)] : 0.f;
        }
    }
    __syncthreads();
    const int nv = s_nv;
    for (int v0 = 0; v0 < nv; v0 += SEQ_JCH) {
        const int nj = nv - v0 < SEQ_JCH ? nv - v0 : SEQ_JCH;
        if (v0 > 0) __syncthreads();
        // the chunk's rows into LDS: consecutive threads on consecutive 16 bytes of a row, four loads in flight per thread
        if (vec4) {
            const int D4 = D >> 2, tot = nj * D4;
            for (int q0 = tid; q0 < tot; q0 += 4 * blockDim.x) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * blockDim.x, qq = q < tot ? q : 0;
                    const int jl = qq / D4, d4 = qq - jl * D4;
                    v[u] = *reinterpret_cast<const float4 *>(X + (int64_t)s_id[v0 + jl] * c.ldx + 4 * d4);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * blockDim.x;
                    if (q < tot) {
                        const int jl = q / D4, d4 = q - jl * D4;
                        *reinterpret_cast<float4 *>(s_x + jl * LDX + 4 * d4) = v[u];
                    }
                }
            }
        } else {
            for (int q = tid; q < nj * D; q += blockDim.x) {
                const int jl = q / D, d = q - jl * D;
                s_x[jl * LDX + d] = X[(int64_t)s_id[v0 + jl] * c.ldx + d];
            }
        }
        __syncthreads();
        for (int p0 = 0; p0 < nj * SEQ_CPB; p0 += blockDim.x) {
            const int p = p0 + tid, jl = p / SEQ_CPB, cidx = p % SEQ_CPB, k = k0 + cidx;
            unsigned long long key = 0ull;
            if (jl < nj && k < m.K_max) {
                const float sc = (vec4 && D >= 8 && D <= 128) ? neg_sqd_exact_v4(s_means + cidx * LDM, s_x + jl * LDX, D)
                                                              : neg_sqd_exact<float>(s_means + cidx * LDM, s_x + jl * LDX, D);
                const unsigned int bits = __float_as_uint(sc);
                const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                key = ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)k);
            }
            // the components of a span sit on SEQ_CPB adjacent lanes
            for (int o = 1; o < SEQ_CPB; o <<= 1) {
                const unsigned long long other = __shfl_xor(key, o);
                key = other > key ? other : key;
            }
            if (cidx == 0 && jl < nj) atomicMax(&keys[s_j[v0 + jl]], key);
        }
    }
    // last workgroup done: unpack (atomic exchanges read the keys where the atomic maxima were performed, and clear them);
    // every wave waits for the acknowledgements of its own maxima before the barrier in front of the ticket (k_kmeans_brute_ls)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned long long arrived = atomicAdd(&keys[tri_max], 1ull);
        s_last = arrived == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    for (int v = tid; v < nv; v += blockDim.x) {
        const unsigned long long key = atomicExch(&keys[s_j[v]], 0ull);
        const int32_t id = s_id[v];
        const unsigned int ord = (unsigned int)(key >> 32);
        const unsigned int bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
        cand.k[id] = (int32_t)(0xffffffffu - (unsigned int)(key & 0xffffffffu));
        cand.s[id] = (double)__uint_as_float(bits);
    }
    if (tid == 0) keys[tri_max] = 0ull;
}

int segk_launch_seq_score(const segk_corpus *c, const segk_kmeans *m, int utt, const segk_cand *cand, unsigned long long *keys,
                          hipStream_t st)
{
    const int tri_max = c->N_max * (c->N_max + 1) / 2;
    const size_t lds = ((size_t)((SEQ_CPB * (c->D + 4) + 3) & ~3) + (size_t)SEQ_JCH * ((c->D + 3) & ~3)) * sizeof(float);
    SEGK_REQUIRE(lds <= 140 * 1024, "D too large for the sequential score kernel");
    SEGK_REQUIRE(tri_max <= SEQ_MAXTRI, "more than 63 landmarks per utterance: use the per-utterance calls");
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_seq_score, lds));
    hipLaunchKernelGGL(k_seq_score, dim3((m->K_max + SEQ_CPB - 1) / SEQ_CPB), dim3(1024), lds, st, *c, *m, utt, tri_max, *cand, keys);
    return SEGK_OK;
}

extern "C" {

int32_t segk_kmeans_resolve(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                            int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream)
{
    return segk_resolve_on(ctx, c, m, ids, row0, n, cand, status, stream);
}

}  // extern "C"

int segk_resolve_on(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
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
    // components in LDS, a chunk of the queue per workgroup (SEGK_BRUTE_LS=0: one workgroup per four rows)
    if (fused && !ctx->capturing && !(getenv("SEGK_BRUTE_LS") && atoi(getenv("SEGK_BRUTE_LS")) == 0)) {
        if (ctx->brute_ws_cap < n) {
            SEGK_CHECK_HIP(hipStreamSynchronize(st));
            if (ctx->brute_ws) (void)hipFree(ctx->brute_ws);
            ctx->brute_ws = nullptr;
            ctx->brute_ws_cap = 0;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->brute_ws, sizeof(unsigned long long) * (size_t)(n + 1)));      // + the ticket
            SEGK_CHECK_HIP(hipMemsetAsync(ctx->brute_ws, 0, sizeof(unsigned long long) * (size_t)(n + 1), st));      // (on the launch stream: a plain
                                                                        // hipMemset is not ordered before kernels of another stream)
            ctx->brute_ws_cap = n;
        }
        const int n_slices = (segk_n_tiles(m->K_max) + BLS_TPS - 1) / BLS_TPS;
        int n_chunks = 256 / n_slices;
        if (n_chunks < 1) n_chunks = 1;
        const int64_t max_chunks = (n + BLS_ROWS - 1) / BLS_ROWS;
        if (n_chunks > max_chunks) n_chunks = (int)max_chunks;
        const size_t lds = ((size_t)BLS_TPS * segk_G(c->D) * 128 + (size_t)BLS_ROWS * ((c->D + 3) & ~3)) * sizeof(float);
        SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_kmeans_brute_ls, lds));
        hipLaunchKernelGGL(k_kmeans_brute_ls, dim3((unsigned)(n_slices * n_chunks)), dim3(BLS_THREADS), lds, st, *c, *m, *cand, (int)c->n_emb,
                           status ? status + 1 : nullptr, n_slices, n_chunks, ctx->brute_ws, (unsigned int *)(ctx->brute_ws + ctx->brute_ws_cap));
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
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

extern "C" {

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

}  // extern "C"
