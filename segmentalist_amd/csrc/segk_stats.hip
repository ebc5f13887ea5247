// segk_stats.hip -- A11 sufficient statistics: sequential (reference order) and batch-synchronous (fixed summation tree)
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"


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


extern "C" {

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
    rc = segk_kmeans_prepare_impl(ctx, c, m, stream, /* mnorm_max already zero */ nslot > 0);
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

}  // extern "C"
