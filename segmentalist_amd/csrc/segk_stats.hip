// segk_stats.hip -- A11 sufficient statistics: sequential (reference order) and batch-synchronous (fixed summation tree)
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"


// ======================================================================================
// A11 sequential: del_item / add_item / clean_components for ONE utterance, one workgroup,
// thread d owns dimension d (kmeans_components.py:93-166, 263-266).
// ======================================================================================
template <typename XT>
__device__ void dev_del_component(const segk_corpus &c, segk_kmeans &m, int k, int *shK, int32_t *relog = nullptr)
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
        // (relog: the relabelling K -> k is only LOGGED -- one workgroup scanning a million labels is 1.3 ms; the caller applies
        // the logged pairs, in order, with the whole chip: k_kmeans_relabel_log)
        if (relog) {
            if (tid == 0) {
                const int n = relog[0];
                relog[1 + 2 * n] = K;
                relog[2 + 2 * n] = k;
                relog[0] = n + 1;
            }
        } else
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
__device__ void dev_clean_components(const segk_corpus &c, segk_kmeans &m, int *shK, int *sh_i, int32_t *relog = nullptr)
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
                dev_del_component<XT>(c, m, k, shK, relog);
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
            dev_del_component<XT>(c, m, w * 32 + bit, shK, relog);
        }
    }
}

// clean_components (kmeans_components.py:263-266) in two launches: the rows, numerators and counts of the emptied components by
// one workgroup, the relabelling pairs (last active row -> emptied row, in the reference's order) logged; then every label
// through the logged pairs in that order, by the whole chip.  relog: [0] pairs, then (from, to) x K_max at most.
template <typename XT>
__global__ __launch_bounds__(256) void k_kmeans_clean_log(segk_corpus c, segk_kmeans m, int32_t *relog)
{
    __shared__ int shK, sh_i;
    if (threadIdx.x == 0) { shK = *m.K; relog[0] = 0; }
    __syncthreads();
    dev_clean_components<XT>(c, m, &shK, &sh_i, relog);
    __syncthreads();
    if (threadIdx.x == 0) *m.K = shK;
}

__global__ __launch_bounds__(256) void k_kmeans_relabel_log(int32_t *assignments, int64_t n, const int32_t *relog)
{
    __shared__ int32_t pf[256], pt[256];
    const int np = relog[0];
    if (np == 0) return;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int a = e < n ? assignments[e] : -1;
    const int a0 = a;
    for (int p0 = 0; p0 < np; p0 += 256) {
        __syncthreads();
        if (p0 + (int)threadIdx.x < np) { pf[threadIdx.x] = relog[1 + 2 * (p0 + threadIdx.x)]; pt[threadIdx.x] = relog[2 + 2 * (p0 + threadIdx.x)]; }
        __syncthreads();
        const int m = np - p0 < 256 ? np - p0 : 256;
        for (int i = 0; i < m; i++)
            if (a == pf[i]) a = pt[i];
    }
    if (e < n && a != a0) assignments[e] = a;
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
// A11 sequential, one utterance, the touched components staged in LDS (segk_kmeans_sequential_sweep).  k_kmeans_update
// walks the items one after the other and pays a global-memory round trip or two for every one of them (~20 us for an
// utterance's ~16 items); here the items' rows, the touched components' numerators and counts are fetched together
// (three round trips for the whole utterance), the del_item / add_item sequence of the reference (kmeans_components.py:
// 93-132, the `k > K -> K` clamp included) is applied in LDS in exactly its order -- per component the same float64
// additions, the same quotient cast to the dtype of X after every item -- and the rows are written back once.
// clean_components (:263-266) runs afterwards on the global state, and only if a touched component emptied.
// ======================================================================================
#define SEQ_MAXOPS 128           /* old + new tokens of an utterance (N_max <= 64) */
template <typename XT>
__global__ __launch_bounds__(256) void k_seq_update(segk_corpus c, segk_kmeans m, int utt, const int32_t *old_tok,
                                                    const int32_t *new_tok, const int32_t *new_k, const int32_t *n_old,
                                                    const int32_t *n_new, int32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char upd_lds[];
    __shared__ int32_t op_e[SEQ_MAXOPS], op_k[SEQ_MAXOPS], op_ci[SEQ_MAXOPS];      // item, component (-1: no-op), slot of its component
    __shared__ int32_t op_a[SEQ_MAXOPS];                                            // the item's assignment before this utterance
    __shared__ int32_t uq_k[SEQ_MAXOPS];                                            // the touched components
    __shared__ long long uq_cnt[SEQ_MAXOPS], op_cnt[SEQ_MAXOPS];
    __shared__ int uq_dirty[SEQ_MAXOPS];
    __shared__ int shK, sh_i, n_uq, any_empty;
    __shared__ int64_t sh_l;
    const int tid = threadIdx.x, nt = blockDim.x, D = c.D;
    const int no = n_old[utt], nn = n_new[utt], nops = no + nn;
    const XT *X = (const XT *)c.X;
    XT *means = (XT *)m.means;
    double *numer = reinterpret_cast<double *>(upd_lds);                            // [n_uq][D]
    // (1) the items and, for the old ones, their current components -- side by side
    if (tid < nops) {
        const bool is_old = tid < no;
        const int32_t e = is_old ? old_tok[(int64_t)utt * c.N_max + tid] : new_tok[(int64_t)utt * c.N_max + (tid - no)];
        const int32_t a = m.assignments[e];
        op_e[tid] = e;
        op_a[tid] = a;
        op_k[tid] = is_old ? a : new_k[(int64_t)utt * c.N_max + (tid - no)];
    }
    if (tid == 0) shK = *m.K;
    __syncthreads();
    // (2) the reference's bookkeeping.  One thread replays the clamp of add_item over the new items in order (:102-106:
    // labels and K, a dozen register operations); everything that needs a search -- the first item with the same
    // component (its slot), an earlier item with the same row (the assert of :101), the running count of the component
    // after every item -- is done by all items side by side, one lane each
    if (tid == 0) {
        int K = shK;
        for (int q = no; q < nops; q++) {
            int k = op_k[q];
            if (k > K) k = K;
            if (k == K) K++;
            op_k[q] = k;
        }
        shK = K;
        any_empty = 0;
    }
    __syncthreads();
    if (tid < 64) {                                         // nops <= 64 (host: N_max <= 32)
        const int q = tid;
        const bool live = q < nops;
        const int k = live ? op_k[q] : -1, e = live ? op_e[q] : -1;
        int first = q, prev_same_e = -1;
        for (int p2 = 0; p2 < nops; p2++) {                 // uniform trip count: every lane reads the same entry (broadcast)
            const int k2 = op_k[p2], e2 = op_e[p2];
            if (p2 < q && k2 == k && p2 < first) first = p2;
            if (p2 < q && e2 == e) prev_same_e = p2;
        }
        const bool is_first = live && k >= 0 && first == q;
        const unsigned long long firsts = __ballot(is_first);
        const int ci = (live && k >= 0) ? __popcll(firsts & ((1ull << first) - 1ull)) : -1;
        if (is_first) uq_k[ci] = k;
        if (live) op_ci[q] = ci;
        if (q == 0) n_uq = __popcll(firsts);
        // add_item's assert: the row must be unassigned -- by an earlier item of this utterance, else by the state before it
        int cur = -1;
        if (live && q >= no) {
            cur = prev_same_e >= 0 ? (prev_same_e < no ? -1 : op_k[prev_same_e]) : op_a[q];
            if (cur != -1) atomicOr(status, 2);
        }
    }
    __syncthreads();
    const int nu = n_uq;
    // (3) numerators and counts of the touched components and the rows of all items into LDS
    const int cap = 2 * c.N_max;                                                    // >= nops >= nu
    XT *mean_l = reinterpret_cast<XT *>(numer + (size_t)cap * D);                   // [cap][D] latest defined mean
    XT *xs = mean_l + (size_t)cap * D;                                              // [cap][D] the items' rows
    // (eight loads of a thread in flight together: a plain loop waits for every load before it issues the next)
    for (int q0 = tid; q0 < nu * D; q0 += 8 * nt) {
        double v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int q = q0 + r * nt, qq = q < nu * D ? q : 0;
            const int u = qq / D, d = qq - u * D;
            v[r] = m.mean_numerators[(int64_t)uq_k[u] * D + d];
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (q0 + r * nt < nu * D) numer[q0 + r * nt] = v[r];
    }
    for (int q0 = tid; q0 < nops * D; q0 += 8 * nt) {
        XT v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int q = q0 + r * nt, qq = q < nops * D ? q : 0;
            const int o = qq / D, d = qq - o * D;
            v[r] = X[(int64_t)op_e[o] * c.ldx + d];
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
            if (q0 + r * nt < nops * D) xs[q0 + r * nt] = v[r];
    }
    if (tid < nu) { uq_cnt[tid] = m.counts[uq_k[tid]]; uq_dirty[tid] = 0; }
    __syncthreads();
    // the count of its component after every item (by the items' lanes: items before it on the same component)
    if (tid < 64 && tid < nops) {
        const int q = tid, ci = op_ci[q];
        long long cn = 0;
        if (ci >= 0) {
            cn = uq_cnt[ci];
            for (int p2 = 0; p2 <= q; p2++)
                if (op_ci[p2] == ci) cn += p2 < no ? -1 : 1;
        }
        op_cnt[q] = cn;
    }
    __syncthreads();
    // (4) the items in the reference's order; thread d owns dimension d of every touched component: no barrier needed
    for (int d = tid; d < D; d += nt)
        for (int q = 0; q < nops; q++) {
            const int ci = op_ci[q];
            if (ci < 0) continue;                           // del_item of an unassigned item: nothing
            const double x = (double)xs[q * D + d];
            const double v = q < no ? numer[ci * D + d] - x : numer[ci * D + d] + x;
            numer[ci * D + d] = v;
            const long long cnt = op_cnt[q];
            if (cnt != 0) mean_l[ci * D + d] = (XT)(v / (double)cnt);               // (:110, :128-129)
        }
    if (tid < nops && op_ci[tid] >= 0) {
        if (op_cnt[tid] != 0) uq_dirty[op_ci[tid]] = 1;
        // the last item of a component leaves its final count
        bool last = true;
        for (int p2 = tid + 1; p2 < nops; p2++)
            if (op_ci[p2] == op_ci[tid]) last = false;
        if (last) uq_cnt[op_ci[tid]] = op_cnt[tid];
    }
    __syncthreads();
    // (5) write back
    for (int q = tid; q < nu * D; q += nt) {
        const int u = q / D, d = q - u * D;
        m.mean_numerators[(int64_t)uq_k[u] * D + d] = numer[q];
        if (uq_dirty[u]) means[(int64_t)uq_k[u] * D + d] = mean_l[q];
    }
    if (tid < nu) {
        m.counts[uq_k[tid]] = uq_cnt[tid];
        if (uq_cnt[tid] == 0 && uq_k[tid] < shK) any_empty = 1;
    }
    if (tid < nops) {
        // final assignment of every item: the last operation on it wins (an item both deleted and added: its new component)
        bool last = true;
        for (int p2 = tid + 1; p2 < nops; p2++)
            if (op_e[p2] == op_e[tid]) last = false;
        if (last) m.assignments[op_e[tid]] = tid < no ? -1 : op_k[tid];
    }
    __threadfence_block();
    __syncthreads();
    // (6) clean_components, on the global state, when something emptied
    if (any_empty) dev_clean_components<XT>(c, m, &shK, &sh_i);
    (void)sh_l;
    __syncthreads();
    if (tid == 0) *m.K = shK;
}

// ======================================================================================
// A11 batch-synchronous statistics (spec: oracle/np_oracle.py kmeans_batch_sweep; rank split:
// oracle/np_dist.py).  Four kernels after the per-utterance kernel:
//
//   k_batch_sort       one workgroup per statistics block: stable counting sort of the block's tokens by
//                      component, straight from the slot arrays new_tok / new_k [U, N_max] (unused slots carry
//                      k = -1); tokens whose argmax is an INACTIVE row (k >= K: they will found new components)
//                      are listed instead, per block and in token order, as (slot, k, embedding row) triples.
//   k_batch_partials   per statistics block and component the sequential fp64 sum of its tokens in token order.
//   [multi-GPU: ONE all-gather of the packed per-rank record -- partial sums, totals, counts, flag lists]
//   k_batch_finalize   one workgroup per eight components.  Every workgroup replays, redundantly and from
//                      the same inputs, the `k > K -> K` clamp over the flagged tokens of all blocks in global
//                      token order (kmeans_components.py:102-106), the combined counts and the bookkeeping
//                      of clean_components (:263-266: which original row ends in which final row); then it
//                      writes ITS final rows -- fixed binary tree over the blocks' partial sums (new
//                      components: sequential sums of their flagged tokens per block, same tree), means =
//                      numerators / counts -- and their part of the fp32 MFMA image.
//   k_batch_post       final labels of the local tokens (remap table), the split-precision image of every tile
//                      (needs max |m|^2 over ALL rows, complete only now) and the duplicate marking.
//
// Packed record of one rank, in 8-byte words (nbl = blocks per rank, FW = flag words per block):
//   [part_sum nbl*K_max*D double][part_tot nbl double][part_cnt nbl*K_max int64][flags nbl*FW]
//   flags of a block, as int32: {count, 0, (slot, k, row) x cap}.
// ======================================================================================
static inline __host__ __device__ int64_t segk_flag_words(int cap) { return (2 + 3 * (int64_t)cap + 1) / 2; }

// ======================================================================================
// (1a) k_batch_sort: ONE workgroup per statistics block.  Stable counting sort of the block's tokens by
//      component: sorted[p0 + i] = embedding row of the i-th token in (component, token order) order,
//      koff[b][k] = index of component k's first token (koff[b][K_max] = number of un-flagged tokens), so
//      that the summing kernel reads every (block, component) list directly -- before, each of the 125
//      workgroups of a block scanned all of its 25 000 slots for its own eight components.
//      Waves own contiguous runs of slots: (P1) per-wave histograms in LDS, (P2) per component the waves'
//      counts turned into running offsets and the totals scanned over the components, (P3) every wave
//      walks its run again, 64 slots at a time in order; the rank of a token among the same-component tokens
//      of its 64 is counted explicitly (a dozen iterations: one per valid lane), never left to the order in
//      which the hardware happens to serve atomics.
//      Tokens whose argmax is an inactive row (k >= K, flagged) are listed separately in token order; the
//      block's total is summed in utterance order.  Also zeroes m.mnorm_max and stores K for (2).
// ======================================================================================
#define SORT_THREADS 1024
#define SORT_KEYS_LDS 32768          /* slots of a block whose keys are staged in LDS (int16); larger blocks read them from memory */
// the lanes of the wave whose key equals this lane's (valid lanes only): one ballot per key bit
__device__ __forceinline__ unsigned long long dev_match_key(int k, bool ok, int nbits)
{
    unsigned long long mask = __ballot(ok);
    for (int bit = 0; bit < nbits; bit++) {
        const unsigned long long bal = __ballot((k >> bit) & 1);
        mask &= ((k >> bit) & 1) ? bal : ~bal;
    }
    return mask;
}

template <int NW>       // waves that own slot runs (NW * K_max int32 counters in LDS: 16 up to K_max 1024, ... 2 up to 8192)
__device__ __forceinline__ void dev_batch_sort(const segk_corpus &c, const segk_kmeans &m, const int Kb, const int u0, const int u1,
                                               const int32_t *new_k, int32_t *sorted, int32_t *koff,
                                               int32_t *cntw /* [NW][K_max] */, int32_t *base /* [K_max + 2] */,
                                               short *keys /* [SORT_KEYS_LDS] */, int32_t *wtot /* [16] */)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int K_max = m.K_max;
    const int64_t p0 = (int64_t)u0 * c.N_max;
    const int S = (u1 - u0) * c.N_max;
    const bool staged = S <= SORT_KEYS_LDS;
    const int per = ((S + NW - 1) / NW + 63) & ~63;              // slots per wave, whole chunks of 64
    int nbits = 1;
    while ((1 << nbits) < K_max) nbits++;
    SEGK_TSTAMP(1, 0);
    for (int i = tid; i < NW * K_max; i += SORT_THREADS) cntw[i] = 0;
    // (P0) the block's keys into LDS: every load of a thread in flight together (flagged tokens and unused slots as -1)
    if (staged) {
        constexpr int NL = SORT_KEYS_LDS / SORT_THREADS;          // 32 loads per thread at most
        int v[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int s = q * SORT_THREADS + tid;
            v[q] = new_k[p0 + (s < S ? s : 0)];
        }
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int s = q * SORT_THREADS + tid;
            if (s < S) keys[s] = (short)((v[q] >= 0 && v[q] < Kb) ? v[q] : -1);
        }
    }
    __syncthreads();
    SEGK_TSTAMP(1, 1);
    auto key_at = [&](int s) -> int {
        if (staged) return keys[s];
        const int k = new_k[p0 + s];
        return (k >= 0 && k < Kb) ? k : -1;
    };
    // (P1) histograms
    if (wv < NW) {
        int32_t *mine = cntw + wv * K_max;
        const int s0 = wv * per, s1 = s0 + per < S ? s0 + per : S;
        for (int s = s0 + lane; s < s1; s += 64) {
            const int k = key_at(s);
            if (k >= 0) atomicAdd(&mine[k], 1);
        }
    }
    __syncthreads();
    SEGK_TSTAMP(1, 2);
    // (P2) per component: the waves' counts -> running offsets; totals -> exclusive scan over the components
    // (K_max <= 8192: at most 8 slabs of 1024 components, thread t owns component 1024 q + t of slab q)
    int carry = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int k0 = q * SORT_THREADS;
        if (k0 >= K_max) break;
        const int k = k0 + tid;
        int run = 0;
        if (k < K_max) {
            int t[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) t[w] = cntw[w * K_max + k];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                cntw[w * K_max + k] = run;
                run += t[w];
            }
        }
        int incl = run;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int woff = 0, slab = 0;
        for (int w = 0; w < SORT_THREADS / 64; w++) {
            if (w < wv) woff += wtot[w];
            slab += wtot[w];
        }
        if (k < K_max) {
            const int ex = carry + woff + incl - run;
            base[k] = ex;
            koff[k] = ex;
        }
        carry += slab;
        __syncthreads();
    }
    if (tid == 0) koff[K_max] = carry;
    SEGK_TSTAMP(1, 3);
    // (P3) placement, stable: `sorted` receives the slot's offset in the block (its embedding row is new_tok[p0 + offset])
    if (wv < NW) {
        int32_t *mine = cntw + wv * K_max;
        const int s0 = wv * per, s1 = s0 + per < S ? s0 + per : S;
        for (int sb = s0; sb < s1; sb += 64) {
            const int s = sb + lane;
            const int k = s < s1 ? key_at(s) : -1;
            const bool ok = k >= 0;
            const unsigned long long same = dev_match_key(k, ok, nbits);
            if (ok) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                const int before = mine[k];
                sorted[p0 + base[k] + before + rank] = s;
                if (rank == __popcll(same) - 1) mine[k] = before + rank + 1;       // the last of its component in the chunk
            }
        }
    }
    SEGK_TSTAMP_MAX(1, 4);
}

// grid = 2 * n_blocks: workgroup b < n_blocks sorts block b; workgroup n_blocks + b lists the block's flagged
// tokens and sums its totals (a sequential fp64 chain of one thread: beside the sort, not behind it)
__global__ __launch_bounds__(SORT_THREADS) void k_batch_sort(
    segk_corpus c, segk_kmeans m, const int32_t *blk_lo, int n_blocks, const int32_t *new_tok, const int32_t *new_k,
    const int32_t *n_flag, const double *out_total, int32_t *sorted, int32_t *koff_all, double *part_tot, int32_t *flags,
    int cap, double *out_scalars)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    __shared__ int32_t wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Kb = *m.K;                          // active components before the sweep: k >= Kb is a flagged token
    const int K_max = m.K_max;
    if ((int)blockIdx.x < n_blocks) {
        const int b = blockIdx.x;
        const int u0 = blk_lo[b], u1 = blk_lo[b + 1];
        if (b == 0 && tid == 0) {
            *(double *)m.mnorm_max = 0.0;         // max |m|^2 of the finalize kernel's prepare
            out_scalars[3] = (double)Kb;          // the finalize kernel's workgroups read K from here: one of them rewrites *m.K
        }
        int32_t *base = reinterpret_cast<int32_t *>(sort_lds);
        int32_t *cntw = base + K_max + 2;
        int32_t *koff = koff_all + (int64_t)b * (K_max + 1);
        if (K_max <= 1024) {
            short *keys = reinterpret_cast<short *>(cntw + 16 * K_max);
            dev_batch_sort<16>(c, m, Kb, u0, u1, new_k, sorted, koff, cntw, base, keys, wtot);
        } else if (K_max <= 2048) {
            short *keys = reinterpret_cast<short *>(cntw + 8 * K_max);
            dev_batch_sort<8>(c, m, Kb, u0, u1, new_k, sorted, koff, cntw, base, keys, wtot);
        } else if (K_max <= 4096) {
            short *keys = reinterpret_cast<short *>(cntw + 4 * K_max);
            dev_batch_sort<4>(c, m, Kb, u0, u1, new_k, sorted, koff, cntw, base, keys, wtot);
        } else {
            short *keys = reinterpret_cast<short *>(cntw + 2 * K_max);
            dev_batch_sort<2>(c, m, Kb, u0, u1, new_k, sorted, koff, cntw, base, keys, wtot);
        }
        return;
    }
    const int b = blockIdx.x - n_blocks;
    const int u0 = blk_lo[b], u1 = blk_lo[b + 1];
    double *stage = reinterpret_cast<double *>(sort_lds);           // [2048]
    SEGK_TSTAMP(1, 0);
    // ---- flagged tokens: rare (none once every component is active), so a plain ordered pass
    int32_t *fl = flags + (int64_t)b * 2 * segk_flag_words(cap);
    int nf = 0;
    for (int u = u0 + tid; u < u1; u += SORT_THREADS) nf += n_flag[u];
    for (int o = 32; o > 0; o >>= 1) nf += __shfl_xor(nf, o);
    if (lane == 0) wtot[wv] = nf;
    __syncthreads();
    nf = 0;
    for (int w = 0; w < SORT_THREADS / 64; w++) nf += wtot[w];
    __syncthreads();
    int written = 0;                               // workgroup-uniform
    if (nf > 0) {
        const int64_t p0 = (int64_t)u0 * c.N_max, p1 = (int64_t)u1 * c.N_max;
        for (int64_t pb = p0; pb < p1; pb += SORT_THREADS) {
            const int64_t p = pb + tid;
            const int k = p < p1 ? new_k[p] : -1;
            const int f = k >= Kb;
            const unsigned long long bal = __ballot(f);
            if (lane == 0) wtot[wv] = __popcll(bal);
            __syncthreads();
            int off = written, tot = 0;
            for (int w = 0; w < SORT_THREADS / 64; w++) {
                if (w < wv) off += wtot[w];
                tot += wtot[w];
            }
            off += __popcll(bal & ((1ull << lane) - 1ull));
            if (f && off < cap) {
                fl[2 + 3 * off + 0] = (int32_t)p;
                fl[2 + 3 * off + 1] = k;
                fl[2 + 3 * off + 2] = new_tok[p];
            }
            written += tot;
            __syncthreads();
        }
    }
    if (tid == 0) { fl[0] = written; fl[1] = 0; }
    SEGK_TSTAMP(1, 1);
    // ---- sequential (utterance order) sum of the block's totals, staged through LDS so that the single summing
    // thread never waits on global memory
    double s = 0.0;
    for (int uc = u0; uc < u1; uc += 2048) {
        const int nu = u1 - uc < 2048 ? u1 - uc : 2048;
        __syncthreads();
        for (int i = tid; i < nu; i += SORT_THREADS) stage[i] = out_total[uc + i];
        __syncthreads();
        if (tid == 0) {
            int i = 0;
            for (; i + 16 <= nu; i += 16) {       // strictly sequential adds; the LDS reads are issued 16 at a time
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = stage[i + q];
#pragma unroll
                for (int q = 0; q < 16; q++) s += v[q];
            }
            for (; i < nu; i++) s += stage[i];
        }
    }
    if (tid == 0) part_tot[b] = s;
    SEGK_TSTAMP(1, 4);
}

// ======================================================================================
// (1a') k_batch_sort_sum (round 3): the same stable counting sort AND the partial sums, spread over the chip.  The
//      one-workgroup-per-block sort spent 20 of its 29 us in the placement pass -- sixteen waves on one CU, every chunk of 64
//      slots a ten-ballot key match -- while 248 CUs idled, and the summing kernel behind it (one wave per (block, component))
//      25 us on workgroup dispatch and three dependent round trips per wave for a dozen tokens.  Here a block is handled by
//      NR + 2 workgroups: workgroup (b, r < NR) owns the components k = r mod NR (at most RS = 32 of them up to K_max 2048: 32 ranges x
//      8 blocks fill the 256 CUs -- a CU takes in ~25 GB/s, so the 32 MB of token rows must be gathered by all of them) and a
//      region of its own in `sorted` (S_b entries: whatever the split of the tokens over the ranges, a region cannot
//      overflow), so nothing crosses workgroups:
//      (P1) every wave compacts the in-range tokens of its run of slots into LDS (ballot + prefix count, order kept) and
//           counts them per component; (P2) offsets; (P3) the placement pass walks the compacted list -- a thirty-second of
//           the tokens, five key bits -- and writes the tokens' embedding rows into the region;
//      (S)  the range's tokens now lie contiguous in the region, component after component: wave w takes the components
//           whose lists START in the w-th sixteenth of them and streams their rows 32 at a time, whatever the component
//           boundaries -- the sums stay strictly sequential per component; an accumulator is written out when its list ends.
//      Workgroup (b, NR) lists the flagged tokens (argmax on an inactive row) in token order with the same compaction;
//      workgroup (b, NR + 1) sums the block's totals.  koff2[b][k] = {offset of component k's list in its region, length}.
// ======================================================================================
#define SORT_CL 512                  /* compacted entries per wave kept in LDS; a wave with more in-range tokens (or a block of
                                        more than 16 * 2048 slots): the placement pass walks all slots again                */
static inline __host__ __device__ int segk_sort_rsh(int K_max) { return K_max <= 2048 ? 5 : 7; }       // log2 of components per range (at most)
// log2 of the number of ranges (a power of two): component k belongs to range k & (NR - 1) as its member k >> nsh -- strided,
// so that the active components (the low indices: K of K_max) and the inactive ones spread evenly over the ranges' workgroups
static inline __host__ __device__ int segk_sort_nsh(int K_max)
{
    const int need = (K_max + (1 << segk_sort_rsh(K_max)) - 1) >> segk_sort_rsh(K_max);
    int nsh = 0;
    while ((1 << nsh) < need) nsh++;
    return nsh;
}
static inline __host__ __device__ int segk_sort_ranges(int K_max) { return 1 << segk_sort_nsh(K_max); }

__global__ __launch_bounds__(SORT_THREADS) void k_batch_sort_sum(
    segk_corpus c, segk_kmeans m, const int32_t *blk_lo, int n_blocks, const int32_t *new_tok, const int32_t *new_k,
    const double *out_total, int32_t *sorted, int32_t *koff2, double *part_tot, int32_t *flags, int cap, double *out_scalars, int NR,
    int rsh, int nsh, double *part_sum, int64_t *part_cnt, int fuse_sum)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    __shared__ int32_t wcount[16], wscan[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int Kb = *m.K;                          // active components before the sweep: k >= Kb is a flagged token
    const int K_max = m.K_max;
    const int RS = 1 << rsh, kmask = RS - 1;
    const int b = blockIdx.x / (NR + 2), r = blockIdx.x % (NR + 2);
    const int u0 = blk_lo[b], u1 = blk_lo[b + 1];
    SEGK_TSTAMP(1, 0);
    if (r == NR + 1) {
        if (b == 0 && tid == 0) {
            *(double *)m.mnorm_max = 0.0;         // max |m|^2 of the finalize kernel's prepare
            out_scalars[3] = (double)Kb;          // the finalize kernel's workgroups read K from here: one of them rewrites *m.K
        }
        // ---- sequential (utterance order) sum of the block's totals, staged through LDS so that the single summing
        // thread never waits on global memory
        double *stage = reinterpret_cast<double *>(sort_lds);           // [2048]
        double s = 0.0;
        for (int uc = u0; uc < u1; uc += 2048) {
            const int nu = u1 - uc < 2048 ? u1 - uc : 2048;
            __syncthreads();
            for (int i = tid; i < nu; i += SORT_THREADS) stage[i] = out_total[uc + i];
            __syncthreads();
            if (tid == 0) {
                int i = 0;
                for (; i + 16 <= nu; i += 16) {       // strictly sequential adds; the LDS reads are issued 16 at a time
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = stage[i + q];
#pragma unroll
                    for (int q = 0; q < 16; q++) s += v[q];
                }
                for (; i < nu; i++) s += stage[i];
            }
        }
        if (tid == 0) part_tot[b] = s;
        SEGK_TSTAMP(1, 4);
        return;
    }
    const bool flg = r == NR;
    const int64_t p0 = (int64_t)u0 * c.N_max;
    const int S = (u1 - u0) * c.N_max;
    const int per = ((S + 15) / 16 + 63) & ~63;                  // slots per wave, whole chunks of 64
    const bool preload = per <= 2048;
    int32_t *cntw = reinterpret_cast<int32_t *>(sort_lds);       // [16][RS]
    int32_t *base = cntw + 16 * RS;                              // [RS]
    int32_t *cnts = base + RS;                                   // [RS]
    uint32_t *cl = reinterpret_cast<uint32_t *>(cnts + RS) + (size_t)wv * SORT_CL;      // this wave's compacted entries
    const int32_t *keys = new_k + (S > 0 ? p0 : 0);            // (an empty block at the end of the corpus: nothing is read)
    const int s0 = wv * per < S ? wv * per : S, s1 = s0 + per < S ? s0 + per : S;
    auto in_range = [&](int k) -> bool { return flg ? k >= Kb : (k >= 0 && k < Kb && (k & (NR - 1)) == r); };
    for (int i = tid; i < 16 * RS; i += SORT_THREADS) cntw[i] = 0;
    int n_mine = 0;                                              // wave-uniform: in-range tokens of this wave's run
    if (preload) {
        // (P1) every key of the run fetched up front (unconditional loads, clamped index), then compacted in order
        constexpr int NC = 2048 / 64;
        int v[NC];
#pragma unroll
        for (int q = 0; q < NC; q++) {
            const int sidx = s0 + q * 64 + lane;
            v[q] = keys[sidx < s1 ? sidx : (S > 0 ? S - 1 : 0)];
        }
        __syncthreads();                                         // the zeroed counters
#pragma unroll
        for (int q = 0; q < NC; q++) {
            if (s0 + q * 64 >= s1) break;                        // (wave-uniform: the run's last chunk is behind)
            const int sidx = s0 + q * 64 + lane;
            const int k = sidx < s1 ? v[q] : -1;
            const bool in = sidx < s1 && in_range(k);
            const unsigned long long bal = __ballot(in);
            if (in) {
                const int at = n_mine + __popcll(bal & ((1ull << lane) - 1ull));
                if (at < SORT_CL) cl[at] = ((unsigned int)sidx << rsh) | (unsigned int)((k >> nsh) & kmask);
                if (!flg) atomicAdd(&cntw[wv * RS + ((k >> nsh) & kmask)], 1);
            }
            n_mine += __popcll(bal);
        }
    } else {
        __syncthreads();
        for (int sb = s0; sb < s1; sb += 64) {
            const int sidx = sb + lane;
            const int k = sidx < s1 ? keys[sidx] : -1;
            const bool in = sidx < s1 && in_range(k);
            if (in && !flg) atomicAdd(&cntw[wv * RS + ((k >> nsh) & kmask)], 1);
            n_mine += __popcll(__ballot(in));
        }
    }
    if (lane == 0) wcount[wv] = n_mine;
    const bool compact = !__syncthreads_or(!preload || n_mine > SORT_CL);      // workgroup-uniform (and the barrier after P1)
    SEGK_TSTAMP(1, 2);
    if (flg) {
        // ---- the flagged tokens, in token order: {count, 0, (slot, k, row) x cap}
        int32_t *fl = flags + (int64_t)b * 2 * segk_flag_words(cap);
        int off = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wv) off += wcount[w];
            tot += wcount[w];
        }
        if (tid == 0) { fl[0] = tot; fl[1] = 0; }
        if (compact) {
            for (int i = lane; i < n_mine; i += 64) {
                const int sidx = (int)(cl[i] >> rsh);
                if (off + i < cap) {
                    fl[2 + 3 * (off + i) + 0] = (int32_t)(p0 + sidx);
                    fl[2 + 3 * (off + i) + 1] = keys[sidx];
                    fl[2 + 3 * (off + i) + 2] = new_tok[p0 + sidx];
                }
            }
        } else {
            int done = 0;
            for (int sb = s0; sb < s1 && done < n_mine; sb += 64) {
                const int sidx = sb + lane;
                const int k = sidx < s1 ? keys[sidx] : -1;
                const bool in = sidx < s1 && k >= Kb;
                const unsigned long long bal = __ballot(in);
                const int at = off + done + __popcll(bal & ((1ull << lane) - 1ull));
                if (in && at < cap) {
                    fl[2 + 3 * at + 0] = (int32_t)(p0 + sidx);
                    fl[2 + 3 * at + 1] = k;
                    fl[2 + 3 * at + 2] = new_tok[p0 + sidx];
                }
                done += __popcll(bal);
            }
        }
        SEGK_TSTAMP_MAX(1, 4);
        return;
    }
    // (P2) per component of the range: the waves' counts -> running offsets; totals -> exclusive scan over the range
    int run = 0, incl = 0;
    if (tid < RS) {
        int t[16];
#pragma unroll
        for (int w = 0; w < 16; w++) t[w] = cntw[w * RS + tid];
#pragma unroll
        for (int w = 0; w < 16; w++) {
            cntw[w * RS + tid] = run;
            run += t[w];
        }
        incl = run;
        for (int o = 1; o < 64; o <<= 1) {
            const int t2 = __shfl_up(incl, o);
            if (lane >= o) incl += t2;
        }
        if (lane == 63) wscan[wv] = incl;                        // (RS = 128: two waves)
    }
    __syncthreads();
    if (tid < RS) {
        const int ex = (wv == 1 ? wscan[0] : 0) + incl - run;
        base[tid] = ex;
        cnts[tid] = run;
        const int k = (tid << nsh) + r;
        if (k < K_max) {
            koff2[((int64_t)b * K_max + k) * 2 + 0] = ex;
            koff2[((int64_t)b * K_max + k) * 2 + 1] = run;
        }
    }
    __syncthreads();
    SEGK_TSTAMP(1, 3);
    // (P3) placement, stable: `sorted` receives the token's embedding row (new_tok of its slot)
    int32_t *region = sorted + p0 * NR + (int64_t)r * S;
    int32_t *mine = cntw + wv * RS;
    if (compact) {
        for (int ib = 0; ib < n_mine; ib += 64) {
            const int i = ib + lane;
            const bool ok = i < n_mine;
            const unsigned int e = ok ? cl[i] : 0u;
            const int kk = (int)(e & (unsigned int)kmask);
            const int row = new_tok[p0 + (ok ? (int)(e >> rsh) : 0)];         // (issued before the key match: off its critical path)
            const unsigned long long same = dev_match_key(kk, ok, rsh);
            if (ok) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                const int before = mine[kk];
                region[base[kk] + before + rank] = row;
                if (rank == __popcll(same) - 1) mine[kk] = before + rank + 1;       // the last of its component in the chunk
            }
        }
    } else {
        for (int sb = s0; sb < s1; sb += 64) {
            const int sidx = sb + lane;
            const int k = sidx < s1 ? keys[sidx] : -1;
            const bool ok = sidx < s1 && in_range(k);
            if (__ballot(ok) == 0ull) continue;
            const int kk = (k >> nsh) & kmask;
            const unsigned long long same = dev_match_key(kk, ok, rsh);
            if (ok) {
                const int rank = __popcll(same & ((1ull << lane) - 1ull));
                const int before = mine[kk];
                region[base[kk] + before + rank] = new_tok[p0 + sidx];
                if (rank == __popcll(same) - 1) mine[kk] = before + rank + 1;
            }
        }
    }
    SEGK_TSTAMP_MAX(1, 4);
    if (!fuse_sum) return;
    // (S) float32 rows of even D <= 128; everything else: k_batch_partials
    // (the barrier orders this workgroup's stores to `region` before its own loads of them: workgroup scope is all that is
    // needed -- one CU, one vector L1 that has never held these lines.  An agent-scope __threadfence() here writes back and
    // invalidates the XCD's L2 per workgroup: +77 us on the sweep)
    __syncthreads();
    {
        const int T = base[RS - 1] + cnts[RS - 1];
        auto owner = [&](int kk) -> int {
            const int o = T > 0 ? (int)(((long long)base[kk] * 16) / T) : 0;
            return o < 15 ? o : 15;
        };
        // the wave's components [ka, kb): owner() is monotone in kk
        const unsigned long long m0 = __ballot(lane < RS && owner(lane < RS ? lane : 0) == wv),
                                 m1 = __ballot(64 + lane < RS && owner(64 + lane < RS ? 64 + lane : 0) == wv);
        if ((m0 | m1) == 0ull) return;
        const int ka = m0 ? __ffsll((long long)m0) - 1 : 64 + __ffsll((long long)m1) - 1;
        const int kb = m1 ? 128 - __clzll((long long)m1) : 64 - __clzll((long long)m0);
        const int D = c.D;
        const float *Xf = (const float *)c.X;
        const int dl = 2 * lane < D ? 2 * lane : 0;              // clamped: always a valid address
        const int t_lo = base[ka], t_hi = base[kb - 1] + cnts[kb - 1];
        double a0 = 0.0, a1 = 0.0;
        int kk = ka, rem = cnts[ka];
        auto flush = [&]() {                                     // component kk is complete (rem == 0): write it out, move on
            for (;;) {
                const int k = (kk << nsh) + r;
                if (k < K_max) {
                    if (2 * lane < D) *reinterpret_cast<double2 *>(part_sum + ((int64_t)b * K_max + k) * D + 2 * lane) = make_double2(a0, a1);
                    if (lane == 0) part_cnt[(int64_t)b * K_max + k] = cnts[kk];
                }
                a0 = 0.0;
                a1 = 0.0;
                kk++;
                if (kk >= kb) { rem = 0x7fffffff; return; }
                rem = cnts[kk];
                if (rem > 0) return;
            }
        };
        if (rem == 0) flush();
        for (int c0 = t_lo; c0 < t_hi; c0 += 64) {
            const int nb = t_hi - c0 < 64 ? t_hi - c0 : 64;
            const int mine_row = region[c0 + (lane < nb ? lane : 0)];
            for (int q0 = 0; q0 < nb; q0 += 32) {
                float2 xv[32];
#pragma unroll
                for (int q = 0; q < 32; q++) {
                    const int e = __shfl(mine_row, q0 + q < nb ? q0 + q : 0);       // clamped: always a valid row
                    xv[q] = *reinterpret_cast<const float2 *>(Xf + (int64_t)e * c.ldx + dl);
                }
#pragma unroll
                for (int q = 0; q < 32; q++) {
                    if (q0 + q < nb) {                                               // wave-uniform
                        a0 += (double)xv[q].x;
                        a1 += (double)xv[q].y;
                        if (--rem == 0) flush();
                    }
                }
            }
        }
        SEGK_TSTAMP_MAX(1, 5);
    }
}

// (1b) k_batch_partials: what k_batch_sort_sum's summing phase does not cover (float64 data, odd D, D > 128).  Per (block,
//      component) the sequential fp64 sum of its tokens in token order: one wave per pair, lanes own dimensions; the list's rows
//      are fetched 64 at a time (one per lane), their elements 16 rows at a time (unconditional loads, clamped index, select
//      after the load: all 16 in flight together) and added strictly in order.
#define PART_WAVES 4
template <typename XT>
__global__ __launch_bounds__(64 * PART_WAVES) void k_batch_partials(segk_corpus c, segk_kmeans m, const int32_t *blk_lo, int n_blocks,
                                                                   const int32_t *sorted, const int32_t *koff2,
                                                                   double *part_sum, int64_t *part_cnt, int NR)
{
    const int groups = (m.K_max + PART_WAVES - 1) / PART_WAVES;
    const int b = blockIdx.x / groups, kg = blockIdx.x % groups;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = kg * PART_WAVES + wv;
    SEGK_TSTAMP(2, 0);
    if (k >= m.K_max) return;
    const int D = c.D;
    const XT *X = (const XT *)c.X;
    const int2 on = *reinterpret_cast<const int2 *>(koff2 + ((int64_t)b * m.K_max + k) * 2);
    const int nm = on.y;
#ifdef SEGK_STAMP
    if (segk_tstamp_buf && lane == 0 && (unsigned long long)nm > *SEGK_TSTAMP_AT(2, 2)) *SEGK_TSTAMP_AT(2, 2) = (unsigned long long)nm;
#endif
    const int64_t p0 = (int64_t)blk_lo[b] * c.N_max;
    const int S = (blk_lo[b + 1] - blk_lo[b]) * c.N_max;
    const int32_t *list = sorted + p0 * NR + (int64_t)(k & (NR - 1)) * S + on.x;    // embedding rows of the list's tokens, token order
    double *out = part_sum + ((int64_t)b * m.K_max + k) * D;
    if (lane == 0) part_cnt[(int64_t)b * m.K_max + k] = nm;
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
        for (int c0 = 0; c0 < nm; c0 += 64) {
            const int mine = list[c0 + lane < nm ? c0 + lane : c0];
            const int nb = nm - c0 < 64 ? nm - c0 : 64;
            for (int q0 = 0; q0 < nb; q0 += 16) {
                XT xv[16][MAXR];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int e = __shfl(mine, q0 + q < nb ? q0 + q : 0);      // clamped: always a valid row
#pragma unroll
                    for (int r = 0; r < MAXR; r++) xv[q][r] = X[(int64_t)e * c.ldx + dcl[r]];
                }
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const bool ok = q0 + q < nb;
#pragma unroll
                    for (int r = 0; r < MAXR; r++) acc[r] += ok ? (double)xv[q][r] : 0.0;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < MAXR; r++) {
            int d = d0 + r * 64 + lane;
            if (d < D) out[d] = acc[r];
        }
    }
    SEGK_TSTAMP_MAX(2, 1);
}

// Record of block b inside the all-gathered buffer: `nbl` blocks per rank, ranks `rank_stride` words apart.
struct PackAddr {
    int nbl, K_max, D, cap;
    int64_t rank_stride;
    int64_t row_words;      // words per block of the flagged tokens' rows carried in the record (0: the rows are read from X)
    __device__ __forceinline__ int64_t rank_base(int b) const { return (int64_t)(b / nbl) * rank_stride; }
    __device__ __forceinline__ int64_t sum(int b) const { return rank_base(b) + (int64_t)(b % nbl) * K_max * D; }
    __device__ __forceinline__ int64_t tot(int b) const { return rank_base(b) + (int64_t)nbl * K_max * D + (b % nbl); }
    __device__ __forceinline__ int64_t cnt(int b) const
    {
        return rank_base(b) + (int64_t)nbl * K_max * D + nbl + (int64_t)(b % nbl) * K_max;
    }
    __device__ __forceinline__ int64_t flg(int b) const
    {
        return rank_base(b) + (int64_t)nbl * K_max * D + nbl + (int64_t)nbl * K_max + (int64_t)(b % nbl) * segk_flag_words(cap);
    }
    __device__ __forceinline__ int64_t rows(int b) const       // behind the flag lists of all the rank's blocks
    {
        return rank_base(b) + (int64_t)nbl * K_max * D + nbl + (int64_t)nbl * K_max + (int64_t)nbl * segk_flag_words(cap) + (int64_t)(b % nbl) * row_words;
    }
};

// words of a block's row area when the flagged tokens' embedding rows travel in the record (a rank that holds a shard of the
// corpus cannot read the rows of other ranks' tokens from X): cap rows of D elements
static inline __host__ __device__ int64_t segk_flag_row_words(int cap, int D, int elem) { return ((int64_t)cap * D * elem + 7) / 8; }

// the rows of a block's flagged tokens into the record (only with flag_rows; one workgroup per local block, behind the kernel
// that wrote the flag lists)
template <typename XT>
__global__ __launch_bounds__(256) void k_batch_flag_rows(segk_corpus c, const int32_t *flags, int cap, XT *rows, int64_t row_words)
{
    const int b = blockIdx.x, D = c.D;
    const int32_t *fl = flags + (int64_t)b * 2 * segk_flag_words(cap);
    int n = fl[0];
    if (n > cap) n = cap;
    XT *out = reinterpret_cast<XT *>(reinterpret_cast<double *>(rows) + (int64_t)b * row_words);
    const XT *X = (const XT *)c.X;
    for (int64_t idx = threadIdx.x; idx < (int64_t)n * D; idx += blockDim.x) {
        const int i = (int)(idx / D), d = (int)(idx - (int64_t)i * D);
        out[idx] = X[(int64_t)fl[2 + 3 * i + 2] * c.ldx + d];
    }
}

// balanced binary tree over n <= 64 parts, pairing neighbours level by level, odd one carried
// (np_oracle.tree_sum); v[] is consumed
__device__ __forceinline__ double tree_reduce_d(double *v, int n)
{
    while (n > 1) {
        int o = 0;
        for (int i = 0; i + 1 < n; i += 2) v[o++] = v[i] + v[i + 1];
        if (n & 1) v[o++] = v[n - 1];
        n = o;
    }
    return v[0];
}

#define SEGK_FLAG_LDS 2048        /* flagged tokens of a sweep (all ranks together) the finalize kernel keeps in LDS; the ones
                                     beyond go through the context's overflow arrays in global memory (identical values
                                     written by every workgroup) -- the only limit left is flag_cap per block              */
#define FIN_ROWS 8                /* final rows per workgroup */
#define FIN_MH 4                  /* ... of which the first FIN_MH are requested together with the records' sums when the list is that short */
#define FIN_ML 128                /* flagged tokens of a new component listed per row (more: the general loop) */

template <typename XT>
__global__ __launch_bounds__(256) void k_batch_finalize(
    segk_corpus c, segk_kmeans m, const double *pack, int n_blocks, int nbl, int64_t rank_stride, int cap, int my_rank,
    int32_t *new_k, int32_t *remap, double *out_scalars, int32_t *status, unsigned long long *row_hash,
    unsigned int *sp_zero_slot, int32_t *ovf, int ovf_cap, int sp_spec, int64_t row_words)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6;
    const int K_max = m.K_max, D = c.D;
    // dynamic LDS: counts [K_max] int32, pos2orig [K_max] u16, holes [K_max] u16, bitmap [K_max/32 + 1] u32, then the
    // workgroup's rows [FIN_ROWS][D] as doubles (the values of `means`, exactly representable)
    int32_t *cnt32 = reinterpret_cast<int32_t *>(fin_lds);
    unsigned short *pos2orig = reinterpret_cast<unsigned short *>(cnt32 + K_max);
    unsigned short *holes = pos2orig + K_max;
    unsigned int *bitmap = reinterpret_cast<unsigned int *>(holes + K_max);      // 8 K_max bytes so far: 4-byte aligned
    double *mrow = reinterpret_cast<double *>(fin_lds + (((size_t)K_max * 8 + ((size_t)K_max / 32 + 2) * 4 + 15) & ~(size_t)15));
    __shared__ int32_t fl_slot[SEGK_FLAG_LDS], fl_row[SEGK_FLAG_LDS];
    __shared__ __attribute__((aligned(16))) unsigned short fl_k[SEGK_FLAG_LDS];
    __shared__ unsigned short fl_blk[SEGK_FLAG_LDS];
    __shared__ unsigned short wsuf[8192 / 32 + 8];               // holes in the bitmap words above word w (K_max <= 8192)
    __shared__ int shK1, shK, n_holes, n_fl;
    __shared__ int32_t fl_cnt[64];
    __shared__ long long red[4];
    __shared__ int32_t ml[FIN_ROWS][FIN_ML];
    __shared__ int ml_cnt[FIN_ROWS];
    const PackAddr pa{nbl, K_max, D, cap, rank_stride, row_words};
    // flagged token q of the sweep: the first SEGK_FLAG_LDS in LDS, the others in ovf [5][ovf_cap] (slot, row, raw label, block,
    // clamped label).  Every workgroup writes the same values into them; the clamped labels have a plane of their own -- written
    // over the raw ones, a workgroup that stages late put raw labels back under a workgroup that had already replayed the clamp
    // (seen once the replay took 7 us instead of 87: wrong sums of the components founded beyond the 2 048th flagged token)
    // (the overflow arrays through an explicitly global pointer: left generic, the compiler merges the two sources of an accessor
    // into one flat pointer, and the LDS-aperture test it then needs does not always survive instruction selection)
    typedef __attribute__((address_space(1))) int32_t gi32;
    gi32 *ovg = (gi32 *)(uintptr_t)ovf;
    auto FL_SLOT = [&](int q) -> int { return q < SEGK_FLAG_LDS ? fl_slot[q] : ovg[q - SEGK_FLAG_LDS]; };
    auto FL_ROW = [&](int q) -> int { return q < SEGK_FLAG_LDS ? fl_row[q] : ovg[(int64_t)ovf_cap + q - SEGK_FLAG_LDS]; };
    auto FL_K = [&](int q) -> int { return q < SEGK_FLAG_LDS ? (int)fl_k[q] : ovg[4 * (int64_t)ovf_cap + q - SEGK_FLAG_LDS]; };
    auto FL_BLK = [&](int q) -> int { return q < SEGK_FLAG_LDS ? (int)fl_blk[q] : ovg[3 * (int64_t)ovf_cap + q - SEGK_FLAG_LDS]; };
    // element d of flagged token q's embedding row: from X, or -- a sharded corpus -- from the rows the token's rank put into
    // its record (FL_ROW is then the token's place in its block's list)
    auto FLX = [&](int q, int d) -> XT {
        if (row_words) return reinterpret_cast<const XT *>(pack + pa.rows(FL_BLK(q)))[(int64_t)FL_ROW(q) * D + d];
        return ((const XT *)c.X)[(int64_t)FL_ROW(q) * c.ldx + d];
    };
    const int64_t *packi = reinterpret_cast<const int64_t *>(pack);
    const int Kb = (int)out_scalars[3];            // K before the sweep (k_batch_sort); *m.K is rewritten by workgroup 0
    const int wg = blockIdx.x;
    const int j0 = wg * FIN_ROWS;
    SEGK_TSTAMP(3, 0);
    // issued with the first loads of the kernel, consumed after the counts below (each was a dependent round trip of its own
    // behind them, ~1.5 us per sweep each): the blocks' numbers of flagged tokens, and -- on speculation, the addresses are
    // valid whatever the numbers turn out to be -- entry (tid & 31) of block (tid >> 5): a settled chain flags a handful
    int fc_early = 0;
    if (tid < n_blocks) fc_early = reinterpret_cast<const int32_t *>(pack + pa.flg(tid))[0];
    const int sp_b = tid >> 5, sp_q = tid & 31;
    const bool sp_ok = sp_b < n_blocks && sp_q < cap;
    int sp_sl = 0, sp_kr = 0, sp_rw = 0;
    if (sp_ok) {
        const int32_t *fl = reinterpret_cast<const int32_t *>(pack + pa.flg(sp_b));
        sp_sl = fl[2 + 3 * sp_q + 0];
        sp_kr = fl[2 + 3 * sp_q + 1];
        sp_rw = fl[2 + 3 * sp_q + 2];
    }

    // ---- (0a) combined counts of the un-flagged tokens (labels < Kb): every load of a thread issued before the first use
    long long csum = 0;
    if (n_blocks == 8) {
        for (int k0 = 0; k0 < K_max; k0 += 4 * 256) {
            long long v[4][8];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + u * 256 + tid;
#pragma unroll
                for (int b = 0; b < 8; b++) v[u][b] = packi[pa.cnt(b) + (k < K_max ? k : 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + u * 256 + tid;
                if (k < K_max) {
                    long long cn = 0;
#pragma unroll
                    for (int b = 0; b < 8; b++) cn += v[u][b];
                    cnt32[k] = (int32_t)cn;
                    csum += cn;
                }
            }
        }
    } else {
        for (int k = tid; k < K_max; k += nt) {
            long long cn = 0;
            for (int b = 0; b < n_blocks; b++) cn += packi[pa.cnt(b) + k];
            cnt32[k] = (int32_t)cn;
            csum += cn;
        }
    }
    // ---- (0b) the flagged tokens of all blocks in global token order: clamp replay by one thread (the blocks'
    // counts are fetched side by side first: normally they are all zero and the replay is over at once)
    if (tid < n_blocks) fl_cnt[tid] = fc_early;
    for (int o = 32; o > 0; o >>= 1) csum += __shfl_xor(csum, o);
    if (lane == 0) red[wv] = csum;
    __syncthreads();
    SEGK_TSTAMP(3, 1);
    // the entries of all blocks are staged side by side first (raw labels), then one thread replays the clamp over the staged
    // labels -- fetched inside the replay loop, every flagged token cost that thread a dependent round trip
    {
        auto put = [&](int at, int b, int slot, int kraw, int row) {
            if (at < SEGK_FLAG_LDS) {
                fl_slot[at] = slot;
                fl_row[at] = row;
                fl_k[at] = (unsigned short)kraw;
                fl_blk[at] = (unsigned short)b;
            } else if (ovf && at - SEGK_FLAG_LDS < ovf_cap) {
                const int o = at - SEGK_FLAG_LDS;
                ovg[o] = slot;
                ovg[(int64_t)ovf_cap + o] = row;
                ovg[2 * (int64_t)ovf_cap + o] = kraw;
                ovg[3 * (int64_t)ovf_cap + o] = b;
            }
        };
        // thread t takes entry q0 + t of EVERY block, eight blocks' loads issued together (their addresses always valid: entry
        // 0 stands in where a block has no such entry): one round trip for a settled chain's handful of entries, and one per
        // 256 entries and eight blocks where a fresh chain founds components by the hundred (a loop over the blocks with
        // the loads inside cost a dependent round trip per block: 16 us of the second sweep)
        int at = 0, maxcb = 0;
        for (int b = 0; b < n_blocks; b++) {
            const int cb = fl_cnt[b] < cap ? fl_cnt[b] : cap;
            maxcb = cb > maxcb ? cb : maxcb;
        }
        for (int b0 = 0; b0 < n_blocks; b0 += 8) {
            int cbs[8], ats[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                cbs[i] = b0 + i < n_blocks ? (fl_cnt[b0 + i] < cap ? fl_cnt[b0 + i] : cap) : 0;
                ats[i] = at;
                at += cbs[i];
                // (entries 0..31 of the first nt / 32 blocks: fetched on speculation at the top of the kernel)
                if (b0 + i == sp_b && sp_ok && sp_q < cbs[i]) put(ats[i] + sp_q, sp_b, sp_sl, sp_kr, row_words ? sp_q : sp_rw);
            }
            const int q_lo = b0 < nt / 32 ? 32 : 0;          // (nt / 32 = 8 blocks: the whole group b0 = 0 or none of a later one)
            for (int q0 = q_lo; q0 < maxcb; q0 += nt) {
                const int q = q0 + tid;
                int sl[8], kr[8], rw[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int32_t *fl = reinterpret_cast<const int32_t *>(pack + pa.flg(b0 + i < n_blocks ? b0 + i : 0));
                    const int qi = q < cbs[i] ? q : 0;
                    sl[i] = fl[2 + 3 * qi + 0];
                    kr[i] = fl[2 + 3 * qi + 1];
                    rw[i] = fl[2 + 3 * qi + 2];
                }
#pragma unroll
                for (int i = 0; i < 8; i++)
                    // (row_words: the token's row travels in the record -- its place there, not its row of X, is what is kept)
                    if (q < cbs[i]) put(ats[i] + q, b0 + i, sl[i], kr[i], row_words ? q : rw[i]);
            }
        }
        if (at > SEGK_FLAG_LDS) __threadfence();                  // overflow entries: written by many threads, read by others below
    }
    __syncthreads();
    if (tid < 64) {
        int K = Kb, nf = 0, over = 0;
        for (int b = 0; b < n_blocks; b++) {
            if (fl_cnt[b] > cap) over = 1;
            nf += fl_cnt[b] < cap ? fl_cnt[b] : cap;
        }
        const int room = SEGK_FLAG_LDS + (ovf ? ovf_cap : 0);
        if (nf > room) { over = 1; nf = room; }
        // The clamp (label k: k = min(k, K); k == K founds component K, K++), 64 tokens per trip by one wave.  A token founds
        // a component iff its raw label >= K at its turn = K at the start of the trip + the founders among the lanes below
        // it: bit q of that mask depends on the bits below q only, so iterating mask -> mask'(mask) from any start has the
        // first t bits right after t rounds and its only fixed point is the sequential answer (one thread replaying label by
        // label: 0.03-0.08 us per flagged token, 36-87 us of the second sweep of a fresh chain with its 1 082 of them).
        for (int q0 = 0; q0 < nf; q0 += 64) {
            const int q = q0 + lane;
            const bool in = q < nf;
            const int kr = !in ? 0 : q < SEGK_FLAG_LDS ? (int)fl_k[q] : ovg[2 * (int64_t)ovf_cap + q - SEGK_FLAG_LDS];
            const unsigned long long lower = (1ull << lane) - 1ull;
            unsigned long long mk = __ballot(in && kr >= K);
            for (;;) {
                const unsigned long long m2 = __ballot(in && kr >= K + __popcll(mk & lower));
                if (m2 == mk) break;
                mk = m2;
            }
            const int Kq = K + __popcll(mk & lower);
            const int k = kr < Kq ? kr : Kq;
            if (in) {
                if (q < SEGK_FLAG_LDS) fl_k[q] = (unsigned short)k;
                else ovg[4 * (int64_t)ovf_cap + q - SEGK_FLAG_LDS] = k;
            }
            K += __popcll(mk);
        }
        if (lane == 0) {
            if (over && wg == 0) atomicOr(status, 4);
            shK1 = K;
            n_fl = nf;
        }
        if (nf > SEGK_FLAG_LDS) __threadfence();      // the overflow entries are read back by the other threads below
    }
    __syncthreads();
    SEGK_TSTAMP(3, 2);
    const int K1 = shK1, nfl = n_fl;
    for (int q = tid; q < nfl; q += nt) atomicAdd(&cnt32[FL_K(q)], 1);
    const long long n_tokens = red[0] + red[1] + red[2] + red[3] + nfl;
    // ---- (0c) clean_components on indices only.  The reference deletes the empty components one at a time
    // in descending order, each time moving the last active row into the hole (kmeans_components.py:129-151,
    // 263-266).  Because the holes above the current one are already gone, the row that moves is never empty,
    // every moved row originates at or above the final K and lands below it.
    // Usually no component emptied (once the chain has settled): one pass over the counts and ONE barrier say so, nothing
    // moves, and final row j holds component j -- no table is built (four barriers and 4 us less than the general path).
    __syncthreads();                                             // the flagged tokens' increments
    int hole_here = 0;
    for (int k = tid; k < K1; k += nt) hole_here |= cnt32[k] == 0;
    const int any_hole = __syncthreads_or(hole_here);
    if (any_hole) {
        const int nwords = (K1 + 31) / 32;
        for (int w = tid; w < nwords; w += nt) bitmap[w] = 0;
        for (int k = tid; k < K1; k += nt) pos2orig[k] = (unsigned short)k;
        __syncthreads();
        for (int k = tid; k < K1; k += nt)
            if (cnt32[k] == 0) atomicOr(&bitmap[k >> 5], 1u << (k & 31));
        __syncthreads();
        // The walk in descending order, every hole by a thread of its own.  Hole number i (1 = the highest) is filled from
        // position P_i = K1 - i, the last active row at that moment; that position holds its own component unless it is itself
        // a hole -- then an earlier one, number j < i (a hole is never above the last row: k_j <= P_j), and it holds what hole
        // j received: the component at P_j, and so on upwards until a position that is no hole.  The number of a hole is
        // its rank in the bitmap (holes in the words above + bits above it in its word), so nothing is sequential: one thread
        // walked the holes at 0.2 us each (30 us in the second sweep of a fresh chain, 2-14 us in the following ones).
        if (tid < 64) {
            int carry = 0;
            for (int hi = nwords - 1; hi >= 0; hi -= 64) {
                const int w = hi - lane;
                const int cw = w >= 0 ? __popc(bitmap[w]) : 0;
                int sc = cw;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(sc, o);
                    if (lane >= o) sc += t;
                }
                if (w >= 0) wsuf[w] = (unsigned short)(carry + sc - cw);
                carry += __shfl(sc, 63);
            }
            if (lane == 0) {
                shK = K1 - carry;
                n_holes = carry;
            }
        }
        __syncthreads();
        for (int k = tid; k < K1; k += nt) {
            auto is_hole = [&](int p) -> bool { return (bitmap[p >> 5] >> (p & 31)) & 1u; };
            auto number = [&](int p) -> int { return (int)wsuf[p >> 5] + __popc((bitmap[p >> 5] >> (p & 31)) >> 1) + 1; };
            if (!is_hole(k)) continue;
            const int i = number(k);
            holes[i - 1] = (unsigned short)k;
            int p = K1 - i;
            if (p == k) continue;                                  // the hole is the last row itself: nothing moves
            while (is_hole(p)) p = K1 - number(p);
            pos2orig[k] = (unsigned short)p;
        }
        __syncthreads();
    }
    SEGK_TSTAMP(3, 3);
    const int K = any_hole ? shK : K1;
    const int nholes = any_hole ? n_holes : 0;
    auto orig = [&](int j) -> int { return any_hole ? (int)pos2orig[j] : j; };        // the component final row j < K holds

    // ---- the LAST workgroup (it has no rows of its own: as part of workgroup 0 this was 4 us on the kernel's critical path)
    // publishes the scalars, the relabel table and the resolved labels of the local flagged tokens
    if (wg == (int)gridDim.x - 1) {
        for (int k = tid; k < K_max; k += nt) remap[k] = k;
        __syncthreads();
        for (int h = tid; h < nholes; h += nt) {
            const int k = holes[h];
            if (k < K) remap[pos2orig[k]] = k;
        }
        for (int q = tid; q < nfl; q += nt)
            if (FL_BLK(q) / nbl == my_rank) new_k[FL_SLOT(q)] = FL_K(q);
        if (tid == 0) {
            double tv[64];
            for (int b = 0; b < n_blocks; b++) tv[b] = pack[pa.tot(b)];
            out_scalars[0] = tree_reduce_d(tv, n_blocks);
            out_scalars[1] = (double)K;
            out_scalars[2] = (double)n_tokens;
            out_scalars[4] = 0.0;                           // accumulator of segk_kmeans_batch_record
            *m.K = K;
            if (sp_zero_slot) *sp_zero_slot = 0u;          // E_m of the fp16 tile image: k_batch_post's atomic maximum
        }
        return;
    }

    // ---- (1) this workgroup's final rows, one element (row, dimension) per thread and step; the loads of four steps
    // (32 with the default eight blocks) are issued together
    // match lists of this workgroup's rows that hold a component founded in this sweep (its tokens are flagged ones only)
    // (a wave per two rows, 64 flagged tokens per trip, the matches kept in token order by ballot and prefix count: one
    // thread per row walked all flagged tokens -- up to 81 us of the second sweep of a fresh chain)
    for (int r = wv * (FIN_ROWS / 4); r < (wv + 1) * (FIN_ROWS / 4); r++) {
        const int j = j0 + r;
        int n = 0;
        if (j < K && j < K_max && orig(j) >= Kb) {                // (wave-uniform)
            const int sc = orig(j);
            for (int q0 = 0; q0 < nfl; q0 += 64) {
                const int q = q0 + lane;
                const bool hit = q < nfl && FL_K(q) == sc;
                const unsigned long long mk = __ballot(hit);
                const int at = n + __popcll(mk & ((1ull << lane) - 1ull));
                if (hit && at < FIN_ML) ml[r][at] = q;
                n += __popcll(mk);
            }
        }
        if (lane == 0) ml_cnt[r] = n;
    }
    __syncthreads();
    SEGK_TSTAMP(3, 4);
    XT *__restrict__ means = (XT *)m.means;
    double *__restrict__ numer = m.mean_numerators;
    const XT *__restrict__ rnd = (const XT *)m.random_means;
    (void)0;   // (the rows of X are read through FLX: from X, or from the record when the corpus is sharded)
    const double *__restrict__ rpack = pack;
    const int nel = FIN_ROWS * D;
    for (int e0 = 0; e0 < nel; e0 += 4 * 256) {
        double tv[4][8];
        int cls[4], src[4];          // 0 nothing, 1 inactive row, 2 eight-block tree from the records, 3 general
        int el[4];
        XT xh[4][FIN_MH];            // founded components with at most FIN_MH tokens: the tokens' elements, requested early
        int bh[4][FIN_MH], nh[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = e0 + u * 256 + tid;
            const int r = e / D, d = e - r * D, j = j0 + r;
            cls[u] = 0;
            src[u] = 0;
            el[u] = e;
            if (e < nel && j < K_max) {
                if (j >= K) cls[u] = 1;
                else {
                    src[u] = orig(j);
                    cls[u] = (n_blocks == 8 && src[u] < Kb) ? 2 : 3;
                }
            }
            const int64_t off = (int64_t)(cls[u] == 2 ? src[u] : 0) * D + (cls[u] == 2 ? d : 0);
            if (n_blocks == 8) {
#pragma unroll
                for (int b = 0; b < 8; b++) tv[u][b] = rpack[pa.sum(b) + off];         // unconditional: always a valid address
            }
            // a component founded this sweep with a short match list (the usual case: a settled chain founds one or two per
            // sweep, of one or two tokens): its tokens' rows are requested HERE, with the records' sums of the other elements
            // -- requested inside the loop below they were one more dependent round trip for the workgroup that holds the
            // component, 4-6 us at the end of the kernel every sweep
            nh[u] = -1;
            if (cls[u] == 3 && n_blocks == 8 && src[u] >= Kb && ml_cnt[r] >= 1 && ml_cnt[r] <= FIN_MH) {
                nh[u] = ml_cnt[r];
#pragma unroll
                for (int i = 0; i < FIN_MH; i++) {
                    const int qq = ml[r][i < nh[u] ? i : nh[u] - 1];
                    bh[u][i] = FL_BLK(qq);
                    xh[u][i] = FLX(qq, d);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (cls[u] == 0) continue;
            const int r = el[u] / D, d = el[u] - r * D;
            const int64_t at = (int64_t)(j0 + r) * D + d;
            if (cls[u] == 1) {                                 // inactive row (kmeans_components.py:163-166)
                const XT rv = rnd[at];
                numer[at] = 0.0;
                means[at] = rv;
                mrow[el[u]] = (double)rv;
                continue;
            }
            double v;
            if (cls[u] == 2) {
                v = ((tv[u][0] + tv[u][1]) + (tv[u][2] + tv[u][3])) + ((tv[u][4] + tv[u][5]) + (tv[u][6] + tv[u][7]));
            } else if (nh[u] >= 0) {
                double g8[8];
#pragma unroll
                for (int b = 0; b < 8; b++) g8[b] = 0.0;
#pragma unroll
                for (int i = 0; i < FIN_MH; i++)
                    if (i < nh[u]) {
#pragma unroll
                        for (int b = 0; b < 8; b++) g8[b] = bh[u][i] == b ? g8[b] + (double)xh[u][i] : g8[b];
                    }
                v = ((g8[0] + g8[1]) + (g8[2] + g8[3])) + ((g8[4] + g8[5]) + (g8[6] + g8[7]));
            } else if (n_blocks == 8 && src[u] >= Kb && ml_cnt[r] <= FIN_ML) {
                // a component founded this sweep: its flagged tokens, block by block -- the row's match list (LDS), eight
                // rows in flight, the block's accumulator chosen by predicate (a dynamically indexed array lives in scratch)
                double g8[8];
#pragma unroll
                for (int b = 0; b < 8; b++) g8[b] = 0.0;
                const int nmr = ml_cnt[r];
                for (int i0 = 0; i0 < nmr; i0 += 8) {
                    XT xq[8];
                    int bq[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int qq = ml[r][i0 + i < nmr ? i0 + i : nmr - 1];
                        bq[i] = FL_BLK(qq);
                        xq[i] = FLX(qq, d);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (i0 + i < nmr) {
#pragma unroll
                            for (int b = 0; b < 8; b++) g8[b] = bq[i] == b ? g8[b] + (double)xq[i] : g8[b];
                        }
                }
                // (registers and the fixed tree: the general path's array lives in scratch, and its tree walks it through memory --
                // seven dependent round trips, 6 us for the workgroup that holds a new component)
                v = ((g8[0] + g8[1]) + (g8[2] + g8[3])) + ((g8[4] + g8[5]) + (g8[6] + g8[7]));
            } else {                                           // (general: any number of blocks, long lists)
                double gv[64];
                if (src[u] < Kb) {
                    for (int b = 0; b < n_blocks; b++) gv[b] = rpack[pa.sum(b) + (int64_t)src[u] * D + d];
                } else {
                    for (int b = 0; b < n_blocks; b++) gv[b] = 0.0;
                    for (int q = 0; q < nfl; q++)
                        if (FL_K(q) == src[u]) gv[FL_BLK(q)] += (double)FLX(q, d);
                }
                v = tree_reduce_d(gv, n_blocks);
            }
            const XT mv = (XT)(v / (double)cnt32[src[u]]);
            numer[at] = v;
            means[at] = mv;
            mrow[el[u]] = (double)mv;
        }
    }
    if (tid < FIN_ROWS) {
        const int j = j0 + tid;
        if (j < K_max) m.counts[j] = j < K ? (int64_t)cnt32[orig(j)] : 0;
    }
    __syncthreads();
    SEGK_TSTAMP(3, 5);
    // ---- (2) this workgroup's part of the fp32 MFMA image (layout: segk_internal.h; the padding of the image -- dimensions
    // beyond D, components beyond K_max -- never changes after segk_kmeans_prepare), |m|^2 maximum, row hashes: the
    // arithmetic of dev_prepare_tile, 8 lanes per component
    // (2') sp_spec: ALSO this workgroup's part of the fp16x2 image, with the exponent the image has NOW (header word 0).
    // The exponent follows max |m|^2 over all rows, which is complete only when this kernel ends -- the reason the image is
    // the post kernel's job -- but from one sweep to the next it almost never moves: the post kernel compares, and rebuilds
    // the image only when it did (same arithmetic as dev_prepare_sp_tile: the two ways give the same bits).  The residual
    // maximum E_m of these rows goes to header word 2; the post kernel moves it to word 1.
    const int G = segk_gmax(D);
    float *T = m.tiles + (int64_t)(j0 >> 5) * segk_tile_stride(D);
    const int KSsp = segk_b3_kp(D) / 16;
    const int eb_prev = sp_spec ? ((const int *)m.tiles_b3)[0] : 0;
    const int ea_sp = sp_spec ? ((const int *)c.Xb3)[1] : 0;
    float *Tsp = sp_spec ? m.tiles_b3 + 1024 + (int64_t)(j0 >> 5) * segk_sp_tile_stride(D, 2) : nullptr;
    // header word 3: the exponent THIS kernel built with -- what the post kernel compares the new exponent with.  (It
    // compared with word 0 until round 4, which its own tile-0 workgroup rewrites when it rebuilds: a tile workgroup that
    // read the word after that took its tile for built and left it in the old exponent under the new header -- scores of
    // that tile's components off by a power of two, intermittently, in the sweep after the exponent moved.)
    if (sp_spec && wg == 0 && tid == 0) ((int *)m.tiles_b3)[3] = eb_prev;
    {
        // three independent reductions over a row's elements, one per wave (8 lanes per component, the summation pattern of
        // dev_prepare_tile / dev_prepare_sp_tile): wave 0 |m|^2, wave 1 the fp16 residual, wave 2 the value hash
        const int what = tid >> 6, t64 = tid & 63;
        const int r = t64 >> 3, sub = t64 & 7, comp = j0 + r;
        const bool live = comp < K_max;
        if (what == 0) {
            double s = 0.0;
            if (live)
                for (int d = sub; d < D; d += 8) {
                    const double v = mrow[r * D + d];
                    s += v * v;
                }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (sub == 0 && live) {
                T[G * 128 + (comp & 31)] = (float)(-0.5 * s);
                atomicMax((unsigned long long *)m.mnorm_max, (unsigned long long)__double_as_longlong(s));
                if (sp_spec) Tsp[KSsp * 2 * 256 + (comp & 31)] = (float)ldexp(-0.5 * s, ea_sp + eb_prev);      // accumulator seed, scaled domain
            }
        } else if (what == 1) {
            if (sp_spec) {
                double rs = 0.0;
                if (live)
                    for (int d = sub; d < D; d += 8) rs += sp_resid2(ldexpf((float)mrow[r * D + d], eb_prev));
                rs += __shfl_xor(rs, 1);
                rs += __shfl_xor(rs, 2);
                rs += __shfl_xor(rs, 4);
                if (sub == 0 && live) {
                    const float em = (float)(ldexp(sqrt(rs), -eb_prev) * (1.0 + 1e-6));
                    atomicMax((unsigned int *)m.tiles_b3 + 2, __float_as_uint(em));
                }
            }
        } else if (what == 2) {
            if (row_hash) {
                unsigned long long hh = 0ull;
                if (live)
                    for (int d = sub; d < D; d += 8) hh += segk_elem_hash(mrow[r * D + d], d);
                hh += __shfl_xor(hh, 1);
                hh += __shfl_xor(hh, 2);
                hh += __shfl_xor(hh, 4);
                if (sub == 0 && live) row_hash[comp] = hh | 1ull;        // never 0: the empty key of the hash table
            }
        }
    }
    if (sp_spec) {
        typedef SegkPiece<2>::T T16;
        typedef SegkPiece<2>::V8 V8;
        T16 *Tb = (T16 *)Tsp;
        for (int it = tid; it < FIN_ROWS * KSsp * 2; it += nt) {
            const int r = it / (KSsp * 2), rem = it - r * (KSsp * 2), sidx = rem >> 1, h = rem & 1;
            const int comp = j0 + r;
            if (comp >= K_max) continue;
            const int ln = (comp & 31) + 32 * h;
            V8 out[2];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int d = segk_b3_dim(16 * sidx + 8 * h + i);
                const float v = d < D ? ldexpf((float)mrow[r * D + d], eb_prev) : 0.f;
                T16 pc[2];
                split_sp<2>(v, pc);
                out[0][i] = pc[0];
                out[1][i] = pc[1];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) *reinterpret_cast<V8 *>(Tb + ((sidx * 2 + q) * 64 + ln) * 8) = out[q];
        }
    }
    for (int e = tid; e < nel; e += nt) {
        const int r = e / D, d = e - r * D, comp = j0 + r;
        if (comp < K_max) T[(d >> 2) * 128 + ((((d >> 1) & 1) * 32 + (comp & 31)) << 1) + (d & 1)] = (float)mrow[e];
    }
    SEGK_TSTAMP(3, 6);
}

// final labels of the local tokens + (tile workgroups) split-precision image and duplicate marking
template <typename XT, int P>
__global__ __launch_bounds__(256) void k_batch_post(segk_corpus c, segk_kmeans m, int lo, int hi, int32_t *new_k, const int32_t *remap,
                                                    int n_tiles, int stride32, int G, float *tiles_sp, int stride_sp, int sp_const_off,
                                                    const unsigned long long *row_hash, int sp_spec)
{
    if ((int)blockIdx.x >= n_tiles) {
        const int64_t idx = (int64_t)(blockIdx.x - n_tiles) * blockDim.x + threadIdx.x;
        const int64_t p0 = (int64_t)lo * c.N_max, tot = (int64_t)(hi - lo) * c.N_max;
        if (idx < tot) {
            const int k = new_k[p0 + idx];
            if (k >= 0) new_k[p0 + idx] = remap[k];
        }
        SEGK_TSTAMP_MAX(4, 4);
        return;
    }
    const int tile = blockIdx.x;
    SEGK_TSTAMP(4, 0);
    if constexpr (P != 0) {
        if (tiles_sp) {
            // P = 2: the finalize kernel has written the image with the exponent of the previous one (its sp_spec path); when
            // the exponent of the new means is the same there is nothing to build (workgroup-uniform)
            bool built = false;
            if (P == 2 && sp_spec) {
                const int eb_new = sp_exponent((float)(sqrt(*m.mnorm_max) * (1.0 + 1e-6)));
                built = eb_new == ((const int *)tiles_sp)[3];           // (word 3: no workgroup of this kernel writes it)
            }
            if (!built)
                dev_prepare_sp_tile<P>((const float *)m.means, m.K_max, c.D, tiles_sp, m.mnorm_max, (const unsigned char *)c.Xb3,
                                       (const double *)nullptr, tile);
            if (P == 2 && sp_spec && tile == 0 && threadIdx.x == 0) {
                if (built) ((unsigned int *)tiles_sp)[1] = ((const unsigned int *)tiles_sp)[2];      // E_m of the rows
                ((unsigned int *)tiles_sp)[2] = 0u;
            }
        }
    }
    SEGK_TSTAMP(4, 1);
    if (!row_hash) return;
    // clean_components leaves exact copies behind (the moved rows, the inactive rows): a duplicate with the higher
    // index can never be np.argmax, but every embedding near the pair is a tie for the full scan -- its accumulator
    // seed becomes the "absent" constant in both images (k_kmeans_mark_dups, one tile per workgroup here)
    __shared__ unsigned long long keys[SEGK_DUP_TB];
    __shared__ int32_t first[SEGK_DUP_TB];
    dev_dup_table(keys, first, row_hash, m.K_max);            // ends with a barrier: the constants above are written
    SEGK_TSTAMP(4, 2);
    const XT *means = (const XT *)m.means;
    const int tid = threadIdx.x, sub = tid & 7, D = c.D;
    const int k = tile * 32 + (tid >> 3);
    int i = -1;
    if (k < m.K_max) {
        i = dev_dup_first(keys, first, row_hash[k]);
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
        m.tiles[(int64_t)tile * stride32 + G * 128 + (k & 31)] = -3.0e38f;
        if (tiles_sp) tiles_sp[1024 + (int64_t)tile * stride_sp + sp_const_off + (k & 31)] = -3.0e38f;
    }
    SEGK_TSTAMP_MAX(4, 3);
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

// the same metric straight from the token lists of a batch sweep (no `assignments` needed), plus the status word: one
// small device-to-host copy then carries every record value of the sweep.  out [>= 7]: out[4] += metric (zeroed by
// k_batch_finalize), out[5] = status[0], out[6] = status[1]
template <typename XT>
__global__ void k_kmeans_record_tokens(segk_corpus c, segk_kmeans m, int lo, int hi, const int32_t *new_tok, const int32_t *new_k,
                                       const int32_t *status, double *out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t p0 = (int64_t)lo * c.N_max, tot = (int64_t)(hi - lo) * c.N_max;
    if (blockIdx.x == 0 && threadIdx.x == 0 && status) { out[5] = (double)status[0]; out[6] = (double)status[1]; }
    double s = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * nw + wv; p < tot; p += (int64_t)gridDim.x * nw) {
        const int k = new_k[p0 + p];
        if (k < 0) continue;
        const int64_t e = new_tok[p0 + p];
        const double cnt = (double)m.counts[k];
        for (int d = lane; d < c.D; d += 64) {
            const double delta = m.mean_numerators[(int64_t)k * c.D + d] / cnt - (double)((const XT *)c.X)[e * c.ldx + d];
            s += delta * delta;
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double part[16];
    if (lane == 0) part[wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < nw; w++) t += part[w];
        if (t != 0.0) atomicAdd(out + 4, -t);
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


// the stable counting sort of (1a), also used by the record metrics of the FBGMM drivers (segk_metrics.hip)
int segk_launch_batch_sort(const segk_corpus *c, const segk_kmeans *m, const int32_t *blk_lo, int n_blocks, const int32_t *new_tok,
                           const int32_t *new_k, const int32_t *n_flag, const double *out_total, int32_t *sorted, int32_t *koff,
                           double *part_tot, int32_t *flags, int cap, double *out_scalars, hipStream_t st)
{
    const int nw = m->K_max <= 1024 ? 16 : m->K_max <= 2048 ? 8 : m->K_max <= 4096 ? 4 : 2;
    size_t lds = (size_t)(m->K_max + 2) * 4 + (size_t)nw * m->K_max * 4 + (size_t)SORT_KEYS_LDS * 2;
    if (lds < 2048 * sizeof(double)) lds = 2048 * sizeof(double);
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_batch_sort, lds));
    segk_tstamp_bind();
    // out_total == NULL: the sort alone (no flag lists, no totals: the second half of the grid is not launched)
    hipLaunchKernelGGL(k_batch_sort, dim3((unsigned)((out_total ? 2 : 1) * n_blocks)), dim3(SORT_THREADS), lds, st, *c, *m, blk_lo,
                       n_blocks, new_tok, new_k, n_flag, out_total, sorted, koff, part_tot, flags, cap, out_scalars);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
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

}  // extern "C"

// the per-utterance update alone (no image refresh): segk_kmeans_sequential_sweep refreshes the images once per sweep
int segk_launch_update_utt(const segk_corpus *c, segk_kmeans *m, int utt, const int32_t *old_tok, const int32_t *new_tok,
                           const int32_t *new_k, const int32_t *n_old, const int32_t *n_new, int32_t *status, hipStream_t st)
{
    // the LDS-staged form for utterances of at most 64 landmarks (SEGK_SEQ_UPDATE=0: the item-by-item kernel)
    const char *e = getenv("SEGK_SEQ_UPDATE");
    if (c->N_max <= 32 && !(e && atoi(e) == 0)) {
        const size_t esz = c->x_dtype == SEGK_F32 ? 4 : 8;
        const size_t lds = (size_t)2 * c->N_max * c->D * (8 + 2 * esz);
        if (lds <= 150 * 1024) {
            DISPATCH_XT(c, {
                SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_seq_update<XT>, lds));
                hipLaunchKernelGGL(k_seq_update<XT>, dim3(1), dim3(256), lds, st, *c, *m, utt, old_tok, new_tok, new_k, n_old, n_new, status);
            });
            SEGK_LAUNCH_CHECK();
            return SEGK_OK;
        }
    }
    return launch_update(c, m, 0, utt, 0, 0, old_tok, new_tok, new_k, n_old, n_new, status, st);
}

extern "C" {

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

int32_t segk_kmeans_batch_scratch_words(int32_t K_max, int64_t n_slots, int32_t n_blocks_local, int64_t *sorted_words,
                                        int64_t *koff_words)
{
    SEGK_REQUIRE(K_max >= 1 && n_slots >= 0 && n_blocks_local >= 0 && sorted_words && koff_words, "batch_scratch_words operands");
    *sorted_words = (n_slots > 0 ? n_slots : 1) * segk_sort_ranges(K_max);
    *koff_words = (int64_t)(n_blocks_local > 0 ? n_blocks_local : 1) * K_max * 2;
    return SEGK_OK;
}

int64_t segk_kmeans_batch_record_words(int32_t K_max, int32_t D, int32_t n_blocks_local, int32_t flag_cap, int32_t flag_row_bytes)
{
    const int64_t rw = flag_row_bytes > 0 ? ((int64_t)flag_cap * flag_row_bytes + 7) / 8 : 0;
    return (int64_t)n_blocks_local * ((int64_t)K_max * D + 1 + K_max + segk_flag_words(flag_cap) + rw);
}

int32_t segk_kmeans_batch_partials(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                   const int32_t *blk_lo, int32_t n_blocks_local,
                                   const int32_t *new_tok, const int32_t *new_k, const int32_t *n_flag,
                                   const double *out_total, int32_t *sorted_scratch, int32_t *koff_scratch,
                                   double *record, int32_t flag_cap, int32_t flag_rows, double *out_scalars, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    if (n_blocks_local <= 0) return SEGK_OK;
    SEGK_REQUIRE(blk_lo && new_tok && new_k && n_flag && out_total && record && out_scalars && sorted_scratch && koff_scratch,
                 "batch_partials operands");
    SEGK_REQUIRE(flag_cap >= 1, "flag_cap");
    SEGK_REQUIRE(m->K_max <= 8192, "batch mode supports K_max <= 8192");
    const int64_t nbl = n_blocks_local, KD = (int64_t)m->K_max * c->D;
    double *part_sum = record;
    double *part_tot = record + nbl * KD;
    int64_t *part_cnt = reinterpret_cast<int64_t *>(record + nbl * KD + nbl);
    int32_t *flags = reinterpret_cast<int32_t *>(record + nbl * KD + nbl + nbl * m->K_max);
    hipStream_t st = (hipStream_t)stream;
    (void)n_flag;
    segk_tstamp_bind();
    const int NR = segk_sort_ranges(m->K_max), rsh = segk_sort_rsh(m->K_max), nsh = segk_sort_nsh(m->K_max);
    const bool fuse = c->x_dtype == SEGK_F32 && c->D <= 128 && (c->D & 1) == 0 && (c->ldx & 1) == 0;
    {
        size_t lds = (size_t)(16 + 2) * (1 << rsh) * 4 + (size_t)16 * SORT_CL * 4;
        if (lds < 2048 * sizeof(double)) lds = 2048 * sizeof(double);
        hipLaunchKernelGGL(k_batch_sort_sum, dim3((unsigned)(nbl * (NR + 2))), dim3(SORT_THREADS), lds, st, *c, *m, blk_lo,
                           n_blocks_local, new_tok, new_k, out_total, sorted_scratch, koff_scratch, part_tot, flags, flag_cap,
                           out_scalars, NR, rsh, nsh, part_sum, part_cnt, fuse ? 1 : 0);
        SEGK_LAUNCH_CHECK();
    }
    if (flag_rows) {
        // a shard of the corpus: the embedding rows of the flagged tokens into the record, behind the flag lists
        const int elem = c->x_dtype == SEGK_F32 ? 4 : 8;
        const int64_t rw = segk_flag_row_words(flag_cap, c->D, elem);
        double *rows = record + nbl * KD + nbl + nbl * m->K_max + nbl * segk_flag_words(flag_cap);
        DISPATCH_XT(c, hipLaunchKernelGGL(k_batch_flag_rows<XT>, dim3((unsigned)nbl), dim3(256), 0, st, *c, flags, flag_cap, (XT *)rows, rw););
        SEGK_LAUNCH_CHECK();
    }
    if (fuse) return SEGK_OK;
    const int64_t grid = nbl * ((m->K_max + PART_WAVES - 1) / PART_WAVES);
    DISPATCH_XT(c, hipLaunchKernelGGL(k_batch_partials<XT>, dim3((unsigned)grid), dim3(64 * PART_WAVES), 0, st, *c, *m, blk_lo,
                                       n_blocks_local, sorted_scratch, koff_scratch, part_sum, part_cnt, NR););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_batch_finalize(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t utt_lo,
                                   int32_t utt_hi, const double *records, int32_t n_blocks_total,
                                   int32_t n_blocks_per_rank, int64_t rank_stride, int32_t flag_cap, int32_t flag_rows,
                                   int32_t my_rank, int32_t *new_k, int32_t *remap_scratch, double *out_scalars, int32_t *status,
                                   void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(n_blocks_total >= 1 && n_blocks_total <= 64, "1 <= n_blocks_total <= 64");
    SEGK_REQUIRE(n_blocks_per_rank >= 1 && n_blocks_total % n_blocks_per_rank == 0, "blocks per rank");
    SEGK_REQUIRE(m->K_max <= 8192, "batch mode supports K_max <= 8192");
    SEGK_REQUIRE(records && new_k && remap_scratch && out_scalars && status && m->tiles && m->mnorm_max, "batch_finalize operands");
    SEGK_REQUIRE(0 <= utt_lo && utt_lo <= utt_hi && utt_hi <= c->n_utt, "utterance range");
    hipStream_t st = (hipStream_t)stream;
    // value hashes of the rows, for the duplicate marking (context-owned, K_max <= 2048 only; SEGK_MARK_DUPS=0: leave
    // the duplicates in the filters' images)
    const char *md = getenv("SEGK_MARK_DUPS");
    unsigned long long *row_hash = nullptr;
    if (ctx && m->K_max <= 2048) {
        if (!ctx->row_hash) SEGK_CHECK_HIP(hipMalloc((void **)&ctx->row_hash, 2048 * sizeof(unsigned long long)));
        ctx->row_hash_means = m->means;
        if (!(md && atoi(md) == 0)) row_hash = ctx->row_hash;
    }
    // overflow arrays of the clamp replay: the flagged tokens of a sweep beyond the SEGK_FLAG_LDS the kernel keeps in LDS
    // (a first sweep with K << K_max: the inactive rows are data points, every token near one founds a component)
    int32_t *ovf = nullptr;
    int ovf_cap = 0;
    const int64_t flag_max = (int64_t)n_blocks_total * flag_cap;
    if (ctx && flag_max > SEGK_FLAG_LDS) {
        const int64_t need = flag_max - SEGK_FLAG_LDS;
        SEGK_REQUIRE(need < (1ll << 28), "n_blocks_total * flag_cap");
        if (ctx->flag_ovf_cap < need) {
            SEGK_REQUIRE(!ctx->capturing, "the overflow arrays of the clamp replay must exist before a graph capture (run one sweep first)");
            SEGK_CHECK_HIP(hipStreamSynchronize(st));
            if (ctx->flag_ovf) (void)hipFree(ctx->flag_ovf);
            ctx->flag_ovf = nullptr;
            ctx->flag_ovf_cap = 0;
            SEGK_CHECK_HIP(hipMalloc((void **)&ctx->flag_ovf, (size_t)need * 5 * sizeof(int32_t)));
            ctx->flag_ovf_cap = need;
        }
        ovf = ctx->flag_ovf;
        ovf_cap = (int)ctx->flag_ovf_cap;
    }
    const bool sp = m->tiles_b3 && c->Xb3 && c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128 && (c->sp_pieces == 2 || c->sp_pieces == 3);
    const int sp_spec = sp && c->sp_pieces == 2 ? 1 : 0;      // fp16x2 image by the finalize kernel, checked by the post kernel
    const int n_tiles = segk_n_tiles(m->K_max);
    const size_t lds = (((size_t)m->K_max * 8 + ((size_t)m->K_max / 32 + 2) * 4 + 15) & ~(size_t)15) + (size_t)FIN_ROWS * c->D * sizeof(double);
    DISPATCH_XT(c, {
        if (lds > 32 * 1024)
            SEGK_CHECK_HIP(hipFuncSetAttribute((const void *)k_batch_finalize<XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_batch_finalize<XT>, dim3((m->K_max + FIN_ROWS - 1) / FIN_ROWS + 1), dim3(256), lds, st, *c, *m, records,
                           n_blocks_total, n_blocks_per_rank, rank_stride, flag_cap, my_rank, new_k, remap_scratch, out_scalars,
                           status, ctx && m->K_max <= 2048 ? ctx->row_hash : (unsigned long long *)nullptr,
                           sp ? (unsigned int *)m->tiles_b3 + 1 : (unsigned int *)nullptr, ovf, ovf_cap, sp_spec,
                           flag_rows ? segk_flag_row_words(flag_cap, c->D, c->x_dtype == SEGK_F32 ? 4 : 8) : (int64_t)0);
    });
    const int64_t nslot = (int64_t)(utt_hi - utt_lo) * c->N_max;
    const unsigned grid = (unsigned)(n_tiles + (nslot + 255) / 256);
    const int kp = segk_b3_kp(c->D);
    const int stride32 = segk_tile_stride(c->D), G = segk_gmax(c->D);
    const int stride_sp = sp ? segk_sp_tile_stride(c->D, c->sp_pieces) : 0, sp_off = sp ? (kp / 16) * c->sp_pieces * 256 : 0;
    if (sp && c->sp_pieces == 2)
        hipLaunchKernelGGL((k_batch_post<float, 2>), dim3(grid), dim3(256), 0, st, *c, *m, utt_lo, utt_hi, new_k, remap_scratch,
                           n_tiles, stride32, G, m->tiles_b3, stride_sp, sp_off, row_hash, sp_spec);
    else if (sp)
        hipLaunchKernelGGL((k_batch_post<float, 3>), dim3(grid), dim3(256), 0, st, *c, *m, utt_lo, utt_hi, new_k, remap_scratch,
                           n_tiles, stride32, G, m->tiles_b3, stride_sp, sp_off, row_hash, 0);
    else
        DISPATCH_XT(c, hipLaunchKernelGGL((k_batch_post<XT, 0>), dim3(grid), dim3(256), 0, st, *c, *m, utt_lo, utt_hi, new_k,
                                           remap_scratch, n_tiles, stride32, G, (float *)nullptr, 0, 0, row_hash, 0););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int32_t segk_kmeans_batch_record(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int32_t utt_lo, int32_t utt_hi,
                                 const int32_t *new_tok, const int32_t *new_k, const int32_t *status, double *out_scalars,
                                 void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(new_tok && new_k && out_scalars, "batch_record operands");
    SEGK_REQUIRE(0 <= utt_lo && utt_lo <= utt_hi && utt_hi <= c->n_utt, "utterance range");
    const int64_t tot = (int64_t)(utt_hi - utt_lo) * c->N_max;
    int64_t grid = (tot + 31) / 32;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_record_tokens<XT>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *c, *m,
                                       utt_lo, utt_hi, new_tok, new_k, status, out_scalars););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
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

// clean_components (kmeans_components.py:263-266) alone, no image refresh: the persistent sequential chain calls it between launches
int segk_launch_clean(const segk_corpus *c, segk_kmeans *m, int32_t *status, hipStream_t st, int32_t *relog)
{
    if (relog) {
        DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_clean_log<XT>, dim3(1), dim3(256), 0, st, *c, *m, relog););
        hipLaunchKernelGGL(k_kmeans_relabel_log, dim3((unsigned)((c->n_emb + 255) / 256)), dim3(256), 0, st, m->assignments, c->n_emb,
                           (const int32_t *)relog);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    return launch_update(c, m, 3, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, status, st);
}
