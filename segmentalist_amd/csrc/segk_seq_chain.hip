// segk_seq_chain.hip -- the reference's sequential k-means chain (segment_i for one utterance after the other,
// kmeans_acoustic_wordseg.py:225-332, 393-399) as ONE persistent kernel
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h, segk_segment_dev.h)
//
// The three-launch form (k_seq_score -> k_kmeans_segment_w8 -> k_seq_update per utterance) costs 44-52 us per utterance:
// three kernel boundaries and a dozen dependent memory round trips, every one of them paid 10 000 times per sweep because
// utterance i + 1 needs the means utterance i leaves behind.  Here G workgroups stay resident for the whole sweep and meet
// at ONE grid barrier per utterance (12.3 us per utterance on the headline corpus; 19 at the end of round 2):
//
//   * owner computes: workgroup g owns the components [g * cpw, (g + 1) * cpw) -- their means, numerators and counts live
//     in its LDS for the whole sweep (written through to memory, never read back);
//   * phase A: every workgroup scores the utterance's candidate rows (staged by band entry; the NEXT utterance's are
//     fetched by the otherwise idle waves during phase B) against ITS components in the reference's arithmetic
//     (neg_sqd_exact; two components per thread, so that a row's elements leave LDS once for both: the phase is bound by
//     LDS instructions), one 64-bit atomic maximum of (score, ~component) per span, every span's word on its own cache line;
//   * grid barrier: arrivals on one counter, the last arriver releases the others through separate flag words; spins are
//     bounded and end in an error, never in a hang;
//   * phase B, replicated: EVERY workgroup reads the spans' maxima and runs the same DP (seg_w8_uniform) on the same inputs,
//     so every workgroup knows the utterance's old and new tokens without another exchange; each applies the del_item /
//     add_item sequence (kmeans_components.py:93-132, the `k > K -> K` clamp included) to the components it owns, in the
//     reference's order, from the rows it staged.  Labels and boundaries are written by every workgroup (the same values:
//     whichever L2 a later read hits holds them).  The items are kept by span end, one lane per span end holding the old
//     and the new token that end there; a workgroup walks only the items on ITS components (two 64-bit masks), so the
//     hundred-odd workgroups an utterance does not touch do nothing after the DP.
//
// Nothing but atomics crosses workgroups inside the kernel (maxima, barrier counter and release words), so no cache
// maintenance is needed between the XCDs' L2s.  clean_components (:263-266) moves rows between owners and relabels the
// whole assignment vector: when a component empties, its owner sets the stop bit of the barrier word, every workgroup
// leaves after the utterance, the host runs the ordinary clean kernel and relaunches from the next utterance (a handful of
// times per sweep once the model has settled; 124 times in the first sweep of the bench corpus).
#include "segk_kmeans_dev.h"
#include "segk_segment_dev.h"
#include "segk_chain_barrier.h"

#define CH_THREADS 1024
#define CH_MAXOPS 64              /* old tokens [0, 32) + new tokens [32, 64) of an utterance, by span end (N_max <= 32) */
#define CH_KEY_PITCH 16           /* 8-byte words between two spans' maxima: one 128-byte line each */

struct ChainArgs {
    segk_corpus c;
    segk_kmeans m;
    const int32_t *order;         // [dev] utterances of the sweep
    int q0, q1;                   // this launch walks order[q0 .. q1)
    int n_max;                    // n_slices_max (1..8)
    double wip;
    unsigned long long *keys;     // [3][nb_cap][16] span maxima (score, ~component), one 128-byte line per span, zero between uses
    int32_t *ctl;                 // [0] barrier counter, [1] stop flag, [2] utterances completed, [3] error
    uint8_t *boundaries;
    int32_t *old_tok, *new_tok, *new_k, *n_old, *n_new, *n_flag;
    double *out_total;
    int32_t *status;
    int cpw;                      // components per workgroup
    int nb_cap;                   // N_max * W: band entries per utterance
    int ldm;                      // floats per staged component row
    unsigned long long *stamp;    // development (SEGK_CHAIN_STAMP=1): wall_clock64 at the phase boundaries of workgroup 0, 8 per utterance
};
#define CH_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (A.stamp && blockIdx.x == 0 && tid == 0 && q - A.q0 < 256) A.stamp[(q - A.q0) * 8 + (slot)] = wall_clock64(); \
    } while (0)

// what phase A needs of one utterance, staged ahead (two sets: the next utterance's is filled while wave 0 runs the DP)
struct ChainSet {
    float *xs, *xo;                                   // [NBC][LDX] candidate rows by band entry, [NM][LDX] rows of the old tokens
    double *bdur;                                     // [NBC]
    int32_t *bid, *basg, *o_e, *o_a;                  // [NBC] row of the entry, its label before the utterance (workgroup 0); [NM] old tokens, their labels
    int32_t *meta;                                    // [12]: u, N, W, nb, old mask (2 words), number of old tokens, band flag, mask of the old tokens with a row (2 words)
};

__global__ __launch_bounds__(CH_THREADS) void k_seq_chain(ChainArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ch_lds[];
    const segk_corpus &c = A.c;
    const segk_kmeans &m = A.m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int D = c.D, D4 = D >> 2, LDM = A.ldm, LDX = D, NBC = A.nb_cap, CPW = A.cpw, NM = c.N_max;
    const int k0 = blockIdx.x * CPW;
    const int kn = k0 + CPW <= m.K_max ? CPW : (m.K_max > k0 ? m.K_max - k0 : 0);      // components this workgroup owns
    const float *X = (const float *)c.X;
    float *means_g = (float *)m.means;
    const int64_t triMax = (int64_t)NM * (NM + 1) / 2;

    // ---- LDS (the size is worked out in the same order by segk_launch_seq_chain)
    unsigned char *lp = ch_lds;
    auto take = [&](size_t bytes) { unsigned char *r = lp; lp += (bytes + 15) & ~(size_t)15; return r; };
    double *numer_l = reinterpret_cast<double *>(take((size_t)CPW * D * 8));            // [CPW][D]
    double *bvec = reinterpret_cast<double *>(take((size_t)NM * 8 * 8));                // [NM][8] the DP's candidates, pitch 8 (seg_w8_uniform)
    double *gam = reinterpret_cast<double *>(take((size_t)(NM + 12) * 8));              // [8 + NM + 4]
    long long *cnt_l = reinterpret_cast<long long *>(take((size_t)CPW * 8));            // [CPW]
    long long *op_cnt = reinterpret_cast<long long *>(take((size_t)CH_MAXOPS * 8));     // [CH_MAXOPS] count of its component after the item
    float *means_l = reinterpret_cast<float *>(take((size_t)CPW * LDM * 4));            // [CPW][LDM]
    int32_t *bk = reinterpret_cast<int32_t *>(take((size_t)NBC * 4));                   // [NBC]
    int32_t *l_new = reinterpret_cast<int32_t *>(take((size_t)NM * 4));                 // [NM]
    int32_t *l_newk = reinterpret_cast<int32_t *>(take((size_t)NM * 4));                // [NM]
    int32_t *l_cnt = reinterpret_cast<int32_t *>(take(8 * 4));                          // [8]
    int32_t *op_k = reinterpret_cast<int32_t *>(take(CH_MAXOPS * 4));                   // [CH_MAXOPS] own items only: slot of the component; [j] old token of span end j + 1, [32 + j] new token
    int32_t *op_x = reinterpret_cast<int32_t *>(take(CH_MAXOPS * 4));                   // [CH_MAXOPS] its row: xo index (old), band entry (new)
    ChainSet S[2];
    for (int z = 0; z < 2; z++) {
        S[z].xs = reinterpret_cast<float *>(take((size_t)NBC * LDX * 4));
        S[z].xo = reinterpret_cast<float *>(take((size_t)NM * LDX * 4));
        S[z].bdur = reinterpret_cast<double *>(take((size_t)NBC * 8));
        S[z].bid = reinterpret_cast<int32_t *>(take((size_t)NBC * 4));
        S[z].basg = reinterpret_cast<int32_t *>(take((size_t)NBC * 4));
        S[z].o_e = reinterpret_cast<int32_t *>(take((size_t)NM * 4));
        S[z].o_a = reinterpret_cast<int32_t *>(take((size_t)NM * 4));
        S[z].meta = reinterpret_cast<int32_t *>(take(12 * 4));
    }
    __shared__ int sh_flag, sh_K, sh_empty;
    __shared__ unsigned long long sh_own[2];           // the old / new tokens (by span end) on components this workgroup owns

    // Everything phase A needs of utterance order[qn] into set `st` -- none of it depends on the utterances before it.
    // Wave `w_old` (all its 64 lanes) lists the old tokens; the threads t0 <= tid < t0 + nth fetch the candidates.  No
    // barrier inside: every thread works from global memory alone (the band table is read again for the rows).
    auto stage = [&](int u, int N, const ChainSet &st, int w_old, int t0, int nth) {
        const int W = (A.n_max > 0 && A.n_max < N) ? A.n_max : N;            // <= 8
        const int nb = N * W;
        const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
        const double *dur = c.durations + (int64_t)u * triMax;
        const bool band = c.band_ids != nullptr && c.band_W == W && W > 0;
        const int32_t *bandi = band ? c.band_ids + (int64_t)u * NM * c.band_W : nullptr;
        const double *bandd = band ? c.band_dur + (int64_t)u * NM * c.band_W : nullptr;
        auto entry_id = [&](int i) -> int {
            const int t = i / W + 1, w = i % W, s0 = t - 1 - w;
            if (s0 < 0) return -1;
            return band ? bandi[i] : vid[t * (t - 1) / 2 + s0];
        };
        if (wv == w_old) {
            const uint8_t *gbnd = A.boundaries + (int64_t)u * NM;
            // the lane of a set bit j looks up its own span [jp, j + 1) (utterances.py:159-174); the W entries of its row
            // are fetched beside the boundary flags, not after them
            int ent[8];
#pragma unroll
            for (int w = 0; w < 8; w++) ent[w] = (lane < N && w < W) ? entry_id(lane * W + w) : -1;
            const unsigned long long oldb = __ballot(lane < N && gbnd[lane < N ? lane : 0] != 0);
            const bool bit = lane < N && ((oldb >> lane) & 1ull);
            const unsigned long long below = oldb & ((1ull << lane) - 1ull);
            const int jp = below ? 64 - __clzll((long long)below) : 0;
            int id = -1;
            if (bit) {
                const int t = lane + 1, w = lane - jp;
                if (w < W) {
#pragma unroll
                    for (int z = 0; z < 8; z++)
                        if (z == w) id = ent[z];
                } else {
                    id = vid[t * (t - 1) / 2 + jp];
                }
            }
            const unsigned long long keep = __ballot(bit && id >= 0);
            const int no = __popcll(keep), r = __popcll(keep & ((1ull << lane) - 1ull));
            if (bit && id >= 0) {
                st.o_e[r] = id;
                st.o_a[r] = m.assignments[id];
            }
            WAVE_SYNC();
            // their rows, 16 bytes per lane and step
            for (int p = lane; p < no * D4; p += 64) {
                const int i = p / D4, d4 = p - i * D4;
                *reinterpret_cast<float4 *>(st.xo + i * LDX + 4 * d4) = *reinterpret_cast<const float4 *>(X + (int64_t)st.o_e[i] * c.ldx + 4 * d4);
            }
            if (lane == 0) {
                st.meta[0] = u; st.meta[1] = N; st.meta[2] = W; st.meta[3] = nb;
                st.meta[4] = (int32_t)(oldb & 0xffffffffull); st.meta[5] = (int32_t)(oldb >> 32);
                st.meta[6] = no; st.meta[7] = band ? 1 : 0;
                st.meta[8] = (int32_t)(keep & 0xffffffffull); st.meta[9] = (int32_t)(keep >> 32);
            }
        }
        const int tt = tid - t0;
        if (tt < 0 || tt >= nth) return;
        for (int i = tt; i < nb; i += nth) {
            const int id = entry_id(i);
            double dd = __builtin_nan("");
            if (id >= 0) {
                const int t = i / W + 1, w = i % W;
                dd = band ? bandd[i] : dur[t * (t - 1) / 2 + (t - 1 - w)];
            }
            st.bid[i] = id;
            st.bdur[i] = dd;
            // add_item's assert (:101) wants the row's label BEFORE this utterance: read ahead of the barrier -- once the
            // first workgroup is through its phase B the labels in memory are the new ones
            if (blockIdx.x == gridDim.x - 1) st.basg[i] = id >= 0 ? m.assignments[id] : -1;
        }
        // the candidate rows, by band entry: consecutive threads on consecutive 16 bytes, four loads in flight per thread
        const int tot = nb * D4;
        for (int p0 = tt; p0 < tot; p0 += 4 * nth) {
            int ids[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = p0 + r * nth;
                ids[r] = entry_id((p < tot ? p : 0) / D4);
            }
            float4 v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = p0 + r * nth, pp = p < tot ? p : 0;
                const int d4 = pp % D4;
                v[r] = *reinterpret_cast<const float4 *>(X + (int64_t)(ids[r] >= 0 ? ids[r] : 0) * c.ldx + 4 * d4);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = p0 + r * nth;
                if (p < tot) *reinterpret_cast<float4 *>(st.xs + (p / D4) * LDX + 4 * (p % D4)) = v[r];
            }
        }
    };

    // ---- this workgroup's components: float32 means (the score operand), float64 numerators, counts
    for (int q = tid; q < kn * D4; q += CH_THREADS) {
        const int ci = q / D4, d4 = q - ci * D4;
        *reinterpret_cast<float4 *>(means_l + ci * LDM + 4 * d4) = *reinterpret_cast<const float4 *>(means_g + (int64_t)(k0 + ci) * D + 4 * d4);
    }
    for (int q = tid; q < kn * D; q += CH_THREADS) numer_l[q] = m.mean_numerators[(int64_t)k0 * D + q];
    if (tid < kn) cnt_l[tid] = m.counts[k0 + tid];
    if (tid == 0) sh_K = *m.K;
    // the utterance numbers and lengths run two and one utterance ahead of the staging, so that it starts with the tables
    int u1 = -1, u2 = -1, N1 = 0;
    if (A.q0 < A.q1) {
        const int u0 = A.order[A.q0];
        stage(u0, c.lengths[u0], S[0], 0, 0, CH_THREADS);
        if (A.q0 + 1 < A.q1) u1 = A.order[A.q0 + 1];
    }
    __syncthreads();

    int phase = 0;
    int q = A.q0;
    for (; q < A.q1; q++) {
        if (q + 2 < A.q1) u2 = A.order[q + 2];
        if (q + 1 < A.q1) N1 = c.lengths[u1];
        const ChainSet &C = S[(q - A.q0) & 1];
        const int u = C.meta[0], N = C.meta[1], W = C.meta[2], nb = C.meta[3];
        const int32_t *bid = C.bid;
        const float *xs = C.xs, *xo = C.xo;
        uint8_t *gbnd = A.boundaries + (int64_t)u * NM;
        unsigned long long *keys = A.keys + (size_t)(q % 3) * NBC * CH_KEY_PITCH;
        CH_STAMP(0);
        CH_STAMP(1);

        // ================================ phase A: this workgroup's share of the scores, the components of a span on CPW
        // adjacent lanes
        // two components per thread (ci and ci + CPW / 2 of one span): the span's row is read once for both
        const int HW = CPW >> 1;
        for (int p0 = 0; p0 < nb * HW; p0 += CH_THREADS) {
            const int p = p0 + tid, i = p / HW, ci = p - i * HW;
            unsigned long long key = 0ull;
            if (i < nb && ci < kn && bid[i] >= 0) {
                const bool two = ci + HW < kn;
                float sa, sb;
                neg_sqd_exact_v4_pk2(means_l + ci * LDM, means_l + (two ? ci + HW : ci) * LDM, xs + i * LDX, D, &sa, &sb);
                auto pack = [&](float sc, int cc) -> unsigned long long {
                    const unsigned int bits = __float_as_uint(sc);
                    const unsigned int ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                    return ((unsigned long long)ord << 32) | (unsigned long long)(0xffffffffu - (unsigned int)(k0 + cc));
                };
                key = pack(sa, ci);
                if (two) {
                    const unsigned long long kb2 = pack(sb, ci + HW);
                    key = kb2 > key ? kb2 : key;
                }
            }
            for (int o = 1; o < HW; o <<= 1) {
                const unsigned long long other = __shfl_xor(key, o);
                key = other > key ? other : key;
            }
            // one atomic maximum per span and workgroup; every span's word on a 128-byte line of its own, so that the
            // 125 x 105 atomics of an utterance spread over the L2 channels
            if (ci == 0 && i < nb && key != 0ull) atomicMax(&keys[(size_t)i * CH_KEY_PITCH], key);
        }
        CH_STAMP(2);
        if (A.stamp && tid == 0 && q - A.q0 >= 100 && q - A.q0 < 104) A.stamp[2048 + (q - A.q0 - 100) * 256 + blockIdx.x] = wall_clock64();
        phase++;
        if (!chain_barrier(A.ctl, phase, &sh_flag, (A.stamp && blockIdx.x == 0 && q - A.q0 < 256) ? A.stamp + 3072 + (q - A.q0) * 8 + 5 : nullptr)) return;
        CH_STAMP(3);
        if (sh_flag & 2) break;            // a component emptied during the previous utterance: clean_components on the host's side
        // the maxima of the utterance before the previous one: every workgroup has read them (it passed this barrier after
        // its phase B), nobody writes them before the next barrier
        // (one word per workgroup: 120 write-through stores by ONE workgroup made it 2 us late at every barrier)
        if (tid == 0)
            for (int i = blockIdx.x; i < NBC; i += gridDim.x)
                __hip_atomic_store(&A.keys[((size_t)((q + 2) % 3) * NBC + i) * CH_KEY_PITCH], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        // ================================ phase B, the same in every workgroup: wave 0 runs the DP, the other waves fetch the
        // next utterance
        if (wv == 0) {
            // the spans' maxima, eight span ends per step: lane (r, w) takes entry (r + 1, w) -- every load first, then the
            // conversions (:346-351); the DP's image of the band has pitch 8, -inf beyond the window
            {
                const int w = lane & 7;
                unsigned long long key[4];
                int idv[4];
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    const int r = 8 * z + (lane >> 3);
                    key[z] = 0ull;
                    idv[z] = (r < N && w < W) ? bid[r * W + w] : -1;
                }
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    const int r = 8 * z + (lane >> 3);
                    if (idv[z] >= 0)
                        key[z] = __hip_atomic_load(&keys[(size_t)(r * W + w) * CH_KEY_PITCH], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    const int r = 8 * z + (lane >> 3);
                    if (r < N) {
                        double v = NEG_INF_D;
                        if (idv[z] >= 0) {
                            const unsigned int ord = (unsigned int)(key[z] >> 32);
                            const unsigned int bits = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
                            const double dd = C.bdur[r * W + w];
                            v = isnan(dd) ? NEG_INF_D : (double)__uint_as_float(bits) * dd;
                        }
                        if (w < W) bk[r * W + w] = idv[z] >= 0 ? (int32_t)(0xffffffffu - (unsigned int)(key[z] & 0xffffffffu)) : -1;
                        bvec[r * 8 + w] = v + A.wip;
                    }
                }
            }
            WAVE_SYNC();
            CH_STAMP(4);
            double total;
            unsigned long long newb, keepN;
            int eN, kN, xN, n_flagged;
            seg_w8_uniform(bvec, gam, bid, bk, N, W, sh_K, l_new, l_newk, l_cnt, &total, lane, newb, keepN, eN, kN, xN, n_flagged,
                           (A.stamp && blockIdx.x == 0 && q - A.q0 < 256) ? A.stamp + 3072 + (q - A.q0) * 8 : nullptr);
            if (A.stamp && blockIdx.x == 0 && lane == 0 && q - A.q0 < 256) A.stamp[(q - A.q0) * 8 + 7] = wall_clock64();
            // ---- the operations of the utterance in the reference's order: del_item of the old tokens, add_item of the new ones
            // (kmeans_components.py:93-132), both in span-end order.  The lane of span end j + 1 holds the old token and the new
            // token that end there; a span that is deleted and added again sits on ONE lane (rows are per span).
            const int no = C.meta[6], nn = __popcll(keepN);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (n_flagged != 0) {                                       // tokens on inactive components: add_item's clamp (:102-106), in order
                WAVE_SYNC();
                if (lane == 0) {
                    int K = sh_K;
                    for (int t2 = 0; t2 < nn; t2++) {
                        int k = l_newk[t2];
                        if (k > K) k = K;
                        if (k == K) K++;
                        l_newk[t2] = k;
                    }
                    sh_K = K;
                }
                WAVE_SYNC();
                if (eN >= 0) kN = l_newk[__popcll(keepN & lt)];
            }
            const unsigned long long keepO = ((unsigned long long)(unsigned int)C.meta[9] << 32) | (unsigned int)C.meta[8];
            const bool vO = (keepO >> lane) & 1ull;
            const int rO = __popcll(keepO & lt);
            int eO = -1, kO = -1;
            if (vO) { eO = C.o_e[rO]; kO = C.o_a[rO]; }
            const bool vN = eN >= 0;
            // labels (every workgroup: the same values, whichever L2 a later read hits holds them)
            if (vN) m.assignments[eN] = kN;
            if (vO && eO != eN) m.assignments[eO] = -1;
            // add_item's assert (:101): the row must be unassigned -- by this utterance's own del_item, else by the state before it
            // (the last workgroup checks, workgroup 0 writes the outputs)
            if (blockIdx.x == gridDim.x - 1 && vN && !(vO && eO == eN) && C.basg[xN] != -1) atomicOr(A.status, 2);
            // the items on the components this workgroup owns; the count of its component after every one of them
            const bool ownO = vO && kO >= k0 && kO < k0 + kn, ownN = vN && kN >= k0 && kN < k0 + kn;
            const unsigned long long mO = __ballot(ownO), mN = __ballot(ownN);
            if (ownO) { op_k[lane] = kO - k0; op_x[lane] = rO; }
            if (ownN) { op_k[32 + lane] = kN - k0; op_x[32 + lane] = xN; }
            WAVE_SYNC();
            if (lane == 0) {
                for (unsigned long long mm = mO; mm; mm &= mm - 1ull) {
                    const int j = __ffsll((long long)mm) - 1, ci = op_k[j];
                    const long long cn = cnt_l[ci] - 1;
                    cnt_l[ci] = cn;
                    op_cnt[j] = cn;
                }
                for (unsigned long long mm = mN; mm; mm &= mm - 1ull) {
                    const int j = __ffsll((long long)mm) - 1, ci = op_k[32 + j];
                    const long long cn = cnt_l[ci] + 1;
                    cnt_l[ci] = cn;
                    op_cnt[32 + j] = cn;
                }
                sh_own[0] = mO;
                sh_own[1] = mN;
                sh_empty = 0;
                if (blockIdx.x == 0) {
                    A.out_total[u] = total;
                    A.n_old[u] = no;
                    A.n_new[u] = nn;
                    if (A.n_flag) A.n_flag[u] = n_flagged;
                    if (l_cnt[5]) atomicOr(A.status, 1);
                }
            }
            WAVE_SYNC();
            if (ownO || ownN) {
                // (a lane that owns both writes the same final count twice)
                const int ciO = ownO ? kO - k0 : kN - k0, ciN = ownN ? kN - k0 : kO - k0;
                const long long cO = cnt_l[ciO], cNn = cnt_l[ciN];
                m.counts[k0 + ciO] = cO;
                m.counts[k0 + ciN] = cNn;
                if ((cO == 0 && k0 + ciO < sh_K) || (cNn == 0 && k0 + ciN < sh_K)) sh_empty = 1;
            }
        } else if (q + 1 < A.q1) {
            stage(u1, N1, S[(q + 1 - A.q0) & 1], 1, 128, CH_THREADS - 128);
        }
        u1 = u2;
        __syncthreads();
        CH_STAMP(5);
        // the items in the reference's order; thread d owns dimension d of every component of this workgroup (:110, :128-129), and
        // writes the rows through as it goes
        if (tid < D) {
            const int d = tid;
            for (int ph = 0; ph < 2; ph++)
                for (unsigned long long mm = sh_own[ph]; mm; mm &= mm - 1ull) {
                    const int o = 32 * ph + __ffsll((long long)mm) - 1;
                    const int ci = op_k[o];
                    const double x = (double)(ph == 0 ? xo[op_x[o] * LDX + d] : xs[op_x[o] * LDX + d]);
                    const double v = ph == 0 ? numer_l[ci * D + d] - x : numer_l[ci * D + d] + x;
                    numer_l[ci * D + d] = v;
                    m.mean_numerators[(int64_t)(k0 + ci) * D + d] = v;
                    const long long cnt = op_cnt[o];
                    if (cnt != 0) {
                        const float mu = (float)(v / (double)cnt);
                        means_l[ci * LDM + d] = mu;
                        means_g[(int64_t)(k0 + ci) * D + d] = mu;
                    }
                }
        }
        // boundaries (every workgroup: the same values) and the utterance's outputs, by threads the loop above leaves idle
        if (tid >= 512 && tid - 512 < N) {
            const unsigned long long newb = ((unsigned long long)(unsigned int)l_cnt[3] << 32) | (unsigned int)l_cnt[2];
            gbnd[tid - 512] = (uint8_t)((newb >> (tid - 512)) & 1ull);
        }
        if (blockIdx.x == 0 && tid >= 576) {
            const int j = tid - 576, no = C.meta[6], nn = l_cnt[1];
            if (j < no) A.old_tok[(int64_t)u * NM + j] = C.o_e[j];
            if (j < NM) {
                if (j < nn) A.new_tok[(int64_t)u * NM + j] = l_new[j];
                A.new_k[(int64_t)u * NM + j] = j < nn ? l_newk[j] : -1;
            }
        }
        __syncthreads();
        if (tid == 0 && sh_empty) {
            __hip_atomic_fetch_or(&A.ctl[0], CH_STOP_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the host
        }
        // (no barrier here: the next one is the grid barrier's, and until then nothing below is written again)
        CH_STAMP(6);
    }
    if (blockIdx.x == 0 && tid == 0) {
        A.ctl[2] = q;                    // utterances order[q0 .. q) are done
        *m.K = sh_K;
    }
}

// One sweep over order[0 .. n_order) (host array): launches of the persistent kernel, a clean_components launch whenever one
// of them stopped because a component emptied.  Returns SEGK_ERR_UNSUPPORTED (nothing enqueued) when the configuration is
// outside the kernel's reach.
int segk_launch_seq_chain(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, const int32_t *order, int n_order, int n_slices_max,
                          double wip, uint8_t *boundaries, int32_t *old_tok, int32_t *new_tok, int32_t *new_k, int32_t *n_old,
                          int32_t *n_new, int32_t *n_flag, double *out_total, int32_t *status, hipStream_t st)
{
    if (c->x_dtype != SEGK_F32 || (c->D & 3) || c->D < 8 || c->D > 128 || (c->ldx & 3) || c->N_max > 32 || n_slices_max < 1 ||
        n_slices_max > 8 || m->K_max < 1 || c->N_max < 1)
        return SEGK_ERR_UNSUPPORTED;
    const int W = n_slices_max < c->N_max ? n_slices_max : c->N_max;
    const int nbc = c->N_max * W;
    // components per workgroup: 8 from K_max = 512 on (125 workgroups for the headline model: measured 22.1 us per utterance
    // against 23.9 with 16 and 27.6 with 32 -- the score phase is bound by LDS bandwidth, 800 bytes per (span, component))
    int G = ctx->n_cu / 2;
    if (G < 1) G = 1;
    int cpw = (m->K_max + G - 1) / G;
    int cp2 = 1;
    while (cp2 < cpw) cp2 <<= 1;                            // the components of a span sit on a power-of-two group of lanes
    cpw = cp2 < 8 ? 8 : cp2;
    if (cpw > 64) return SEGK_ERR_UNSUPPORTED;
    G = (m->K_max + cpw - 1) / cpw;
    const int D = c->D, ldm = ((D >> 2) & 1) ? D : D + 4;
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t NM = (size_t)c->N_max;
    const size_t set_bytes = al((size_t)nbc * D * 4) + al(NM * D * 4) + al((size_t)nbc * 8) + 2 * al((size_t)nbc * 4) + 2 * al(NM * 4) + al(12 * 4);
    const size_t lds = al((size_t)cpw * D * 8) + al(NM * 8 * 8) + al((NM + 12) * 8) + al((size_t)cpw * 8) + al(CH_MAXOPS * 8) +
                       al((size_t)cpw * ldm * 4) + al((size_t)nbc * 4) + 2 * al(NM * 4) + al(8 * 4) + 2 * al(CH_MAXOPS * 4) + 2 * set_bytes;
    if (lds > 158 * 1024) return SEGK_ERR_UNSUPPORTED;
    SEGK_CHECK_HIP(segk_dyn_lds((const void *)k_seq_chain, lds));
    // all workgroups must be resident together: one per CU at most
    if (G > ctx->n_cu || (int64_t)G * (n_order + 1) >= CH_STOP_BIT) return SEGK_ERR_UNSUPPORTED;

    const size_t key_bytes = 3 * (size_t)nbc * CH_KEY_PITCH * sizeof(unsigned long long), ctl_bytes = (8 + 32 * (CH_FLAGS + 1)) * sizeof(int32_t);
    // (behind the order: the relabelling log of clean_components, 1 + 2 K_max words)
    const size_t order_bytes = ((size_t)n_order * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t need = key_bytes + ctl_bytes + order_bytes + (size_t)(2 * m->K_max + 2) * sizeof(int32_t);
    if (ctx->chain_bytes < need) {
        if (ctx->chain_buf) (void)hipFree(ctx->chain_buf);
        ctx->chain_buf = nullptr;
        ctx->chain_bytes = 0;
        SEGK_CHECK_HIP(hipMalloc(&ctx->chain_buf, need));
        ctx->chain_bytes = need;
    }
    unsigned char *buf = (unsigned char *)ctx->chain_buf;
    ChainArgs A{};
    A.c = *c; A.m = *m;
    A.keys = (unsigned long long *)buf;
    A.ctl = (int32_t *)(buf + key_bytes);
    A.order = (const int32_t *)(buf + key_bytes + ctl_bytes);
    A.n_max = n_slices_max; A.wip = wip;
    A.boundaries = boundaries; A.old_tok = old_tok; A.new_tok = new_tok; A.new_k = new_k; A.n_old = n_old; A.n_new = n_new;
    A.n_flag = n_flag; A.out_total = out_total; A.status = status;
    A.cpw = cpw; A.nb_cap = nbc; A.ldm = ldm;
    SEGK_CHECK_HIP(hipMemcpyAsync((void *)A.order, order, (size_t)n_order * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const bool stamping = segk_dev_env("SEGK_CHAIN_STAMP") != 0;      // -DSEGK_DEV builds only
    static unsigned long long *stamp_dev = nullptr;
    if (stamping && !stamp_dev) SEGK_CHECK_HIP(hipMalloc((void **)&stamp_dev, (256 * 8 + 4 * 256 + 256 * 8) * sizeof(unsigned long long)));
    A.stamp = stamping ? stamp_dev : nullptr;
    int q = 0;
    while (q < n_order) {
        SEGK_CHECK_HIP(hipMemsetAsync(buf, 0, key_bytes + ctl_bytes, st));
        A.q0 = q; A.q1 = n_order;
        hipLaunchKernelGGL(k_seq_chain, dim3(G), dim3(CH_THREADS), lds, st, A);
        SEGK_LAUNCH_CHECK();
        int32_t ctl[8];
        SEGK_CHECK_HIP(hipMemcpyAsync(ctl, A.ctl, sizeof(ctl), hipMemcpyDeviceToHost, st));
        SEGK_CHECK_HIP(hipStreamSynchronize(st));
        if (ctl[3] != 0) {
            segk_set_error("sequential chain: grid barrier timed out (%d workgroups were not resident together?)", G);
            return SEGK_ERR_HIP;
        }
        if (ctl[2] <= q && ctl[1] == 0) {
            segk_set_error("sequential chain: no progress at utterance %d", q);
            return SEGK_ERR_HIP;
        }
        if (stamping && ctl[2] - q >= 200) {          // a long launch: mean time between the stamps, in 10 ns ticks of the 100 MHz clock
            static unsigned long long hs[256 * 8];
            SEGK_CHECK_HIP(hipMemcpy(hs, stamp_dev, sizeof(hs), hipMemcpyDeviceToHost));
            {
                static unsigned long long ar[4 * 256];
                SEGK_CHECK_HIP(hipMemcpy(ar, stamp_dev + 2048, sizeof(ar), hipMemcpyDeviceToHost));
                for (int z = 0; z < 4; z++) {
                    unsigned long long mn = ~0ull, mx = 0;
                    int imx = 0;
                    for (int g = 0; g < G; g++) {
                        if (ar[z * 256 + g] < mn) mn = ar[z * 256 + g];
                        if (ar[z * 256 + g] > mx) { mx = ar[z * 256 + g]; imx = g; }
                    }
                    int late = 0;
                    for (int g = 0; g < G; g++) late += (ar[z * 256 + g] - mn) > 200;
                    fprintf(stderr, "  arrivals at barrier %d: spread %.2f us, workgroup 0 at +%.2f, last is workgroup %d, %d of %d later than +2 us\n",
                            100 + z, (double)(mx - mn) / 100, (double)(ar[z * 256] - mn) / 100, imx, late, G);
                }
            }
            double acc[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int i = 20; i < 199; i++) {
                for (int j = 0; j < 6; j++) acc[j] += (double)(hs[i * 8 + j + 1] - hs[i * 8 + j]);
                acc[6] += (double)(hs[(i + 1) * 8] - hs[i * 8]);
            }
            {
                double dpw = 0;
                for (int i = 20; i < 199; i++) dpw += (double)(hs[i * 8 + 7] - hs[i * 8 + 4]);
                fprintf(stderr, "  of dp: seg_w8_uniform %.2f us\n", dpw / 179 / 100);
                static unsigned long long us[256 * 8];
                SEGK_CHECK_HIP(hipMemcpy(us, stamp_dev + 3072, sizeof(us), hipMemcpyDeviceToHost));
                double ph[5] = {0, 0, 0, 0, 0};
                for (int i = 20; i < 199; i++) {
                    ph[0] += (double)(us[i * 8] - hs[i * 8 + 4]);
                    for (int j = 1; j < 5; j++) ph[j] += (double)(us[i * 8 + j] - us[i * 8 + j - 1]);
                }
                double aw = 0;
                for (int i = 20; i < 199; i++) aw += (double)(us[i * 8 + 5] - hs[i * 8 + 2]);
                fprintf(stderr, "    of barrier: the wait for the atomics + workgroup barrier %.2f us\n", aw / 179 / 100);
                fprintf(stderr, "    (entry) %.2f  forward %.2f  decisions %.2f  walk %.2f  new tokens %.2f\n", ph[0] / 179 / 100,
                        ph[1] / 179 / 100, ph[2] / 179 / 100, ph[3] / 179 / 100, ph[4] / 179 / 100);
            }
            fprintf(stderr, "chain stamps (us): score %.2f  barrier %.2f  keys %.2f  dp %.2f  update %.2f  | per utterance %.2f\n",
                    (acc[0] + acc[1]) / 179 / 100, acc[2] / 179 / 100, acc[3] / 179 / 100, acc[4] / 179 / 100, acc[5] / 179 / 100,
                    acc[6] / 179 / 100);
        }
        q = ctl[2];
        if (ctl[1] != 0) {               // a component emptied: clean_components, then on with the next utterance
            int rc = segk_launch_clean(c, m, status, st, (int32_t *)(buf + key_bytes + ctl_bytes + order_bytes));
            if (rc) return rc;
        }
    }
    return SEGK_OK;
}
