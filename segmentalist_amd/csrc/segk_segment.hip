// segk_segment.hip -- per-utterance kernels: A5 vector + A8 max-plus DP + tokens (band layout), function-level DPs (triangular layout)
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"
#include "segk_segment_dev.h"

// Per-utterance kernel: A5 (vec from the candidates), A8 (max-plus DP), tokens.
//   ONE WAVE per utterance, no workgroup barriers: lanes gather the band of candidate spans,
//   lane 0 runs the DP on LDS, lanes write the results.
//   band layout: entry (t, w), t = 1..N (span end), w = 0..W-1 (span length w+1, start
//   s = t-1-w) at [(t-1)*W + w]; W = n_slices_max, or N when n_slices_max == 0.
// ======================================================================================

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
    // banded span tables (segk_corpus.band_ids / band_dur), when they were built for this window
    const bool band = c.band_ids != nullptr && c.band_W == W && W > 0;
    const int32_t *bandi = band ? c.band_ids + (int64_t)u * c.N_max * c.band_W : nullptr;
    const double *bandd = band ? c.band_dur + (int64_t)u * c.N_max * c.band_W : nullptr;
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
            id = band ? bandi[i] : vid[j];               // banded image: lane i reads entry i
            if (id >= 0) {
                k = cand.k[id];
                const double dd = band ? bandd[i] : dur[j];
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
    // slots beyond the utterance's tokens carry k = -1: the batch statistics scan new_k as it stands
    for (int j = lane; j < c.N_max; j += 64) {
        if (j < nn) new_tok[(int64_t)u * c.N_max + j] = l_new[j];
        new_k[(int64_t)u * c.N_max + j] = j < nn ? l_newk[j] : -1;
    }
}

// Fast path of the per-utterance kernel for windows of at most 8 slices and utterances of at most 64
// landmarks (the common configuration: n_slices_max = 6).  Same arithmetic and the same decisions as
// k_kmeans_segment; what changes is how lane 0 gets at its operands.  The generic kernel walks the
// DP as a chain of dependent LDS round trips (store gamma[t], load it back for t+1, byte loads of the
// boundary flags with a branch on each); here the last eight gammas live in registers, the eight
// candidates of a step are fetched together (predicated, fully unrolled), and the boundary vectors
// are 64-bit masks.
__global__ __launch_bounds__(256) void k_kmeans_segment_w8(segk_corpus c, segk_kmeans m, const int32_t *utts, int utt0, int n_utts,
                                    int n_max, double wip, segk_cand cand, uint8_t *boundaries, int32_t *old_tok,
                                    int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag,
                                    double *out_total, int32_t *status, int band_cap, int wave_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * (blockDim.x >> 6) + wv;
    SEGK_TSTAMP(0, 0);
    if (slot >= n_utts) return;
    const int u = utts ? utts[slot] : utt0 + slot;
    const int N = c.lengths[u];
    const int W = (n_max > 0 && n_max < N) ? n_max : N;          // <= 8 (host checks n_max <= 8)
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
    const double *dur = c.durations + (int64_t)u * triMax;
    // banded span tables (segk_corpus.band_ids / band_dur), when they were built for this window
    const bool band = c.band_ids != nullptr && c.band_W == W && W > 0;
    const int32_t *bandi = band ? c.band_ids + (int64_t)u * c.N_max * c.band_W : nullptr;
    const double *bandd = band ? c.band_dur + (int64_t)u * c.N_max * c.band_W : nullptr;
    uint8_t *gbnd = boundaries + (int64_t)u * c.N_max;

    char *base = smem + (size_t)wv * wave_bytes;
    double *bvec = (double *)base;                    // [N_max][8]: the DP's candidates, pitch 8, -inf beyond the window (seg_w8_uniform)
    double *gam = bvec + c.N_max * 8;                 // [8 + N_max + 4]
    int32_t *bk = (int32_t *)(gam + c.N_max + 12);    // [band_cap]
    int32_t *bid = bk + band_cap;                     // [band_cap]
    int32_t *l_old = bid + band_cap;                  // [N_max]
    int32_t *l_new = l_old + c.N_max;                 // [N_max]
    int32_t *l_newk = l_new + c.N_max;                // [N_max]
    int32_t *l_cnt = l_newk + c.N_max;                // [6]: n_old, n_new, new boundary mask (2 words), flagged, bad

    // the band (A5, kmeans_acoustic_wordseg.py:334-351): lane (r, w) takes entry (r + 1, w), 32 span ends per batch: every span id
    // first, then every gather -- two round trips per batch (a loop over the entries with id, label and score in turn was three
    // per 64 entries)
    {
        const int w = lane & 7;
        for (int z0 = 0; z0 < N; z0 += 32) {
            int id[4], kq[4];
            double dd[4], sc[4];
#pragma unroll
            for (int z = 0; z < 4; z++) {
                const int r = z0 + 8 * z + (lane >> 3), t = r + 1, sp = t - 1 - w;
                id[z] = -1;
                dd[z] = 0.0;
                if (r < N && w < W && sp >= 0) {
                    const int j = t * (t - 1) / 2 + sp;
                    id[z] = band ? bandi[r * W + w] : vid[j];
                    dd[z] = band ? bandd[r * W + w] : dur[j];
                }
            }
#pragma unroll
            for (int z = 0; z < 4; z++) {
                kq[z] = -1;
                sc[z] = 0.0;
                if (id[z] >= 0) {
                    kq[z] = cand.k[id[z]];
                    sc[z] = cand.s[id[z]];
                }
            }
#pragma unroll
            for (int z = 0; z < 4; z++) {
                const int r = z0 + 8 * z + (lane >> 3);
                if (r < N) {
                    double v = NEG_INF_D;
                    if (id[z] >= 0) v = isnan(dd[z]) ? NEG_INF_D : sc[z] * dd[z];      // :346-349
                    if (w < W) {
                        bid[r * W + w] = id[z];
                        bk[r * W + w] = kq[z];
                    }
                    bvec[r * 8 + w] = v + wip;                                          // :351
                }
            }
        }
    }
    const unsigned long long oldb = __ballot(lane < N && gbnd[lane < N ? lane : 0] != 0);
    WAVE_SYNC();
    SEGK_TSTAMP(0, 1);
    {
        const int no_ = seg_old_tokens_wave(bid, vid, N, W, oldb, l_old, lane);
        if (lane == 0) l_cnt[0] = no_;
    }
    double total;
    {
        unsigned long long newb_, keep_;
        int e_, k_, x_, f_;
        seg_w8_uniform(bvec, gam, bid, bk, N, W, *m.K, l_new, l_newk, l_cnt, &total, lane, newb_, keep_, e_, k_, x_, f_);
    }
    WAVE_SYNC();
    SEGK_TSTAMP(0, 2);
    if (lane == 0) {
        out_total[u] = total;
        n_old[u] = l_cnt[0];
        n_new[u] = l_cnt[1];
        if (n_flag) n_flag[u] = l_cnt[4];
        if (l_cnt[5]) atomicOr(status, 1);
    }
    WAVE_SYNC();
    const int no = l_cnt[0], nn = l_cnt[1];
    const unsigned long long newb = ((unsigned long long)(unsigned int)l_cnt[3] << 32) | (unsigned int)l_cnt[2];
    if (lane < N) gbnd[lane] = (uint8_t)((newb >> lane) & 1ull);
    for (int j = lane; j < no; j += 64) old_tok[(int64_t)u * c.N_max + j] = l_old[j];
    // slots beyond the utterance's tokens carry k = -1: the batch statistics scan new_k as it stands
    for (int j = lane; j < c.N_max; j += 64) {
        if (j < nn) new_tok[(int64_t)u * c.N_max + j] = l_new[j];
        new_k[(int64_t)u * c.N_max + j] = j < nn ? l_newk[j] : -1;
    }
    SEGK_TSTAMP_MAX(0, 3);
}

// EIGHT utterances per wave (round 3; launches over many utterances): lanes 8g..8g+7 run utterance g's DP -- the DPP steps of
// round 2's eight-lane DP act inside groups of eight lanes already, so the eight DPs advance in lockstep in one instruction stream, and a
// launch over 10 000 utterances is 1 250 waves instead of 10 000.  (One utterance per wave took 29 us: 2 500 workgroups to
// dispatch, 14 us of life each -- gathers 5.7, DP 7.9 under that load.)  Token lists by the group's eight lanes, eight
// boundary bits per step; the backward pass as a state machine with one candidate evaluation per trip, so that groups in
// different phases share the loop.  Same values and decisions as seg_w8_uniform.
__global__ __launch_bounds__(256) void k_kmeans_segment_oct(segk_corpus c, segk_kmeans m, const int32_t *utts, int utt0, int n_utts,
                                                            int n_max, double wip, segk_cand cand, uint8_t *boundaries, int32_t *old_tok,
                                                            int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag,
                                                            double *out_total, int32_t *status, int band_cap, int utt_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 3, w8 = lane & 7;
    const int slot = (blockIdx.x * (blockDim.x >> 6) + wv) * 8 + g;
    SEGK_TSTAMP(0, 0);
    const bool valid = slot < n_utts;
    const int u = valid ? (utts ? utts[slot] : utt0 + slot) : (utts ? utts[0] : utt0);
    const int N = valid ? c.lengths[u] : 0;
    const int W = (n_max > 0 && n_max < N) ? n_max : N;          // <= 8 (host checks n_max <= 8)
    const int nb = N * W;
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    const int32_t *vid = c.vec_ids + (int64_t)u * triMax;
    const double *dur = c.durations + (int64_t)u * triMax;
    const bool band = c.band_ids != nullptr && c.band_W == W && W > 0;
    const int32_t *bandi = band ? c.band_ids + (int64_t)u * c.N_max * c.band_W : nullptr;
    const double *bandd = band ? c.band_dur + (int64_t)u * c.N_max * c.band_W : nullptr;
    uint8_t *gbnd = boundaries + (int64_t)u * c.N_max;
    const int Kact = *m.K;
    const int gsh = 8 * g;                                          // this group's byte of a ballot
    const unsigned long long lt8 = (1ull << w8) - 1ull;

    char *base = smem + (size_t)((wv * 8 + g)) * utt_bytes;
    double *bvec = (double *)base;                    // [band_cap]
    double *gam = bvec + band_cap;                    // [N_max + 1]
    int32_t *bk = (int32_t *)(gam + c.N_max + 1);     // [band_cap]
    int32_t *bid = bk + band_cap;                     // [band_cap]
    int32_t *l_old = bid + band_cap;                  // [N_max]
    int32_t *l_new = l_old + c.N_max;                 // [N_max]
    int32_t *l_newk = l_new + c.N_max;                // [N_max]

    // ---- the band (A5, kmeans_acoustic_wordseg.py:334-351): eight entries of each utterance per step, sixteen steps (128
    // entries) at a time: every span id first, then every gather -- two round trips for the whole band
    int nb_wave = nb;
    nb_wave = max(nb_wave, __shfl_xor(nb_wave, 8));
    nb_wave = max(nb_wave, __shfl_xor(nb_wave, 16));
    nb_wave = max(nb_wave, __shfl_xor(nb_wave, 32));
    for (int i0 = 0; i0 < nb_wave; i0 += 128) {
        constexpr int NQ = 16;
        int id[NQ], kq[NQ];
        double dd[NQ], sc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int i = i0 + q * 8 + w8;
            id[q] = -1;
            if (i < nb) {
                const int t = i / W + 1, w = i % W, s = t - 1 - w;
                if (s >= 0) id[q] = band ? bandi[i] : vid[t * (t - 1) / 2 + s];      // banded image: consecutive lanes, consecutive entries
                dd[q] = s >= 0 ? (band ? bandd[i] : dur[t * (t - 1) / 2 + s]) : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            kq[q] = -1;
            sc[q] = 0.0;
            if (id[q] >= 0) {
                kq[q] = cand.k[id[q]];
                sc[q] = cand.s[id[q]];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int i = i0 + q * 8 + w8;
            if (i < nb) {
                double v = NEG_INF_D;
                if (id[q] >= 0) v = isnan(dd[q]) ? NEG_INF_D : sc[q] * dd[q];           // :346-349
                bid[i] = id[q];
                bk[i] = kq[q];
                bvec[i] = v + wip;                                                    // :351
            }
        }
    }
    // ---- old boundary mask, eight bits per step
    int N_wave = N;
    N_wave = max(N_wave, __shfl_xor(N_wave, 8));
    N_wave = max(N_wave, __shfl_xor(N_wave, 16));
    N_wave = max(N_wave, __shfl_xor(N_wave, 32));
    unsigned long long oldb = 0ull;
    for (int jj = 0; jj < N_wave; jj += 8) {
        const int j = jj + w8;
        const unsigned long long bal = __ballot(j < N && gbnd[j < N ? j : 0] != 0);
        oldb |= ((bal >> gsh) & 0xFFull) << jj;
    }
    WAVE_SYNC();
    // the tokens of a boundary mask: the lane of bit j looks up its span [jp, j + 1)
    auto tokens = [&](unsigned long long mask, int32_t *out_id, int32_t *out_k, int &n_out, int &n_fl, int &bad, bool is_new) {
        int n = 0, nf = 0, bd = 0;
        for (int jj = 0; jj < N_wave; jj += 8) {
            const int j = jj + w8;
            const bool bit = j < N && ((mask >> j) & 1ull);
            const unsigned long long below = mask & ((1ull << j) - 1ull);
            const int jp = below ? 64 - __clzll((long long)below) : 0;
            const int w = j - jp;
            int id = -1, kk = -1;
            if (bit) {
                if (w < W) {
                    id = bid[j * W + w];
                    kk = bk[j * W + w];
                } else if (!is_new) {
                    id = vid[(j + 1) * j / 2 + jp];              // an old span longer than the window: the triangular table
                }
            }
            const bool ok = bit && id >= 0;
            const unsigned long long keep = (__ballot(ok) >> gsh) & 0xFFull;
            const unsigned long long badm = (__ballot(bit && !ok) >> gsh) & 0xFFull;
            const unsigned long long flm = (__ballot(ok && kk >= Kact) >> gsh) & 0xFFull;
            if (ok) {
                const int rk = n + __popcll(keep & lt8);
                out_id[rk] = id;
                if (out_k) out_k[rk] = kk;
            }
            n += __popcll(keep);
            nf += __popcll(flm);
            bd |= badm != 0ull;
        }
        n_out = n;
        n_fl = nf;
        bad = bd;
    };
    SEGK_TSTAMP(0, 1);
    int no, dummy1, dummy2;
    tokens(oldb, l_old, nullptr, no, dummy1, dummy2, false);
    // ---- A8 forward (kmeans_acoustic_wordseg.py:494-506): lane w of a group holds gamma[t - 1 - w].  The candidates of span
    // end t are the ones the backward pass would evaluate again (:510-553), so the step also records its decision: kbarr[t] =
    // length of the best span ending at t (first maximum in w order: the shortest span on ties), 0 when every candidate is -inf
    uint8_t *kbarr = (uint8_t *)(l_newk + c.N_max);                  // [N_max + 1] (the one-per-wave kernel's counters + tail)
    double gw = w8 == 0 ? 0.0 : NEG_INF_D;
    if (w8 == 0 && N > 0) gam[0] = 0.0;
    double vn = nb > 0 ? bvec[0] : NEG_INF_D;
    for (int t = 1; t <= N_wave; t++) {
        const bool act = t <= N;
        const bool ok = act && w8 < W && t - 1 - w8 >= 0;
        const double v = vn;
        {
            const bool okn = w8 < W && t - w8 >= 0 && t + 1 <= N;
            vn = bvec[okn ? t * W + w8 : 0];
        }
        const double x = ok ? v + gw : NEG_INF_D;
        const unsigned long long fin = (__ballot(ok && x != NEG_INF_D) >> gsh) & 0xFFull;
        double mx = x;
        mx = fmax(mx, seg_dpp_f64<SEG_DPP_XOR1>(mx));
        mx = fmax(mx, seg_dpp_f64<SEG_DPP_XOR2>(mx));
        mx = fmax(mx, seg_dpp_f64<SEG_DPP_HMIRROR>(mx));               // the group's eight lanes hold the maximum
        const unsigned long long at = (__ballot(ok && x == mx) >> gsh) & 0xFFull;
        if (w8 == 0 && act) {
            if (t < N) gam[t] = mx;
            kbarr[t] = (uint8_t)(fin ? __ffsll((long long)at) : 0);
        }
        const double gs = seg_dpp_f64<SEG_DPP_SHR1>(gw);            // lane 0 of a group reads its neighbour group's lane 7: overwritten below
        if (act) gw = w8 == 0 ? mx : gs;
    }
    WAVE_SYNC();
    // ---- A8 backward (:510-553): a walk over the recorded decisions; state = (t, searching)
    unsigned long long newb = N > 0 ? 1ull << (N - 1) : 0ull;
    int t = N;
    bool searching = false, done = N <= 0;
    double total = 0.0;
    while (__ballot(!done) != 0ull) {
        const int kb = done ? 1 : (int)kbarr[t];
        if (!done) {
            if (kb == 0) {                                           // every candidate -inf: step back (:516-530)
                t = t - 1;
                if (t == 0) {
                    newb |= 1ull << (N - 1);
                    total += bvec[(N - 1) * W];                      // python vec[-1]: the last span [N-1, N)
                    done = true;
                } else {
                    searching = true;
                }
            } else {
                if (searching) {
                    newb |= 1ull << (t - 1);
                    searching = false;
                }
                total += bvec[(t - 1) * W + (kb - 1)];
                if (t - kb - 1 < 0) done = true;
                else {
                    newb |= 1ull << (t - kb - 1);
                    t = t - kb;
                }
            }
        }
    }
    // ---- new tokens + their best components (:312-313)
    int nn, nfl, bad;
    tokens(newb, l_new, l_newk, nn, nfl, bad, true);
    WAVE_SYNC();
    SEGK_TSTAMP(0, 2);
    if (valid) {
        if (w8 == 0) {
            out_total[u] = total;
            n_old[u] = no;
            n_new[u] = nn;
            if (n_flag) n_flag[u] = nfl;
            if (bad) atomicOr(status, 1);
        }
        for (int j = w8; j < N; j += 8) gbnd[j] = (uint8_t)((newb >> j) & 1ull);
        for (int j = w8; j < no; j += 8) old_tok[(int64_t)u * c.N_max + j] = l_old[j];
        // slots beyond the utterance's tokens carry k = -1: the batch statistics scan new_k as it stands
        for (int j = w8; j < c.N_max; j += 8) {
            if (j < nn) new_tok[(int64_t)u * c.N_max + j] = l_new[j];
            new_k[(int64_t)u * c.N_max + j] = j < nn ? l_newk[j] : -1;
        }
    }
    SEGK_TSTAMP_MAX(0, 3);
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


extern "C" {

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
                        + (size_t)(2 * band_cap + 3 * c->N_max + 8) * sizeof(int32_t) + (size_t)c->N_max;
    wave_bytes = (wave_bytes + 15) & ~(size_t)15;
    // the one-utterance-per-wave kernel keeps the DP's candidates with pitch 8 and the gammas with eleven words of slack
    size_t w8_bytes = (size_t)(c->N_max * 8 + c->N_max + 12) * sizeof(double)
                      + (size_t)(2 * band_cap + 3 * c->N_max + 8) * sizeof(int32_t) + (size_t)c->N_max;
    w8_bytes = (w8_bytes + 15) & ~(size_t)15;
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
    (void)ctx;
    segk_tstamp_bind();
    // eight utterances per wave: SEGK_SEGMENT_OCT=1 only.  It was the shorter form from 4 096 utterances on (21 against 30 us
    // at 10 000) until the one-per-wave kernel got the chain's uniform DP and the batched band fetch: interleaved on one box the
    // one-per-wave form is now ahead at every size (10 000 utterances: 1 916 / 1 923 against 1 906 / 1 906 sweeps/s; 5 000:
    // 3 365 / 3 367 against 3 272 / 3 268)
    const char *oce = getenv("SEGK_SEGMENT_OCT");           // 1: eight utterances per wave (tests keep it covered)
    const int oct_mode = oce ? atoi(oce) : -1;
    if (w8_ok && oct_mode == 1 && 8 * wave_bytes <= 64 * 1024) {
        int ow = 4;
        while (ow > 1 && (size_t)ow * 8 * wave_bytes > 64 * 1024) ow >>= 1;
        const int per_block = 8 * ow;
        hipLaunchKernelGGL(k_kmeans_segment_oct, dim3((n_utts + per_block - 1) / per_block), dim3(64 * ow), (size_t)ow * 8 * wave_bytes, st,
                           *c, *m, utts, utt0, n_utts, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old, n_new,
                           n_flag, out_total, status, band_cap, (int)wave_bytes);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    if (n_slices_max >= 1 && n_slices_max <= 8 && c->N_max <= 64 && !(getenv("SEGK_SEGMENT_GENERIC") && atoi(getenv("SEGK_SEGMENT_GENERIC")))) {
        // four waves per workgroup (10 000 utterances, interleaved on one box: 1, 2, 4 and 8 waves within 0.2 % of each other,
        // 16 waves 0.6 % of the sweep behind)
        int w8w = 4;
        while (w8w > 1 && w8w * w8_bytes > 64 * 1024) w8w >>= 1;
        hipLaunchKernelGGL(k_kmeans_segment_w8, dim3((n_utts + w8w - 1) / w8w), dim3(64 * w8w), w8w * w8_bytes, st, *c, *m, utts, utt0,
                           n_utts, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old, n_new, n_flag,
                           out_total, status, band_cap, (int)w8_bytes);
        SEGK_LAUNCH_CHECK();
        return SEGK_OK;
    }
    hipLaunchKernelGGL(k_kmeans_segment, dim3((n_utts + waves - 1) / waves), dim3(64 * waves), lds, st, *c, *m, utts,
                       utt0, n_utts, n_slices_min, n_slices_max, wip, *cand, boundaries, old_tok, new_tok, new_k, n_old,
                       n_new, n_flag, out_total, status, band_cap, (int)wave_bytes);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
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
