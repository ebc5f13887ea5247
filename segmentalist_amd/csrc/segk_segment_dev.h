// segk_segment_dev.h -- device code shared by the per-utterance segmenter (segk_segment.hip) and the persistent
// sequential chain (segk_seq_chain.hip)
#pragma once
#include "segk_kmeans_dev.h"

#define WAVE_SYNC()                                             \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

// The window-of-eight segmenter of ONE utterance by a whole wave, on wave-private LDS arrays (k_kmeans_segment_w8 and the
// persistent sequential chain, segk_seq_chain.hip): A8 forward and backward, the new tokens and their components --
// seg_w8_uniform below.  (Round 2's form, seg_w8_wave, spread a step's eight candidates over eight lanes: one add, a three-step
// DPP maximum, a DPP shift of the gammas -- 230 clocks per step for a lone wave, 180 per backward token; k_kmeans_segment_oct
// still advances eight utterances that way in one wave.  A one-lane form before it: 14 us per utterance.)
// The DPP helpers:
template <int CTRL>
__device__ __forceinline__ double seg_dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi2, lo2);
}
template <int CTRL>
__device__ __forceinline__ int seg_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
#define SEG_DPP_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define SEG_DPP_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define SEG_DPP_HMIRROR 0x141      /* row_half_mirror: lane i <-> 7 - i inside every group of eight */
#define SEG_DPP_SHR1 0x111         /* row_shr:1: lane i reads lane i - 1 (lane 0 of a row keeps its own) */

// the tokens of a boundary mask, by a whole wave: the lane of a set bit j looks up its own span [jp, j + 1) (band entry, or the
// triangular table for a span longer than the window); spans without an embedding are skipped.  Returns their number.
__device__ __forceinline__ int seg_old_tokens_wave(const int32_t *bid, const int32_t *vid, int N, int W, unsigned long long oldb,
                                                   int32_t *l_old, int lane)
{
    const bool bit = lane < N && ((oldb >> lane) & 1ull);
    const unsigned long long below = oldb & ((1ull << lane) - 1ull);
    const int jp = below ? 64 - __clzll((long long)below) : 0;
    int id = -1;
    if (bit) {
        const int t = lane + 1, w = lane - jp;
        id = w < W ? bid[(t - 1) * W + w] : vid[t * (t - 1) / 2 + jp];
    }
    const unsigned long long keep = __ballot(bit && id >= 0);
    if (bit && id >= 0) l_old[__popcll(keep & ((1ull << lane) - 1ull))] = id;
    return __popcll(keep);
}

// ---- the same once more, for a wave that has nothing else to do (the persistent sequential chain: one utterance at a time, its
// latency IS the throughput).  A lone wave issues a dependent instruction every 8-16 clocks: the DPP form above (three
// cross-lane stages per step) measured 230 clocks a step and 180 a backward token -- 4.7 us per utterance inside the chain.
// Here the forward recurrence is UNIFORM: every lane keeps the last eight gammas in registers and computes all the
// candidates of a step itself -- WW independent adds and a tree of maxima, no cross-lane traffic; the candidates come from
// LDS as broadcast reads, one step ahead, into alternating registers.  The decisions of ALL span ends are then taken at
// once, lane t - 1 for span end t (the same adds on the same operands, so the values the forward pass saw: the first w whose
// candidate equals gamma[t]; 0 when gamma[t] is -inf), and the backward pass is a walk over lanes with v_readlane -- no
// candidate is evaluated twice, nothing on the walk touches LDS.
//   bvec8: [N][8]   candidate (t, w) at [(t - 1) * 8 + w]; -inf beyond the window (w >= W) and before the utterance's start
//   gamp:  [8 + N + 4]   gamp[8 + t] = gamma[t]; gamp[0..7] = -inf (written here); three more entries of slack
// bid / bk keep the pitch W of the band.  The old tokens are not listed here (the chain's staging lists them ahead of time);
// otherwise the values, decisions and outputs of round 2's seg_w8_wave (N <= 64): l_new, l_newk, l_cnt[1..5], and the lane of span end
// j + 1 gets its own new token back in registers.
__device__ __forceinline__ double seg_readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// the forward pass over a window of WW (2, 4, 6, 8) candidates
template <int WW>
__device__ __forceinline__ void seg_forward_uniform(const double *bvec8, double *gamp, int N, int lane)
{
    double ring[8];                                                    // ring[s] = the latest gamma[t'] with t' % 8 == s
#pragma unroll
    for (int s = 0; s < 8; s++) ring[s] = NEG_INF_D;
    ring[0] = 0.0;
    double va[WW], vb[WW];
    auto fetch = [&](double (&dst)[WW], int row) {
        const double2 *src = reinterpret_cast<const double2 *>(bvec8 + (size_t)row * 8);
#pragma unroll
        for (int h = 0; h < WW / 2; h++) {
            const double2 d = src[h];
            dst[2 * h] = d.x;
            dst[2 * h + 1] = d.y;
        }
    };
    fetch(va, 0);
    // four steps at a time without a test in between (a test per step kept the compiler from hoisting the next step's reads over
    // this step's arithmetic): up to three steps past N compute gammas nobody reads -- gamp has room for them
    for (int t0 = 1; t0 <= N; t0 += 8) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            if (half == 1 && t0 + 4 > N) break;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * half + jj;
                const int t = t0 + j;                                   // t % 8 == (1 + j) % 8
                double (&v)[WW] = (j & 1) ? vb : va;
                fetch((j & 1) ? va : vb, t < N ? t : N - 1);
                double x[WW];
#pragma unroll
                for (int w = 0; w < WW; w++) x[w] = v[w] + ring[(j - w) & 7];      // gamma[t - 1 - w]
#pragma unroll
                for (int h = 1; h < WW; h <<= 1)
#pragma unroll
                    for (int w = 0; w + h < WW; w += 2 * h) x[w] = fmax(x[w], x[w + h]);
                ring[(j + 1) & 7] = x[0];
                if (lane == 0) gamp[8 + t] = x[0];
            }
        }
    }
}
__device__ __forceinline__ void seg_w8_uniform(const double *bvec8, double *gamp, const int32_t *bid, const int32_t *bk, int N, int W,
                                               int Kact, int32_t *l_new, int32_t *l_newk, int32_t *l_cnt, double *total_out, int lane,
                                               unsigned long long &newb_out, unsigned long long &keep_out, int &id_out, int &k_out,
                                               int &entry_out, int &n_flag_out, unsigned long long *tstamp = nullptr)
{
#define SEG_U_STAMP(i) do { if (tstamp && lane == 0) tstamp[i] = wall_clock64(); } while (0)
    N = __builtin_amdgcn_readfirstlane(N);             // wave-uniform by contract: keep the loops below on the scalar unit
    W = __builtin_amdgcn_readfirstlane(W);
    SEG_U_STAMP(0);
    // ---- A8 forward (kmeans_acoustic_wordseg.py:494-506)
    if (lane < 8) gamp[lane] = NEG_INF_D;
    if (lane == 0) gamp[8] = 0.0;
    if (W <= 2) seg_forward_uniform<2>(bvec8, gamp, N, lane);
    else if (W <= 4) seg_forward_uniform<4>(bvec8, gamp, N, lane);
    else if (W <= 6) seg_forward_uniform<6>(bvec8, gamp, N, lane);
    else seg_forward_uniform<8>(bvec8, gamp, N, lane);
    WAVE_SYNC();
    SEG_U_STAMP(1);
    // ---- the decision of every span end at once: lane t - 1 holds kbv = length of the best span ending at t (first maximum in
    // w order: the reference's reversed np.argmax takes the shortest span on ties; 0: every candidate is -inf) and its entry
    int kbv = 0;
    double cv = 0.0;
    if (lane < N) {
        const double mx = gamp[8 + lane + 1];
        double v[8], g[8];
        const double2 *src = reinterpret_cast<const double2 *>(bvec8 + (size_t)lane * 8);
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const double2 d = src[h];
            v[2 * h] = d.x;
            v[2 * h + 1] = d.y;
        }
#pragma unroll
        for (int w = 0; w < 8; w++) g[w] = gamp[8 + lane - w];
        if (mx != NEG_INF_D) {
#pragma unroll
            for (int w = 7; w >= 0; w--)
                if (v[w] + g[w] == mx) {
                    kbv = w + 1;
                    cv = v[w];
                }
        }
    }
    SEG_U_STAMP(2);
    // ---- A8 backward (:510-553): a walk over the lanes' decisions
    unsigned long long newb = 1ull << (N - 1);
    int t = N;
    double total = 0.0;
    bool blocked = false;
    for (;;) {                                         // the walk as long as every span end on it has a finite candidate
        const int kb = __builtin_amdgcn_readlane(kbv, t - 1);
        if (kb == 0) {
            blocked = true;
            break;
        }
        total += seg_readlane_f64(cv, t - 1);
        t = t - kb;
        if (t < 1) break;
        newb |= 1ull << (t - 1);
    }
    if (blocked)
        for (;;) {                                     // ... and from the first span end whose candidates are all -inf on (:516-530)
            int kb = __builtin_amdgcn_readlane(kbv, t - 1);
            if (kb == 0) {
                do {
                    t = t - 1;
                    if (t == 0) break;
                    kb = __builtin_amdgcn_readlane(kbv, t - 1);
                } while (kb == 0);
                if (t == 0) {                          // python vec[-1]: the last span [N-1, N)
                    newb |= 1ull << (N - 1);
                    total += bvec8[(size_t)(N - 1) * 8];
                    break;
                }
                newb |= 1ull << (t - 1);
            }
            total += seg_readlane_f64(cv, t - 1);
            t = t - kb;
            if (t < 1) break;
            newb |= 1ull << (t - 1);
        }
    SEG_U_STAMP(3);
    // ---- new tokens + their best components (:312-313)
    {
        const bool bit = lane < N && ((newb >> lane) & 1ull);
        const unsigned long long below = newb & ((1ull << lane) - 1ull);
        const int jp = below ? 64 - __clzll((long long)below) : 0;
        const int w = lane - jp;
        int id = -1, kk = -1;
        if (bit && w < W) {
            id = bid[lane * W + w];
            kk = bk[lane * W + w];
        }
        const bool valid = bit && id >= 0;
        const unsigned long long keep = __ballot(valid), badm = __ballot(bit && !valid), fl = __ballot(valid && kk >= Kact);
        if (valid) {
            const int r = __popcll(keep & ((1ull << lane) - 1ull));
            l_new[r] = id;
            l_newk[r] = kk;
        }
        if (lane == 0) {
            l_cnt[1] = __popcll(keep);
            l_cnt[2] = (int32_t)(newb & 0xffffffffull);
            l_cnt[3] = (int32_t)(newb >> 32);
            l_cnt[4] = __popcll(fl);
            l_cnt[5] = badm != 0ull;
        }
        // the same for the caller's wave, in registers: this lane's new token (row, component, band entry), if any
        newb_out = newb;
        keep_out = keep;
        id_out = valid ? id : -1;
        k_out = kk;
        entry_out = lane * W + (w < W ? w : 0);
        n_flag_out = __popcll(fl);
    }
    *total_out = total;
    SEG_U_STAMP(4);
#undef SEG_U_STAMP
}

