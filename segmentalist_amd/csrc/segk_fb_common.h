// Device helpers shared by the FBGMM kernels (segk_fbgmm.hip: the reference's serial chain;
// segk_fbbatch.hip: the batch-synchronous sampler).  Everything is `static`: each translation
// unit gets its own copy (no relocatable device code).
#pragma once
#include "segk_internal.h"

#define NEG_INF_D (-__builtin_huge_val())
#define LOG_2PI 1.8378770664093453
#define LOG_PI 1.1447298858494002

static __device__ __forceinline__ double fb_readlane(double v, int l)       // l wave-uniform
{
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return u.d;
}

// _cython_utils.pyx:13-25 (max, then the sum of exp(a[j] - max) in index order, then log) by one
// full wave: the exponentials are evaluated one per lane, the additions stay sequential.
// Cross-lane moves inside a row of sixteen lanes (DPP: a few clocks; __shfl_xor on a double is two ds_bpermute round trips, and
// a lone wave -- the DP is one wave per utterance -- waits out every one of them)
template <int CTRL>
static __device__ __forceinline__ double fb_dpp_f64(double v)
{
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(u.i[0], u.i[0], CTRL, 0xF, 0xF, false);
    r.i[1] = __builtin_amdgcn_update_dpp(u.i[1], u.i[1], CTRL, 0xF, 0xF, false);
    return r.d;
}
// maximum over the lanes 0..15 of a row, in all of them (quad xor 1, xor 2, half-row mirror, row mirror)
static __device__ __forceinline__ double fb_row16_max(double v)
{
    double o = fb_dpp_f64<0xB1>(v);
    v = o > v ? o : v;
    o = fb_dpp_f64<0x4E>(v);
    v = o > v ? o : v;
    o = fb_dpp_f64<0x141>(v);
    v = o > v ? o : v;
    o = fb_dpp_f64<0x140>(v);
    v = o > v ? o : v;
    return v;
}

// maximum over all 64 lanes, wave-uniform: the rows' maxima by DPP, the four rows' by v_readlane (one row when `one_row`)
static __device__ __forceinline__ double fb_wave_max(double v, bool one_row)
{
    v = fb_row16_max(v);
    double m = fb_readlane(v, 0);
    if (!one_row) {
        const double m1 = fb_readlane(v, 16), m2 = fb_readlane(v, 32), m3 = fb_readlane(v, 48);
        m = m1 > m ? m1 : m;
        const double m23 = m3 > m2 ? m3 : m2;
        m = m23 > m ? m23 : m;
    }
    return m;
}

// sum over all 64 lanes, in every lane, in a fixed order: inside the rows of sixteen by DPP (quad xor 1, xor 2, half-row mirror,
// row mirror: at every step the partners hold the totals of disjoint groups), the four rows' totals by v_readlane as
// (r0 + r1) + (r2 + r3).  (__shfl_xor on a double is two ds_bpermute round trips per step: six steps were ~0.5 us of every
// block-wide reduction, and the serial chains do a dozen of those per token.)
static __device__ __forceinline__ double fb_wave_sum(double v)
{
    v += fb_dpp_f64<0xB1>(v);
    v += fb_dpp_f64<0x4E>(v);
    v += fb_dpp_f64<0x141>(v);
    v += fb_dpp_f64<0x140>(v);
    const double r0 = fb_readlane(v, 0), r1 = fb_readlane(v, 16), r2 = fb_readlane(v, 32), r3 = fb_readlane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// float32 maximum / sum over the 64 lanes, wave-uniform: rows of sixteen by DPP, the four rows by v_readlane (a __shfl_xor
// butterfly is six ds_bpermute round trips: 96 of them per thread were 5 us of k_fbb_score_diag32's 30)
template <int CTRL>
static __device__ __forceinline__ float fb_dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}
static __device__ __forceinline__ float fb_wave_max_f32(float v)
{
    v = fmaxf(v, fb_dpp_f32<0xB1>(v));
    v = fmaxf(v, fb_dpp_f32<0x4E>(v));
    v = fmaxf(v, fb_dpp_f32<0x141>(v));
    v = fmaxf(v, fb_dpp_f32<0x140>(v));
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
static __device__ __forceinline__ float fb_wave_sum_f32(float v)
{
    v += fb_dpp_f32<0xB1>(v);
    v += fb_dpp_f32<0x4E>(v);
    v += fb_dpp_f32<0x141>(v);
    v += fb_dpp_f32<0x140>(v);
    const int i = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(i, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(i, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(i, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}

// inclusive prefix sum over the 64 lanes (lane l: v_0 + ... + v_l): inside the rows of sixteen by DPP row shifts (the lanes
// a shift leaves without a source add zero), then the totals of the rows before by v_readlane
template <int CTRL>
static __device__ __forceinline__ double fb_dpp_f64_z(double v)
{
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, u.i[0], CTRL, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, u.i[1], CTRL, 0xF, 0xF, true);
    return r.d;
}
static __device__ __forceinline__ double fb_wave_scan(double v)
{
    v += fb_dpp_f64_z<0x111>(v);          // row_shr:1
    v += fb_dpp_f64_z<0x112>(v);          // row_shr:2
    v += fb_dpp_f64_z<0x114>(v);          // row_shr:4
    v += fb_dpp_f64_z<0x118>(v);          // row_shr:8
    const double t0 = fb_readlane(v, 15), t1 = fb_readlane(v, 31), t2 = fb_readlane(v, 47);
    const int row = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 4);
    const double add = row == 0 ? 0.0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : (t0 + t1) + t2));
    return v + add;
}

// ---------------------------------------------------------------------------------------
// block-wide helpers (blockDim.x a multiple of 64, <= 1024; `red` has >= 16 doubles)
// ---------------------------------------------------------------------------------------
// The wave's total by DPP + v_readlane (fb_wave_sum), then the per-wave partials (<= 16) are added in wave order by every
// thread: two barriers per reduction and a fixed, launch-independent order.
static __device__ double block_sum(double v, double *red)
{
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    v = fb_wave_sum(v);
    __syncthreads();                      // `red` may still be read from a previous reduction
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; w++) r += red[w];
    return r;
}

static __device__ double block_max(double v, double *red)
{
    const int tid = threadIdx.x, nw = blockDim.x >> 6;
    v = fb_wave_max(v, false);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; w++) r = red[w] > r ? red[w] : r;
    return r;
}

static __device__ double fb_logsumexp_wave(const double *a, int n, int lane)
{
    double mx = NEG_INF_D;
    for (int j = lane; j < n; j += 64) mx = a[j] > mx ? a[j] : mx;
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    double s = 0.0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const double ej = (j0 + lane < n) ? exp(a[j0 + lane] - mx) : 0.0;
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        for (int q = 0; q < cnt; q++) s += fb_readlane(ej, q);
    }
    return log(s) + mx;
}

// The same with the hardware exponential and logarithm (v_exp_f32 / v_log_f32) on differences formed in fp64: ~3e-6
// relative in the sum, for the batch sampler's tolerance modes (`score_precision` f32 / f16; segk_fbatch.fast_dp).
static __device__ __forceinline__ double fb_exp_fast(double d)      // exp(d), d <= ~0
{
    return (double)__builtin_amdgcn_exp2f((float)d * 1.4426950408889634f);
}
static __device__ __forceinline__ double fb_log_fast(double x)      // log(x), x of order 1
{
    return (double)(__builtin_amdgcn_logf((float)x) * 0.6931471805599453f);
}
static __device__ double fb_logsumexp_wave_fast(const double *a, int n, int lane)
{
    double mx = NEG_INF_D;
    for (int j = lane; j < n; j += 64) mx = a[j] > mx ? a[j] : mx;
    for (int o = 32; o > 0; o >>= 1) {
        double other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    double s = 0.0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const double ej = (j0 + lane < n) ? fb_exp_fast(a[j0 + lane] - mx) : 0.0;
        const int cnt = n - j0 < 64 ? n - j0 : 64;
        for (int q = 0; q < cnt; q++) s += fb_readlane(ej, q);
    }
    return fb_log_fast(s) + mx;
}

// embedding ids of the segments the boundaries of one utterance select, -1 (no embedding) skipped
// (unigram_acoustic_wordseg.py:340-342)
static __device__ int fb_collect_tokens(const int32_t *vid, const uint8_t *bnd, int N, int32_t *tok)
{
    int nn = 0, jp = 0;
    for (int j = 0; j < N; j++)
        if (bnd[j]) {
            int id = vid[(j + 1) * j / 2 + jp];
            if (id >= 0) tok[nn++] = id;
            jp = j + 1;
        }
    return nn;
}

// The same by a whole wave (N <= 64): one load of the boundary flags, the lane of every set flag looks up its own segment
// [jp, j + 1), the kept ids are compacted in order by a prefix count.  All 64 lanes must call it; returns the count in every
// lane.  (The one-lane loop above is N dependent global loads: 5-10 us per call in the boundary kernels.)
static __device__ int fb_collect_tokens_wave(const int32_t *vid, const uint8_t *bnd, int N, int32_t *tok, int lane,
                                             int32_t *tokj = nullptr)         // tokj: the kept segments' triangular indices
{
    // the flags may have been written a moment ago by other lanes of this wave: wait for those stores and read past the L1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint8_t flag = lane < N ? __hip_atomic_load(&bnd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint8_t)0;
    const unsigned long long mask = __ballot(flag != 0);
    const bool bit = lane < N && ((mask >> lane) & 1ull);
    const unsigned long long below = mask & ((1ull << lane) - 1ull);
    const int jp = below ? 64 - __clzll((long long)below) : 0;
    int id = -1;
    if (bit) id = vid[(lane + 1) * lane / 2 + jp];
    const unsigned long long keep = __ballot(bit && id >= 0);
    if (bit && id >= 0) {
        tok[__popcll(keep & ((1ull << lane) - 1ull))] = id;
        if (tokj) tokj[__popcll(keep & ((1ull << lane) - 1ull))] = (lane + 1) * lane / 2 + jp;
    }
    return __popcll(keep);
}


// utils.draw (utils.py:10-21): the first q with u - p[0] - ... - p[q] < 0, subtractions in index
// order (else n - 1).  Executed redundantly by every calling lane (uniform LDS addresses are
// broadcast reads): sixteen probabilities are fetched per step, then sixteen dependent
// subtractions, then the exit checks.
static __device__ int fb_draw_seq(const double *p, int n, double u)
{
    double uu = u;
    for (int q0 = 0; q0 < n; q0 += 16) {
        double v[16], r[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int q = q0 + j < n ? q0 + j : n - 1;      // clamped, unconditional load
            v[j] = p[q];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uu = uu - (q0 + j < n ? v[j] : 0.0);
            r[j] = uu;
        }
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (q0 + j < n && r[j] < 0) return q0 + j;
    }
    return n - 1;
}

// Inverse-CDF draw of the batch sampler (oracle/np_fbgmm_batch.py `draw_chunked`): the n
// probabilities are cut into 64 runs of per = ceil(n/64) consecutive entries; run sums are
// accumulated left to right, the runs are walked in order subtracting whole run sums from u, and
// the run where the remainder would turn negative is walked entry by entry.  One full wave;
// result in all lanes.  Same distribution as utils.draw, O(n/64 + 64) dependent steps.
static __device__ int fb_draw_chunked(const double *p, int n, double u, int lane)
{
    const int per = (n + 63) >> 6;
    const int lo = lane * per < n ? lane * per : n, hi = lo + per < n ? lo + per : n;
    double s = 0.0;
    for (int q = lo; q < hi; q++) s += p[q];
    double r = u;
    int run = -1;
    for (int l = 0; l < 64; l++) {
        const double sl = fb_readlane(s, l);
        if (r - sl < 0) { run = l; break; }
        r = r - sl;
    }
    if (run < 0) return n - 1;
    const int rlo = run * per < n ? run * per : n, rhi = rlo + per < n ? rlo + per : n;
    for (int q = rlo; q < rhi; q++) {
        r = r - p[q];
        if (r < 0) return q;
    }
    return rhi - 1 >= 0 ? rhi - 1 : 0;
}

// Counter-based uniform in [0, 1) of the batch sampler: u01(seed, sweep, utterance, j) -- two rounds
// of the splitmix64 finaliser over a linear combination of the counters (oracle/np_fbgmm_batch.py).
static __device__ __forceinline__ double segk_u01(uint64_t seed, uint64_t sweep, uint64_t utt, uint64_t j)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + sweep * 0xBF58476D1CE4E5B9ull + utt * 0x94D049BB133111EBull
                 + j * 0xD6E8FEB86659FD93ull + 0x2545F4914F6CDD1Dull;
    for (int r = 0; r < 2; r++) {
        z ^= z >> 30;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        z *= 0x94D049BB133111EBull;
        z ^= z >> 31;
    }
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// where the backward-sampling steps take their uniforms from
struct StreamUniforms {        // the pre-drawn `random.random()` values of the serial chain
    const double *us;
    int64_t cur, cap;
    int32_t *status;
    int64_t base = 0;          // us[0] is value `base` of the stream (a window of it staged in LDS)
    __device__ double next(int lane)
    {
        const double u = (cur < cap) ? us[cur - base] : 0.5;
        if (cur >= cap && lane == 0) atomicOr(status, 8);
        cur++;
        return u;
    }
};

struct CounterUniforms {       // the batch sampler's counter-based stream
    uint64_t seed, sweep, utt, j;
    __device__ double next(int) { return segk_u01(seed, sweep, utt, j++); }
};

// Span tables of one utterance for a DP window of n_max slices: the banded image (segk_corpus.band_ids / band_dur: entry
// (t, w), t = 1..N the span's end, w its length minus one, at [(t - 1) W + w]) when it was built for this window, else the
// triangle (utterances.py:91-105).  The band is what the DP reads (spans of at most n_max slices): 120 entries instead of
// 210 at N = 20, n_slices_max = 6, and entry i is read by thread i.  These kernels take the band as COMPLETE -- every
// triangular entry outside it is -1 (no embedding); the host side checks that before it hands the band over (segk.h).
struct FbSpanTab {
    const int32_t *vid;      // triangle
    const double *dur;
    const int32_t *bandi;    // band, when `band`
    const double *bandd;
    int W;
    bool band;
};
static __device__ __forceinline__ FbSpanTab fb_span_tab(const segk_corpus &c, int u, int N, int n_max)
{
    const int64_t triMax = (int64_t)c.N_max * (c.N_max + 1) / 2;
    FbSpanTab T;
    T.vid = c.vec_ids + (int64_t)u * triMax;
    T.dur = c.durations + (int64_t)u * triMax;
    T.W = (n_max > 0 && n_max < N) ? n_max : N;
    T.band = c.band_ids != nullptr && c.band_dur != nullptr && c.band_W == T.W && T.W > 0 && T.W < N;
    T.bandi = T.band ? c.band_ids + (int64_t)u * c.N_max * c.band_W : nullptr;
    T.bandd = T.band ? c.band_dur + (int64_t)u * c.N_max * c.band_W : nullptr;
    return T;
}

// vec (unigram_acoustic_wordseg.py:474-511), triangular in LDS: score * duration ** time_power_term + wip per span, -inf where
// there is none.  `score_of(id)`: the span score of embedding id.  Every thread of the workgroup calls (a barrier inside).
template <typename SC>
static __device__ void fb_fill_vec(const FbSpanTab &T, int N, int tri, SC score_of, double time_power_term, double wip, double *vec,
                                   int tid, int nt)
{
    // (x ** 1.0 is x -- numpy's power, the specification, returns it exactly; the software pow costs ~300 instructions per span)
    if (T.band) {
        for (int j = tid; j < tri; j += nt) vec[j] = NEG_INF_D;            // (-inf + wip)
        __syncthreads();
        const int W = T.W;
        for (int i = tid; i < N * W; i += nt) {
            const int t = i / W + 1, s = t - 1 - (i - (t - 1) * W);
            if (s < 0) continue;
            const int id = T.bandi[i];
            if (id < 0) continue;
            const double dd = T.bandd[i];
            const double v = isnan(dd) ? NEG_INF_D : score_of(id) * (time_power_term == 1.0 ? dd : pow(dd, time_power_term));
            vec[t * (t - 1) / 2 + s] = v + wip;
        }
    } else {
        for (int j = tid; j < tri; j += nt) {
            const int id = T.vid[j];
            double v = NEG_INF_D;
            if (id >= 0) {
                const double dd = T.dur[j];
                v = isnan(dd) ? NEG_INF_D : score_of(id) * (time_power_term == 1.0 ? dd : pow(dd, time_power_term));
            }
            vec[j] = v + wip;
        }
    }
}

// A6 / A7 by one full wave (unigram_acoustic_wordseg.py:653-864): forward filtering, then backward
// sampling (or Viterbi back-tracking) writing the boundaries; returns the summed score of the chosen
// segments.  Control flow and values are wave-uniform; the exponentials of each logsumexp /
// normalisation are spread over the lanes, sums and draws keep the reference order.
// The sampling DP of the tolerance modes (segk_fbatch.fast_dp; windows of at most sixteen slices, at most 64 landmarks) in
// FLOAT32: lane w holds candidate s = t - 1 - w and alpha[s] (a delay line shifted by one lane per step), maximum and sum over
// the window's lanes by DPP inside the row of sixteen, v_exp_f32 / v_log_f32 -- half the instructions of the fp64 form with
// hardware exponentials, whose lone wave was 20 us of a Gibbs step's critical path.  alpha is of the order of 1e3: float32
// carries it to ~1e-7 relative, the contract of these modes is 1e-4 (tests/test_gpu_tolerance_modes.py holds the values
// against the fp64 recurrence).  The chosen segments' scores are summed in fp64 as before; a [N] receives alpha (probe).
static __device__ __forceinline__ float fb_row16_max_f32(float v)
{
    v = fmaxf(v, fb_dpp_f32<0xB1>(v));
    v = fmaxf(v, fb_dpp_f32<0x4E>(v));
    v = fmaxf(v, fb_dpp_f32<0x141>(v));
    v = fmaxf(v, fb_dpp_f32<0x140>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}
static __device__ __forceinline__ float fb_row16_sum_f32(float v)
{
    v += fb_dpp_f32<0xB1>(v);
    v += fb_dpp_f32<0x4E>(v);
    v += fb_dpp_f32<0x141>(v);
    v += fb_dpp_f32<0x140>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}
template <typename USRC>
static __device__ double fb_dp_sample_fast32(const double *vec, double *a, int N, int tri, int n_max, float log_p_continue,
                                             double anneal_temp, uint8_t *bnd, int lane, USRC &usrc)
{
    const float NINF = -__builtin_huge_valf(), LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    for (int j = lane; j < N; j += 64) bnd[j] = (j == N - 1) ? 1 : 0;
    if (lane == 0) a[0] = 0.0;
    int i = 0;
    float g = 0.f;                                                   // alpha[t - 1 - lane]
    for (int t = 1; t < N; t++) {
        const int lo = t - n_max < 0 ? 0 : t - n_max, n = t - lo;
        const float v = lane < n ? (float)vec[i + t - 1 - lane] + g : NINF;
        const bool all_inf = __ballot(lane < n && v != NINF) == 0ull;
        float at = NINF;
        if (!all_inf) {
            const float m0 = fb_row16_max_f32(v);
            const float sm = fb_row16_sum_f32(lane < n ? __builtin_amdgcn_exp2f((v - m0) * LOG2E) : 0.f);
            at = __builtin_amdgcn_logf(sm) * LN2 + m0 + log_p_continue;
        }
        const float gs = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g), 0x138, 0xF, 0xF, false));      // wave_shr:1
        g = lane == 0 ? at : gs;
        if (lane == 0) a[t] = (double)at;
        i += t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int t = N;
    double total = 0.0;
    for (;;) {
        i = (t - 1) * t / 2;
        int lo = t - n_max < 0 ? 0 : t - n_max;
        // lane w: candidate s = t - 1 - w (the shortest segment first: the order the draw walks)
        float xw = lane < t - lo ? (float)vec[i + t - 1 - lane] + (float)a[t - 1 - lane] : NINF;
        bool all_inf = __ballot(lane < t - lo && xw != NINF) == 0ull;
        if (all_inf) {                                               // unigram_acoustic_wordseg.py:815-825: step back to a landmark that can end a segment
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (t - 1) * t / 2;
                lo = t - n_max < 0 ? 0 : t - n_max;
                xw = lane < t - lo ? (float)vec[i + t - 1 - lane] + (float)a[t - 1 - lane] : NINF;
                all_inf = __ballot(lane < t - lo && xw != NINF) == 0ull;
            }
            if (lane == 0) bnd[(t - 1 + N) % N] = 1;
        }
        int k = 1;
        if (t > 0) {
            const int n = t - lo;
            const float m0 = fb_row16_max_f32(xw);
            float lse = __builtin_amdgcn_logf(fb_row16_sum_f32(lane < n ? __builtin_amdgcn_exp2f((xw - m0) * LOG2E) : 0.f)) * LN2 + m0;
            if (anneal_temp != 1.0) {
                const float inv = (float)(1. / anneal_temp);
                xw = lane < n ? inv * (xw - lse) : NINF;
                const float m1 = fb_row16_max_f32(xw);
                lse = __builtin_amdgcn_logf(fb_row16_sum_f32(lane < n ? __builtin_amdgcn_exp2f((xw - m1) * LOG2E) : 0.f)) * LN2 + m1;
            }
            const float pl = lane < n ? __builtin_amdgcn_exp2f((xw - lse) * LOG2E) : 0.f;
            double uu = usrc.next(lane);
            int kk = n - 1;
            for (int j = 0; j < n; j++) {
                uu = uu - (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pl), j));
                if (uu < 0) { kk = j; break; }
            }
            k = kk + 1;
        }
        int idx = i + t - k;
        if (idx < 0) idx += tri;
        total += vec[idx];
        if (t - k - 1 < 0) break;
        if (lane == 0) bnd[t - k - 1] = 1;
        t = t - k;
    }
    return total;
}

//   vec [tri] scores (LDS), a [N], w [N+1], pr [N+1] scratch (LDS), bnd [N] boundaries (global)
template <typename USRC>
static __device__ double fb_dp_sample(const double *vec, double *a, double *w, double *pr, int N, int tri, int n_max,
                                      int viterbi, double log_p_continue, double anneal_temp, uint8_t *bnd, int lane,
                                      USRC &usrc, int fast = 0)
{
    if (fast && !viterbi && n_max > 0 && n_max <= 16 && N <= 64)
        return fb_dp_sample_fast32(vec, a, N, tri, n_max, (float)log_p_continue, anneal_temp, bnd, lane, usrc);
    for (int j = lane; j < N; j += 64) { a[j] = 1.0; bnd[j] = (j == N - 1) ? 1 : 0; }
    __builtin_amdgcn_wave_barrier();
    a[0] = 0.0;
    __builtin_amdgcn_wave_barrier();
    int i = 0;
    // A window of at most 64 slices (n_slices_max, or the whole utterance when it is unbounded and N <= 64): lane w holds
    // candidate s = t - 1 - w and alpha[s] in a register -- a delay line shifted by one lane per step (DPP wave_shr:1), the
    // new alpha entering at lane 0 --, the maximum by DPP inside the rows (and v_readlane across them when the window is
    // wider than sixteen); the exponentials one per lane and their sum in the order of s, as below.  Same values: maxima do
    // not depend on the order, everything else is the same operation on the same operands.
    const bool lanes16 = (n_max > 0 && n_max <= 64) || (n_max == 0 && N <= 64);
    const bool one_row = n_max > 0 && n_max <= 16;
    double g = 0.0;                                                  // alpha[t - 1 - lane]; alpha[0] = 0
    for (int t = 1; t < N; t++) {
        int lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
        int n = t - lo;
        double at;
        if (lanes16) {
            const double v = lane < n ? vec[i + t - 1 - lane] + g : NEG_INF_D;
            const bool all_inf = __ballot(lane < n && v != NEG_INF_D) == 0ull;
            const double m0 = fb_wave_max(v, one_row);
            if (viterbi) at = m0;
            else if (all_inf) at = NEG_INF_D;
            else {
                const double ej = lane < n ? (fast ? fb_exp_fast(v - m0) : exp(v - m0)) : 0.0;
                double sm = 0.0;
                for (int q = n - 1; q >= 0; q--) sm += fb_readlane(ej, q);       // s = lo .. t - 1
                at = (fast ? fb_log_fast(sm) : log(sm)) + m0 + log_p_continue;
            }
            const double gs = fb_dpp_f64<0x138>(g);                  // wave_shr:1: lane w takes lane w - 1's
            g = lane == 0 ? at : gs;
        } else if (n <= 64) {
            // one candidate per lane: the same maximum, the same exponentials and the same left-to-right sum as
            // fb_logsumexp_wave, without the trip of the candidates through LDS (a serial loop of n dependent reads and
            // stores per step)
            const double v = lane < n ? vec[i + lo + lane] + a[lo + lane] : NEG_INF_D;
            const bool all_inf = __ballot(lane < n && v != NEG_INF_D) == 0ull;
            double mx = v;
            for (int o = 32; o > 0; o >>= 1) {
                const double other = __shfl_xor(mx, o);
                mx = other > mx ? other : mx;
            }
            if (viterbi) at = mx;
            else if (all_inf) at = NEG_INF_D;
            else {
                const double ej = lane < n ? (fast ? fb_exp_fast(v - mx) : exp(v - mx)) : 0.0;
                double sm = 0.0;
                for (int q = 0; q < n; q++) sm += fb_readlane(ej, q);
                at = (fast ? fb_log_fast(sm) : log(sm)) + mx + log_p_continue;
            }
        } else {
            bool all_inf = true;
            double best = NEG_INF_D;
            for (int s = lo; s < t; s++) {
                double v = vec[i + s] + a[s];
                if (lane == 0) w[s - lo] = v;
                if (v != NEG_INF_D) all_inf = false;
                if (v > best) best = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (viterbi) at = best;
            else at = all_inf ? NEG_INF_D : fb_logsumexp_wave(w, n, lane) + log_p_continue;
        }
        if (lane == 0) a[t] = at;
        __builtin_amdgcn_wave_barrier();
        i += t;
    }
    int t = N, lo = 0;
    double total = 0.0;
    for (;;) {
        i = (t - 1) * t / 2;
        lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
        bool all_inf = true;
        double xw = NEG_INF_D;                                       // (register path) candidate lo + lane of span end t
        if (lanes16) {
            xw = lane < t - lo ? vec[i + lo + lane] + a[lo + lane] : NEG_INF_D;
            all_inf = __ballot(lane < t - lo && xw != NEG_INF_D) == 0ull;
        } else
        for (int s = lo; s < t; s++)
            if (vec[i + s] + a[s] != NEG_INF_D) { all_inf = false; break; }
        if (all_inf) {
            while (all_inf) {
                t = t - 1;
                if (t == 0) break;
                i = (t - 1) * t / 2;
                lo = (n_max == 0 || t - n_max < 0) ? 0 : t - n_max;
                all_inf = true;
                if (lanes16) {
                    xw = lane < t - lo ? vec[i + lo + lane] + a[lo + lane] : NEG_INF_D;
                    all_inf = __ballot(lane < t - lo && xw != NEG_INF_D) == 0ull;
                } else
                for (int s = lo; s < t; s++)
                    if (vec[i + s] + a[s] != NEG_INF_D) { all_inf = false; break; }
            }
            if (lane == 0) bnd[(t - 1 + N) % N] = 1;
        }
        int k = 1, n = 1;
        if (lanes16 && !viterbi && anneal_temp == 1.0 && t > 0) {
            // the step in registers: lane j holds w[j]; maximum by DPP, the exponentials one per lane, their sum in index
            // order, the draw walking pr[j] = exp(w[n - 1 - j] - lse) -- the arithmetic of the general form below
            n = t - lo;
            const double m0 = fb_wave_max(xw, one_row);
            const double ej = lane < n ? (fast ? fb_exp_fast(xw - m0) : exp(xw - m0)) : 0.0;
            double sm = 0.0;
            for (int q = 0; q < n; q++) sm += fb_readlane(ej, q);
            const double lse = (fast ? fb_log_fast(sm) : log(sm)) + m0;
            const double pl = lane < n ? (fast ? fb_exp_fast(xw - lse) : exp(xw - lse)) : 0.0;
            double uu = usrc.next(lane);
            int kk = n - 1;
            for (int j = 0; j < n; j++) {
                uu = uu - fb_readlane(pl, n - 1 - j);
                if (uu < 0) { kk = j; break; }
            }
            k = kk + 1;
            int idx = i + t - k;
            if (idx < 0) idx += tri;
            total += vec[idx];
            if (t - k - 1 < 0) break;
            if (lane == 0) bnd[t - k - 1] = 1;
            t = t - k;
            continue;
        }
        if (t > 0) {
            n = t - lo;
            for (int j = lane; j < n; j += 64) w[j] = vec[i + lo + j] + a[lo + j];
        } else {
            if (lane == 0) w[0] = NEG_INF_D;
        }
        __builtin_amdgcn_wave_barrier();
        const double lse = fast ? fb_logsumexp_wave_fast(w, n, lane) : fb_logsumexp_wave(w, n, lane);
        if (viterbi) {
            if (t > 0) {
                for (int j = lane; j < n; j += 64) pr[j] = exp(w[j] - lse);
                __builtin_amdgcn_wave_barrier();
                double best = 0.0;
                bool first = true;
                for (int s = t - 1; s >= lo; s--) {
                    double q = pr[s - lo];
                    if (first || q > best) { best = q; k = t - s; first = false; }
                }
            }
        } else {
            if (anneal_temp != 1.0) {
                const double inv = 1. / anneal_temp;
                for (int j = lane; j < n; j += 64) pr[j] = w[n - 1 - j] - lse;
                __builtin_amdgcn_wave_barrier();
                for (int j = lane; j < n; j += 64) w[j] = inv * pr[j];
                __builtin_amdgcn_wave_barrier();
                const double lse2 = fast ? fb_logsumexp_wave_fast(w, n, lane) : fb_logsumexp_wave(w, n, lane);
                for (int j = lane; j < n; j += 64) pr[j] = fast ? fb_exp_fast(w[j] - lse2) : exp(w[j] - lse2);
            } else {
                for (int j = lane; j < n; j += 64) pr[j] = fast ? fb_exp_fast(w[n - 1 - j] - lse) : exp(w[n - 1 - j] - lse);
            }
            __builtin_amdgcn_wave_barrier();
            double uu = usrc.next(lane);
            int kk = n - 1;
            for (int j = 0; j < n; j++) {
                uu = uu - pr[j];
                if (uu < 0) { kk = j; break; }
            }
            k = kk + 1;
        }
        int idx = i + t - k;
        if (idx < 0) idx += tri;
        total += vec[idx];
        if (t - k - 1 < 0) break;
        if (lane == 0) bnd[t - k - 1] = 1;
        t = t - k;
    }
    return total;
}
