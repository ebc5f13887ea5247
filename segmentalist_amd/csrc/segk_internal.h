// segk_internal.h -- shared by the translation units of libsegk.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/segk.h"

#define SEGK_WS_ENTRIES (256 * 1024)      /* (row, split) partial candidates of a split-K score launch */
#define SEGK_PROF_SLOTS 256

struct segk_ctx {
    int device_id;
    int n_cu;
    char arch[64];
    // workspace of the split-K tail of the score stage: k [entries] int32, f [entries][2] float
    int32_t *ws_k;
    float *ws_f;
    unsigned long long *ws_u64;   // split full scan: (score, component) per queue entry, zero between uses
    // rows the one-product pre-filter could not decide: [0] count, [16..] row ids (grown on demand)
    int32_t *pre_queue;
    int64_t pre_cap;
    // split second stage of the pre-filter: partial candidates [row block][split][128 rows][4] and one ticket per row block
    float *sp2_part;
    int32_t *sp2_ticket;
    int64_t sp2_blocks;
    // persistent sequential chain (segk_seq_chain.hip): span maxima, control words, the sweep's utterance order
    void *chain_buf;
    size_t chain_bytes;
    double *fb_ktab;              // persistent FBGMM chain: fb_diag_const by count (diagonal components)
    int64_t fb_ktab_n;
    double fb_ktab_v0;
    int fb_ktab_D;
    double *fb_ptab;              // persistent FBGMM chain: log prior predictive of every row, and what it was computed from
    int64_t fb_ptab_n;
    unsigned long long fb_ptab_fp;     // fingerprint of the rows and the prior it was computed from (k_fb_fingerprint), and their shape
    int64_t fb_ptab_rows;
    int fb_ptab_D;
    unsigned long long *fb_fp_dev;     // [dev] the fingerprint's accumulator
    int fb_ptab_nt, fb_ptab_cov;
    double fb_ptab_k0, fb_ptab_v0;
    void *fbchain_buf;            // persistent FBGMM chain (segk_fbgmm.hip k_fb_chain): control words, the sweep's utterance order
    size_t fbchain_bytes;
    void *fbchain_lm;             // ... with a language model: the workgroups' copies of the bigram counts
    size_t fbchain_lm_bytes;
    void *fbchain_terms;          // ... the spans' predictive terms of one utterance, the exchange between the workgroups
    size_t fbchain_terms_bytes;
    // batch sampler: the block's tokens bucketed by slot (k_fbb_sort) + offsets
    int32_t *fbs_buf;
    size_t fbs_bytes;
    const void *fbb_scal_zeroed;  // segk_fbb_partials cleared these totals on the stream: the next segk_fbb_prepare need not
    const void *fbb_scal_stream;  // ... provided it is enqueued on the same stream
    int prof_launches;
    int32_t *defer_zero;          // segk_kmeans_score: queue length the chosen filter path still has to clear
    int pre_zeroed;
    int capturing;                // segk_graph_begin .. segk_graph_end
    // value hashes of the rows of the means most recently prepared (segk_kmeans_mark_duplicates)
    unsigned long long *row_hash;
    const void *row_hash_means;
    // rows bucketed by label (segk_rows_by_label): sorted row offsets [cap], per-block offsets, block bounds, small scratch
    int32_t *rb_sorted;
    int64_t rb_cap;
    int32_t *rb_koff;            // [64 * (rb_K + 1)]
    int rb_K;
    int32_t *rb_misc;            // blk_lo [68], dummy K, flags; then doubles (part_tot, scalars, terms)
    double *rb_term;             // [rb_K] per-component terms of the record metrics
    // hinted score path (segk_score_hint.hip): the filter's partial top-2 per (range, row), the hint map [K_max]
    void *hint_part;
    size_t hint_part_bytes;
    int32_t *hint_map;
    int hint_map_k;
    float *pre_thr;              // band stage (segk_score_band.hip): threshold per entry of pre_queue
    int64_t pre_thr_cap;
    void *band_mask;             // its candidate masks, [ranges][queue capacity][2] x 16 bytes
    size_t band_mask_bytes;
    // hinted path feedback (segk_kmeans_hint_feedback): host-mapped word the band kernel writes, (call number << 32) | permille
    // of the call's rows the certificate could not decide; no stream command, no synchronisation
    volatile unsigned long long *miss_host;
    unsigned long long *miss_dev;
    unsigned int miss_seq;
    void *hint_fb;               // per-XCD shares and wave lifetimes of the matrix kernel's last launches, [3][8] floats + [3][8] uint32
    unsigned int hint_fb_launch;
    // full scan with the components in LDS (k_kmeans_brute_ls): (score, component) per queue entry, zero between uses
    unsigned long long *brute_ws;
    int64_t brute_ws_cap;
    // k-means batch finalize: the flagged tokens of a sweep beyond the kernel's LDS list, [5][flag_ovf_cap] int32
    int32_t *flag_ovf;
    int64_t flag_ovf_cap;
    // diagnostic probes of the batch sampler's tolerance modes (segk_fbb_set_probe); NULL = off
    double *probe_alpha, *probe_ll;
    int64_t probe_ll_ld;
    // optional timing of the main score launch (segk_profile_*): event pairs used round-robin
    int prof_on, prof_n, prof_kind;      // prof_on: 0 off, N: every Nth timed launch records its event pair
    unsigned int prof_calls;
    hipEvent_t prof_ev[SEGK_PROF_SLOTS][2];
    int64_t prof_rows[SEGK_PROF_SLOTS];
};

// does THIS launch record its event pair?  (segk_profile_enable(ctx, N): every Nth -- an event record between two kernels costs
// the stream ~3 us of bubble, 6.5 us per sweep around the headline kernel with N = 1)
static inline bool segk_prof_now(segk_ctx *ctx)
{
    if (!ctx->prof_on) return false;
    return (ctx->prof_calls++ % (unsigned int)ctx->prof_on) == 0u;
}


void segk_set_error(const char *fmt, ...);

// Development switches that ALTER RESULTS (timing ablations, debug modes) or hand raw pointers to kernels exist only in builds
// with -DSEGK_DEV: the shipped library cannot be told through the environment to compute wrong answers.  (The launch-plan
// switches that remain -- SEGK_SCORE_HINT, SEGK_SCORE_PRE, SEGK_SCORE_B3, ... -- choose between paths with identical results.)
#ifdef SEGK_DEV
#include <stdlib.h>
static inline int segk_dev_env(const char *name) { const char *e = getenv(name); return e ? atoi(e) : 0; }
#else
static inline int segk_dev_env(const char *) { return 0; }
#endif

// segk_metrics.hip: the rows 0..n-1 bucketed by label (labels[row] in [0, K_max), anything else: not listed), every
// bucket in ascending row order -- the stable counting sort of the k-means batch statistics run over blocks of rows.
// Row q of label k inside block b: blk_lo[b] + sorted[blk_lo[b] + koff[b * (K_max + 1) + k] + q].  Buffers are owned
// by the context (valid until the next call).  K_max <= 8192.
int segk_rows_by_label(segk_ctx *ctx, const int32_t *labels, int64_t n, int K_max, const int32_t **blk_lo, int *n_blocks,
                       const int32_t **sorted, const int32_t **koff, void *stream);

#define SEGK_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            segk_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SEGK_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define SEGK_REQUIRE(cond, msg)                                             \
    do {                                                                    \
        if (!(cond)) {                                                      \
            segk_set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond); \
            return SEGK_ERR_ARG;                                            \
        }                                                                   \
    } while (0)

#define SEGK_LAUNCH_CHECK() SEGK_CHECK_HIP(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize for `fn` on the CURRENT device, raised when `lds` exceeds 48 KB and what has been set
// there before.  The attribute is per device: a function-local static flag (rounds 1-2) skipped it for a second context on
// another GPU of the same process.  segk_occupancy: hipOccupancyMaxActiveBlocksPerMultiprocessor, cached per (device, function,
// threads, lds) the same way.
hipError_t segk_dyn_lds(const void *fn, size_t lds);
hipError_t segk_occupancy(const void *fn, int threads, size_t lds, int *wg_per_cu);

// ---------------------------------------------------------------------------------------
// Layout of the MFMA operand image of the k-means means ("tiles"), shared by the prepare
// and score kernels.  One tile = 32 components:
//   floats [g][lane][2], g < G = ceil(D/4), lane < 64:
//        M[32*tile + (lane & 31)][4*g + 2*(lane >> 5) + {0,1}]
//   (g runs to the bucket GB = segk_gmax(D) >= G, zero filled beyond D)
//   followed, at float offset GB*128, by 32 floats c[i] = -|m_i|^2 / 2   (-3e38 for rows >= K_max)
//   padded with zeros to a multiple of 1024 floats (whole 16-byte x 256-thread passes).
// ---------------------------------------------------------------------------------------
static inline __host__ __device__ int segk_G(int D) { return (D + 3) / 4; }
// k-extent bucket (in groups of 4 dims) of the score-kernel instantiation that serves
// dimension D: exact for the common embedding sizes, next larger bucket otherwise (the
// operands are zero padded to the bucket, so the surplus MFMAs add exact zeros).
#define SEGK_G_BUCKETS {1, 2, 4, 6, 8, 10, 13, 16, 20, 25, 26, 28, 32, 33, 34, 40, 50, 64, 75, 100}
static inline __host__ __device__ int segk_gmax(int D)
{
    const int b[] = SEGK_G_BUCKETS;
    int G = segk_G(D);
    for (unsigned i = 0; i < sizeof(b) / sizeof(b[0]); i++)
        if (G <= b[i]) return b[i];
    return -1;
}
// tile stride in floats: the bucket's image ([GB][64][2] operand floats + 32 constants)
// rounded up to whole 1024-float passes
static inline __host__ __device__ int segk_tile_stride(int D)
{
    int gm = segk_gmax(D);
    if (gm < 0) gm = segk_G(D);
    int f = gm * 128 + 32;
    return (f + 1023) / 1024 * 1024;
}
static inline __host__ __device__ int segk_n_tiles(int K_max) { return (K_max + 31) / 32; }

// ---------------------------------------------------------------------------------------
// Operand images of the split-precision k-means filter (float32 data, 8 <= D <= 128; segk_score_sp.hip, segk_score_h1.hip).
// P = 3: three bf16 pieces, x = x1 + x2 + x3 exactly.  P = 2: two fp16 pieces of 2^a x (power-of-two
// scaling), the second one carried at 2^11 times its weight.
//   rows   [SEGK_SP_HEADER bytes: int32 {P, exponent a, bits of max |x_d|, 0}, int64 n_emb * KP] then P planes
//          [n_emb][KP] of 16-bit pieces (piece p of every row together), KP = D rounded up to 16 (zero padded,
//          dimensions permuted by segk_b3_dim); for P = 2 the float [n_emb] residual norms follow the planes
//   tiles  [1024 floats header: int32 exponent b at [0]; float E_m at [1]; [2], [3]: the batch finalize's residual
//          maximum and the exponent it built its rows with (k_batch_post reads them)] then per tile of 32 components:
//          16-bit [s][p][lane 64][8], s < KS = KP/16 (k-step), p < P (piece): piece p of
//          2^b M[32*tile + (lane & 31)][segk_b3_dim(16 s + 8 (lane >> 5) + i)], i < 8 -- the A operand of
//          v_mfma_f32_32x32x16_{f16,bf16} as one 16-byte load per lane; followed, at float offset KS*P*256,
//          by 32 floats -2^(a+b) |m|^2/2 (-3e38 beyond K_max); padded to a multiple of 1024 floats.
// ---------------------------------------------------------------------------------------
#define SEGK_SP_HEADER 64
// Slot -> dimension inside a 16-wide k-step: slot q = 8h + i (h = lane half, i < 8) carries dimension
// 8 (i >> 2) + 4 h + (i & 3): each lane half owns the dimensions d with (d mod 8) in {4h .. 4h+3} of
// both 8-blocks of the step -- four complete strided accumulators of numpy's pairwise sum, so the
// winner's exact score can be finished in the score kernel without exchanging terms (the contraction
// itself is indifferent to the order of the dimensions).
static inline __host__ __device__ int segk_b3_dim(int pos)       // position in a piece row -> dimension
{
    const int q = pos & 15, h = q >> 3, i = q & 7;
    return (pos & ~15) + 8 * (i >> 2) + 4 * h + (i & 3);
}
static inline __host__ __device__ int segk_b3_kp(int D) { return (D + 15) & ~15; }
static inline __host__ __device__ int segk_sp_tile_stride(int D, int P)
{
    return ((segk_b3_kp(D) / 16) * P * 256 + 32 + 1023) / 1024 * 1024;
}

// segk_score_f32.hip: the MFMA score kernel in log-sum-exp mode (used by segk_fbbatch.hip); not ABI
int segk_launch_score_lse(segk_ctx *ctx, const float *Y, int64_t ldy, int D2, const int32_t *ids, int64_t row0, int64_t n,
                          const float *tiles, int n_tiles, double norm, double *out, void *stream);

// segk_prepare.hip / segk_score_sp.hip: fp16x2 images of arbitrary float32 matrices and the log-sum-exp kernel on them
int segk_sp_prepare_rows(const float *Y, int64_t ldy, int64_t n, int D2, void *img, void *stream);
int segk_sp_prepare_tiles(const float *rows, const double *consts, const double *rowmax2, int K, int D2, float *tiles_sp,
                          const void *ximg, void *stream);
int segk_launch_score_lse_sp(segk_ctx *ctx, const void *ximg, int D2, const int32_t *ids, int64_t row0, int64_t n,
                             const float *tiles_sp, int n_tiles, double norm, double *out, void *stream,
                             const int32_t *n_tiles_dev = nullptr);
int segk_launch_score_mat_sp(const void *ximg, int D2, const int32_t *ids, int64_t n, const float *tiles_sp, int n_tiles,
                             float *mat, int64_t mat_ld, void *stream, const int32_t *n_tiles_dev = nullptr, int tiles_hint = 0);
