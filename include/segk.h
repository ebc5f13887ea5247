/*
 * segk.h -- C ABI of libsegk.so: the MI355X (gfx950) implementation of the segmentalist
 * per-utterance hot path (score every candidate acoustic-word embedding against every
 * mixture component -> segmentation DP over candidate boundaries -> (re)assignment ->
 * component sufficient statistics).
 *
 * The reference (kamperh/segmentalist, Python 2 + one Cython file) has no FFI for this
 * path other than `_cython_utils.pyx`; the boundary it sits behind is a set of duck-typed
 * Python methods.  Each entry point below names the reference
 * function(s) it replaces, file:line relative to /root/reference/segmentalist/.  The
 * Python package `segmentalist_amd` binds this header with ctypes (see INTEGRATION.md for
 * the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; segk_last_error() gives the text
 *     (thread local).  No exception crosses the ABI.
 *   - the CALLER owns every buffer.  Pointers marked [dev] are device (HBM) pointers,
 *     [host] are host pointers.  The library allocates nothing but the opaque segk_ctx.
 *   - `stream` is a hipStream_t passed as void*; all [dev] work is asynchronous on it.
 *   - there is NO CPU fallback: without a gfx950 device segk_create() fails.
 */
#ifndef SEGK_H
#define SEGK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGK_OK 0
#define SEGK_ERR_ARG (-1)
#define SEGK_ERR_HIP (-2)
#define SEGK_ERR_NO_DEVICE (-3)
#define SEGK_ERR_UNSUPPORTED (-4)

#define SEGK_F32 0
#define SEGK_F64 1

typedef struct segk_ctx segk_ctx;

int32_t segk_create(int32_t device_id, segk_ctx **out_ctx);
int32_t segk_destroy(segk_ctx *ctx);
const char *segk_last_error(void);
/* ABI version, bumped on any change of a signature or of a structure below (SEGK_ABI_VERSION is the version this
 * header describes; a binding must refuse a library that reports another one: segmentalist_amd/_abi.py does).
 *   1 rounds 1-2 | 2 round 2: segk_corpus gained band_W / band_ids / band_dur (the bump was forgotten then)
 *   3 round 3: segk_fbb_set_probe, segk_kmeans_score_hinted; this check
 *   4 round 3: scratch sizes of segk_kmeans_batch_partials from segk_kmeans_batch_scratch_words; segk_profile_enable(N)
 *   5 round 3: segk_fbatch.prior_rows, segk_fbb_prior_rows
 *   6 round 4: segk_kmeans_hint_feedback; flag_rows / flag_row_bytes of the batch statistics (sharded corpus);
 *              segk_fbgmm_sequential_sweep
 *   7 round 4: segk_fbatch.consts16 holds 2 (K_max + 2) + 32 doubles (the column maps of the packed operand image behind the
 *              constants) and the token-likelihood matrix has the image's columns; the FBGMM / bigram kernels read
 *              segk_corpus.band_ids / band_dur (a COMPLETE band only); segk_fbgmm_sequential_sweep with a language model
 *   8 round 4: segk_fbb_step_diag32                                                                                           */
#define SEGK_ABI_VERSION 8
int32_t segk_abi_version(void);

/* Timing of the MAIN launch of the MFMA score kernel (k_kmeans_score<..., 0>) with HIP events
 * recorded on its launch stream inside segk_kmeans_filter (and of its log-sum-exp twin inside
 * segk_fbb_score_f32) -- what bench.py's roofline.achieved is computed from.  segk_profile_read synchronises and returns, oldest first, the duration (ms) and
 * the row count of the most recent recorded launches (at most `max`, at most 256 kept).
 * on = 0: off; on = N >= 1: every Nth timed launch records its event pair (an event record between two kernels costs the
 * stream a few microseconds of bubble: bench.py samples every 8th sweep of its timed region).                            */
int32_t segk_profile_enable(segk_ctx *ctx, int32_t on);
int32_t segk_profile_read(segk_ctx *ctx, float *ms_out, int64_t *rows_out, int32_t max);
/* Which kernel the most recent recorded launch was: 0 the fp32-MFMA filter, 2 / 3 the split-precision
 * filter (fp16x2 / bf16x3), 1 the one-product fp16 pre-filter (k_kmeans_score_h1: rows above ~260 k,
 * D % 4 == 0), 4 the log-sum-exp kernels of the FBGMM batch sampler, 5 the range-stationary one-product
 * top-2 kernel of segk_kmeans_score_hinted (k_kmeans_top2_rs); -1 none recorded.                      */
int32_t segk_profile_last_kind(segk_ctx *ctx);
/* Number of back-to-back launches of that kernel the most recent recorded interval spans (1 since round 3 retired the
 * chunked pre-filter pipeline); the recorded duration and row count cover all of them.                            */
int32_t segk_profile_last_launches(segk_ctx *ctx);

/* hipGraph capture of a launch sequence (no reference counterpart: the reference has no device).  Every
 * kernel the library enqueues on `stream` between begin and end becomes one executable graph; segk_graph_launch replays it
 * with a single host call.  Run the sequence once before capturing it (workspaces, streams, events and
 * kernel attributes are created on first use, which a capture cannot contain); `stream` must not be the
 * legacy default stream; pointers and scalar arguments are frozen into the graph.                    */
int32_t segk_graph_begin(segk_ctx *ctx, void *stream);
int32_t segk_graph_end(segk_ctx *ctx, void *stream, void **exec_out);
int32_t segk_graph_launch(segk_ctx *ctx, void *exec, void *stream);
int32_t segk_graph_destroy(segk_ctx *ctx, void *exec);

/* -------------------------------------------------------------------------------------
 * Corpus (read-only during sampling): the device image of `Utterances` + the embedding
 * matrix (utterances.py:74-105; unigram_acoustic_wordseg.py:571-646).
 * ------------------------------------------------------------------------------------- */
typedef struct segk_corpus {
    const void *X;           /* [dev] embeddings, row-major [n_emb, ldx], dtype x_dtype      */
    const float *X32;        /* [dev] float32 image of X, row-major [n_emb, ld32] zero-padded;
                                ld32 = D rounded up to a multiple of 4.  (== X when x_dtype is
                                SEGK_F32 and ldx == ld32)                                     */
    int32_t x_dtype;         /* SEGK_F32 / SEGK_F64: dtype of X and of k-means `means`       */
    int32_t D;               /* embedding dimension                                           */
    int64_t n_emb;           /* rows of X                                                     */
    int64_t ldx;             /* leading dimension of X in elements                            */
    int64_t ld32;            /* leading dimension of X32 in floats                            */
    const float *xnorm;      /* [dev] [n_emb] upper bound of ||X[e]||_2 (segk_corpus_prepare)  */
    const int32_t *vec_ids;  /* [dev] [n_utt, tri]  span (s,t) at t(t-1)/2+s -> row of X, -1 */
    const double *durations; /* [dev] [n_utt, tri]  frames; NaN = span disallowed             */
    const int32_t *lengths;  /* [dev] [n_utt] landmarks per utterance                         */
    int32_t n_utt;
    int32_t N_max;           /* max landmarks; tri = N_max (N_max+1)/2                        */
    const void *Xb3;         /* [dev] optional (float32 data, 8 <= D <= 128): the rows split into 16-bit pieces
                                (segk_corpus_b3_bytes bytes, written by segk_corpus_prepare_b3); enables the
                                split-precision k-means filter on the 16-bit matrix pipe (NULL: fp32 MFMA)  */
    int32_t sp_pieces;       /* 2 = fp16x2, 3 = bf16x3: what Xb3 holds                        */
    int32_t band_W;          /* window of the banded span tables below (0: none)              */
    /* optional banded image of vec_ids / durations (utterances.py:91-105 keeps the triangular tables; the
     * per-utterance kernels only ever read the band t - s <= n_slices_max): entry (t, w), t = 1..N_max the span's
     * end, w = 0..band_W-1 its length minus one, at [(utt * N_max + t - 1) * band_W + w] = the triangular entry
     * t(t-1)/2 + (t-1-w) (-1 / NaN where that span does not exist).  Consecutive lanes read consecutive entries.
     * The k-means kernels fall back to the triangle for a span outside the band; the FBGMM / bigram kernels
     * (segk_fbgmm.hip, segk_fbbatch.hip) read the band alone when band_W equals the DP window and take it as
     * COMPLETE: pass it to them only when no triangular entry outside it names an embedding
     * (Utterances.complete_band_tables on the host side checks).                                    */
    const int32_t *band_ids; /* [dev] [n_utt, N_max, band_W] or NULL                          */
    const double *band_dur;  /* [dev] [n_utt, N_max, band_W] or NULL                          */
} segk_corpus;

/* Fill the derived members of a corpus: X32 (when X is float64 or ldx != ld32 the caller
 * passes a separate [n_emb, ld32] float buffer, written here) and xnorm. */
int32_t segk_corpus_prepare(segk_ctx *ctx, const segk_corpus *c, float *X32_out, float *xnorm_out,
                            void *stream);
/* Xb3_out [dev] segk_corpus_b3_bytes(n_emb, D) bytes (the size of three piece planes whatever `pieces` is).
 * pieces = 3: x = x1 + x2 + x3 exactly in bf16; pieces = 2: 2^a x = x1 + 2^-11 x2 (+ two dropped bits) in
 * fp16, a chosen from max |x|, followed -- in the room of the third plane -- by float [n_emb]
 * |x - x1| per row, which the one-product pre-filter's margin uses, and float [n_emb] -|x|^2 in the reference's
 * float32 summation order, which segk_kmeans_score_hinted uses.                                          */
int64_t segk_corpus_b3_bytes(int64_t n_emb, int32_t D);
int32_t segk_corpus_prepare_b3(segk_ctx *ctx, const segk_corpus *c, void *Xb3_out, int32_t pieces,
                               void *stream);

/* -------------------------------------------------------------------------------------
 * k-means components: device image of `KMeansComponents` (kmeans_components.py:18-91).
 * `means` has the dtype of X (kmeans_components.py:75-76: means = random_means.copy()).
 * ------------------------------------------------------------------------------------- */
typedef struct segk_kmeans {
    void *means;               /* [dev] [K_max, D] dtype x_dtype (inactive rows = random_means) */
    double *mean_numerators;   /* [dev] [K_max, D]                                              */
    int64_t *counts;           /* [dev] [K_max]                                                 */
    const void *random_means;  /* [dev] [K_max, D] dtype x_dtype                                */
    int32_t *assignments;      /* [dev] [n_emb]  component of each embedding, -1 = unassigned   */
    int32_t *K;                /* [dev] [1] number of active components                         */
    int32_t K_max;
    /* derived operands of the MFMA score kernel, maintained by the library: */
    float *tiles;              /* [dev] segk_kmeans_tiles_floats(K_max, D) floats               */
    double *mnorm_max;         /* [dev] [1] max_k ||means[k]||_2^2 (kept with atomicMax on the bits) */
    float *tiles_b3;           /* [dev] optional: segk_kmeans_tiles_b3_floats(K_max, D) floats, the bf16x3
                                  operand image of the means (maintained beside `tiles` when non-NULL) */
} segk_kmeans;

/* number of floats the caller must allocate for segk_kmeans.tiles / .tiles_b3 */
int64_t segk_kmeans_tiles_floats(int32_t K_max, int32_t D);
/* After segk_kmeans_prepare (optional; the batch sweeps call it): rows of `means` that are exact duplicates of a
 * row with a LOWER index are taken out of the filters' tile images (their accumulator seed becomes the "absent"
 * constant).  Such a row can never be np.argmax (kmeans_components.py:231: the first maximum wins, the scores are
 * bit-identical), but it turns every embedding near the pair into a tie for the full scan.  The full scan itself
 * does not read the constants.  n_marked [dev, optional]: += number of rows marked.  No-op for K_max > 2048.    */
int32_t segk_kmeans_mark_duplicates(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int32_t *n_marked,
                                    void *stream);
int64_t segk_kmeans_tiles_b3_floats(int32_t K_max, int32_t D);

/* KMeansComponents.__init__ (kmeans_components.py:59-81): from `assignments` (and
 * `random_means`) build counts, mean_numerators (sequential fp64 sums in ascending row order,
 * i.e. the order of the reference's add_item loop), means, K, tiles. */
int32_t segk_kmeans_init_stats(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream);

/* (Re)build `tiles` and `mnorm_max` from `means` (all K_max rows).  Called after any
 * wholesale change of the means. */
int32_t segk_kmeans_prepare(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream);

/* Candidate buffers of the A1 stage, indexed by embedding row (caller-owned, n_emb entries). */
typedef struct segk_cand {
    int32_t *k;      /* [dev] [n_emb]    argmax component                                       */
    float *f;        /* [dev] [n_emb, 2] the two largest values of the fp32 filter              */
    double *s;       /* [dev] [n_emb]    max score in REFERENCE arithmetic, widened to double   */
    int32_t *queue;  /* [dev] [n_emb]    rows the filter could not decide                       */
    int32_t *count;  /* [dev] [1]        length of `queue`                                      */
} segk_cand;

/* A1 -- KMeansComponents.neg_sqrd_norm / max_ / argmax_neg_sqrd_norm_i
 * kmeans_components.py:225-232, called per embedding from get_vec_embed_neg_len_sqrd_norms
 * kmeans_acoustic_wordseg.py:334-351, for rows `ids[0..n)` (rows row0..row0+n-1 when ids == NULL;
 * entries of ids equal to -1 are skipped).  Three kernels on the stream:
 *   (1) filter: fp32 MFMA contraction of the rows of X32 against all K_max means; per row the
 *       component with the largest f[k] = x.m_k - |m_k|^2/2 and the two largest values; for
 *       float32 data with 8 <= D <= 128 the winner's score in reference arithmetic is fused in
 *       the epilogue; rows whose two best filter values are closer than the proven error
 *       margin are queued;
 *   (2) (other dtypes / D) the winner's reference-arithmetic score per row;
 *   (3) full scan of the queued rows: the reference's own computation over all K_max
 *       components, first maximum.
 * Afterwards cand->k[e] / cand->s[e] are bit-identical to np.argmax / np.max of
 * neg_sqrd_norm(e) for every processed row e (bit-exact contract).
 * status [dev] int32 [8] or NULL: status[1] accumulates the number of fully scanned rows. */
int32_t segk_kmeans_score(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                          const int32_t *ids, int64_t row0, int64_t n, const segk_cand *cand,
                          int32_t *status, void *stream);

/* segk_kmeans_score with a HINT per row (no reference counterpart; the reference recomputes every argmax from nothing,
 * kmeans_components.py:225-232): on entry cand->k[e] of every row to be processed names the component the caller expects to
 * win -- normally what the previous call left there, the row's argmax under the previous statistics -- as a label of the
 * numbering `hint_remap` translates into the current one (hint_remap [dev] int32 [K_max], e.g. the relabel table
 * segk_kmeans_batch_finalize leaves in remap_scratch; NULL = identity; values outside [0, K_max) = no hint).  The library
 * still evaluates every (row, component) product on the matrix cores, but only VERIFIES the hint against them (the two
 * largest filter values and the hinted component's reference-arithmetic score; proof in segk_score_hint.hip) instead of
 * tracking which component won; rows whose hint cannot be verified are settled among the components inside the band of the
 * filter's maximum (segk_score_band.hip; tables of more than 2048 slots: the second stage of segk_kmeans_score), what is left
 * takes the full scan.  On return cand->k /
 * cand->s are bit-identical to segk_kmeans_score's whatever the hints were: a wrong hint costs time, never correctness.
 * A segk_ctx serves one stream at a time: the path's workspaces and the per-XCD shares of its matrix kernel are the context's.
 * Applies to float32 data with the fp16x2 row image (c->Xb3, sp_pieces 2), D % 4 == 0, launches of more than 384 rows per
 * CU; everything else is forwarded to segk_kmeans_score.                                                               */
int32_t segk_kmeans_score_hinted(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                 const int32_t *ids, int64_t row0, int64_t n, const segk_cand *cand,
                                 const int32_t *hint_remap, int32_t *status, void *stream);

/* The three steps of segk_kmeans_score as separate calls (same arguments), for callers that
 * want to time or overlap them: clear_queue (cand->count = 0), filter (kernel 1), resolve
 * (kernels 2 and 3). */
int32_t segk_kmeans_clear_queue(segk_ctx *ctx, const segk_cand *cand, void *stream);
int32_t segk_kmeans_filter(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                           const int32_t *ids, int64_t row0, int64_t n, const segk_cand *cand,
                           void *stream);
int32_t segk_kmeans_resolve(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                            const int32_t *ids, int64_t row0, int64_t n, const segk_cand *cand,
                            int32_t *status, void *stream);

/* Diagnostics of the most recent segk_kmeans_score on this context (no reference counterpart; the
 * full-size parity tests use it to prove that every stage of the path was exercised): out[0] = rows
 * the one-product pre-filter passed to its second stage (-1: the pre-filter has never run), out[1] =
 * rows in the ambiguity queue `cand->queue` (full reference-arithmetic scan).  out [host] int32 [2];
 * synchronises `stream`.                                                                          */
int32_t segk_kmeans_stage_counts(segk_ctx *ctx, const segk_cand *cand, int32_t *out, void *stream);

/* How well the hints of segk_kmeans_score_hinted are doing, WITHOUT touching any stream (no reference counterpart): hinted
 * calls on a context are numbered 1, 2, ...; *launched = number of the last one enqueued, *seen = number of the latest one
 * whose figure has reached the host (0: none yet; the device writes it into pinned host memory while the call runs),
 * *permille = thousandths of that call's rows the certificate could not decide (wrong or missing hints + near-ties).  A
 * driver that enqueues sweeps asynchronously uses it to leave the hints out (segk_kmeans_score) while most of them are
 * wrong -- the first sweeps of a chain --, where the hinted path would be the slower one; results do not depend on it.
 * launched, seen [host] uint32, permille [host] int32.                                                                  */
int32_t segk_kmeans_hint_feedback(segk_ctx *ctx, uint32_t *launched, uint32_t *seen, int32_t *permille);

/* Gather of the A1 results for rows ids[0..n) (0..n-1 when NULL): out_max[r] (double, widened
 * from the dtype of X) and out_arg[r] = np.max / np.argmax of neg_sqrd_norm(ids[r])
 * (kmeans_components.py:228-232), from a `cand` filled by segk_kmeans_score. */
int32_t segk_kmeans_exact_max(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                              const int32_t *ids, int64_t n, const segk_cand *cand,
                              double *out_max, int32_t *out_arg, void *stream);

/* A1 full vector: out[k], k < K_max, = neg_sqrd_norm(row) in reference arithmetic; `out`
 * has the dtype of X (kmeans_components.py:169-226).                                    */
int32_t segk_kmeans_neg_sqrd_norm(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                  int64_t row, void *out, void *stream);

/* A5 + A8 + new-segment argmax -- SegmentalKMeansWordseg.segment_i minus the statistics
 * update: get_vec_embed_neg_len_sqrd_norms (kmeans_acoustic_wordseg.py:334-351) from `cand`,
 * forward_backward_kmeans_viterbi (:449-555), get_max_unsup_transcript_i (:313,:437-446).
 * One wavefront per utterance utts[0..n_utts) (utterances utt0..utt0+n_utts-1 when utts == NULL).
 *   boundaries [dev] uint8 [n_utt, N_max]  in: current segmentation, out: new
 *   old_tok    [dev] int32 [n_utt, N_max]  embeddings of the OLD segmentation (-1 skipped)
 *   new_tok    [dev] int32 [n_utt, N_max]  embeddings of the NEW segmentation
 *   new_k      [dev] int32 [n_utt, N_max]  argmax component of each new segment
 *   n_old/n_new[dev] int32 [n_utt]
 *   n_flag     [dev] int32 [n_utt] or NULL: new segments whose argmax is an inactive row (k >= K)
 *   out_total  [dev] double [n_utt]        sum of chosen scores (:332)
 *   status     [dev] int32 [8]  [0] bit 1: a new segment has no embedding
 *              (kmeans_components.py:100 assert)                                        */
int32_t segk_kmeans_segment(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                            const int32_t *utts, int32_t utt0, int32_t n_utts,
                            int32_t n_slices_min, int32_t n_slices_max, double wip,
                            const segk_cand *cand, uint8_t *boundaries, int32_t *old_tok,
                            int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new,
                            int32_t *n_flag, double *out_total, int32_t *status, void *stream);

/* A11 sequential update -- the tail of segment_i (kmeans_acoustic_wordseg.py:314-320):
 * del_item(old) (kmeans_components.py:113-132), add_item(new, k) (:93-111, incl. the
 * `k > K -> K` clamp), clean_components (:263-266, del_component :149-166), in exactly the
 * reference's order and floating-point arithmetic, for ONE utterance; also refreshes the
 * changed rows of `tiles` / `mnorm_max`.  Device-side, single workgroup.                */
int32_t segk_kmeans_update_utt(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                               int32_t utt, const int32_t *old_tok, const int32_t *new_tok,
                               const int32_t *new_k, const int32_t *n_old, const int32_t *n_new,
                               int32_t *status, void *stream);

/* Single-item mutators with the reference's semantics (kmeans_components.py:93-166),
 * for the drop-in `KMeansComponents.add_item/del_item/del_component/clean_components`. */
int32_t segk_kmeans_add_item(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int64_t i,
                             int32_t k, int32_t *status, void *stream);
int32_t segk_kmeans_del_item(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int64_t i,
                             int32_t *status, void *stream);
int32_t segk_kmeans_clean_components(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                                     int32_t *status, void *stream);

/* del_component(k) (kmeans_components.py:149-166). */
int32_t segk_kmeans_del_component(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, int32_t k,
                                  int32_t *status, void *stream);

/* One sweep of the reference's sequential chain (SegmentalKMeansWordseg.segment's inner loop,
 * kmeans_acoustic_wordseg.py:393-399): for every utterance of `order` [HOST] int32 [n_order] in turn, segment_i
 * (:225-332) -- A1 for its spans directly in the reference's arithmetic (no filter, no operand images), the DP,
 * the del_item / add_item updates, clean_components; the operand images are refreshed once at the end.  Same
 * results as calling segk_kmeans_score / segk_kmeans_segment / segk_kmeans_update_utt per utterance (bit-identical
 * to the reference's chain).  Where the configuration allows (D % 4 = 0, 8 <= D <= 128, at most 32 landmarks per
 * utterance, 1 <= n_slices_max <= 8) the sweep runs as ONE persistent kernel (segk_seq_chain.hip: ~19 us per
 * utterance on the headline corpus); the call then synchronises `stream` after every launch of it (one launch per
 * stretch of utterances between two emptied components; not inside a graph capture).  Otherwise -- also when an
 * utterance occurs twice in `order`, which the persistent kernel's prefetch could not handle -- and with SEGK_SEQ_CHAIN=0: three launches
 * per utterance, all enqueued (~44 us per utterance).  keys_scratch [dev] uint64 [N_max (N_max + 1) / 2 + 2],
 * zeroed by the caller once.  float32 data; SEGK_ERR_UNSUPPORTED otherwise.                                   */
int32_t segk_kmeans_sequential_sweep(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                                     const int32_t *order, int32_t n_order, int32_t n_slices_min,
                                     int32_t n_slices_max, double wip, const segk_cand *cand,
                                     uint64_t *keys_scratch, uint8_t *boundaries, int32_t *old_tok,
                                     int32_t *new_tok, int32_t *new_k, int32_t *n_old, int32_t *n_new,
                                     int32_t *n_flag, double *out_total, int32_t *status, void *stream);

/* A11 batch-synchronous update (DESIGN.md "batch mode"; spec: oracle/np_oracle.py
 * kmeans_batch_sweep, rank split: oracle/np_dist.py).  No reference counterpart: the reference
 * updates its statistics item by item (kmeans_components.py:93-166); the batch sweep rebuilds
 * them once per sweep in a fixed order.  Two calls, so that a multi-GPU run can exchange ONE packed
 * record per rank between them (an all-gather); `utt_lo..utt_hi` is the range of utterances owned
 * by this rank.  The sweep never touches `assignments` (all old items are deleted, all new tokens
 * added: the array is a pure function of the token lists) -- it is materialised on demand by
 * segk_kmeans_assignments_from_tokens.
 *
 * Record of one rank, in 8-byte words (nbl = n_blocks_local, FW = (2 + 3*flag_cap + 1) / 2):
 *   [part_sum nbl*K_max*D double][part_tot nbl double][part_cnt nbl*K_max int64][flags nbl*FW]
 *   flags of a block, as int32: {count, 0, (slot = utt*N_max + t, k, embedding row) x flag_cap};
 *   with flag_rows (ABI 6: a rank that holds only a SHARD of the corpus -- its own utterances' rows, numbered from 0 -- cannot
 *   read the rows of other ranks' tokens from X) followed by [rows nbl*RW], RW = (flag_cap * D * sizeof(dtype of X) + 7) / 8:
 *   the embedding rows of the block's flagged tokens themselves, in list order, which segk_kmeans_batch_finalize then reads
 *   instead of X (same values, same sums).  segk_kmeans_batch_record_words() returns the record's length
 *   (flag_row_bytes = D * sizeof(dtype of X) with flag_rows, else 0).
 *
 *  (1) segk_kmeans_batch_partials: per statistics block b (utterances [blk_lo[b], blk_lo[b+1]),
 *      blk_lo [dev] int32 [n_blocks_local + 1]) the sequential fp64 sum of its tokens per component, in
 *      token order (utterance, segment), read from the slot arrays new_tok / new_k [dev] int32
 *      [n_utt, N_max] as segk_kmeans_segment leaves them (unused slots: k = -1): a stable counting sort
 *      of the block's tokens by component (sorted_scratch: one region per (block, range of components);
 *      koff_scratch: {offset, length} of every (block, component) list; both [dev] int32, their sizes in words from
 *      segk_kmeans_batch_scratch_words -- ABI version 4; a block's regions sit at sorted_scratch + blk_lo[b] * N_max * NR
 *      words (NR = sorted_words / n_slots), so n_slots = n_utt * N_max covers any blocks, and a rank whose blocks begin at
 *      utterance u0 > 0 may allocate for n_slots = (its utterances) * N_max and pass the address of its buffer MINUS
 *      u0 * N_max * NR words: only the regions of the blocks named are touched), then one sequential sum per
 *      (block, component);
 *      part_tot = sum of out_total in utterance order.  Tokens whose argmax is an inactive row (k >= K;
 *      n_flag [dev] int32 [n_utt] counts them per utterance) are listed instead, in token order.
 *      Writes out_scalars[3] = K (before the sweep) and zeroes m->mnorm_max for (2).
 *  (2) segk_kmeans_batch_finalize: over the records of ALL ranks (records [dev], rank r at
 *      records + r * rank_stride words; n_blocks_total <= 64 blocks, n_blocks_per_rank per record):
 *      replay of the reference's `k > K -> K` clamp (kmeans_components.py:102-106) over the flagged
 *      tokens in global token order, fixed balanced-tree combination of the blocks' partial sums
 *      (components founded this sweep: sequential sums of their flagged tokens per block, same tree),
 *      means = numerators / counts, clean_components (:263-266) as a relabel table (remap_scratch [dev]
 *      int32 [K_max]), final labels of the local tokens' new_k, rebuild of the MFMA operand images
 *      (as segk_kmeans_prepare + segk_kmeans_mark_duplicates).
 *      out_scalars [dev] double [4] = {sum of totals, K, n_tokens, K before the sweep}.
 *  status bits: 1 new segment without embedding, 2 add_item on an assigned item
 *  (kmeans_components.py:101 assert), 4 more flagged tokens in one block than flag_cap (the record has room for
 *  flag_cap per block; the tokens beyond were dropped and the statistics of this sweep are NOT usable: restore a
 *  checkpoint or rebuild the state, and sweep again with a larger flag_cap.  There is no per-sweep limit).
 */
int64_t segk_kmeans_batch_record_words(int32_t K_max, int32_t D, int32_t n_blocks_local, int32_t flag_cap,
                                       int32_t flag_row_bytes);
int32_t segk_kmeans_batch_scratch_words(int32_t K_max, int64_t n_slots, int32_t n_blocks_local, int64_t *sorted_words,
                                        int64_t *koff_words);
int32_t segk_kmeans_batch_partials(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                   const int32_t *blk_lo, int32_t n_blocks_local,
                                   const int32_t *new_tok, const int32_t *new_k, const int32_t *n_flag,
                                   const double *out_total, int32_t *sorted_scratch,
                                   int32_t *koff_scratch, double *record, int32_t flag_cap, int32_t flag_rows,
                                   double *out_scalars, void *stream);
int32_t segk_kmeans_batch_finalize(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                                   int32_t utt_lo, int32_t utt_hi, const double *records,
                                   int32_t n_blocks_total, int32_t n_blocks_per_rank,
                                   int64_t rank_stride, int32_t flag_cap, int32_t flag_rows, int32_t my_rank,
                                   int32_t *new_k, int32_t *remap_scratch, double *out_scalars,
                                   int32_t *status, void *stream);
/* Record values of a batch sweep in one place (kmeans_acoustic_wordseg.py:405-420 keeps sum_neg_sqrd_norm,
 * sum_neg_len_sqrd_norm, components, n_tokens per iteration): after segk_kmeans_batch_finalize, adds
 * KMeansComponents.sum_neg_sqrd_norm (kmeans_components.py:234-247) of the tokens of utterances [utt_lo, utt_hi)
 * -- from the token lists, `assignments` is not needed -- to out_scalars[4] (zeroed by the finalize call) and copies
 * status[0], status[1] to out_scalars[5], [6]: out_scalars [dev] double [8] = {sum of totals, K, n_tokens, K before,
 * sum_neg_sqrd_norm, status bits, fully scanned rows, -}; one device-to-host copy serves the whole record.        */
int32_t segk_kmeans_batch_record(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int32_t utt_lo,
                                 int32_t utt_hi, const int32_t *new_tok, const int32_t *new_k,
                                 const int32_t *status, double *out_scalars, void *stream);
/* assignments[:] = -1, then assignments[new_tok] = new_k for the tokens of utterances
 * [utt_lo, utt_hi). */
int32_t segk_kmeans_assignments_from_tokens(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m,
                                            int32_t utt_lo, int32_t utt_hi, const int32_t *new_tok,
                                            const int32_t *new_k, const int32_t *n_new,
                                            void *stream);

/* KMeansComponents.sum_neg_sqrd_norm (kmeans_components.py:234-247), record metric.
 * out [dev] double [1]; tolerance-level parity (summation order differs). */
int32_t segk_kmeans_sum_neg_sqrd_norm(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m,
                                      double *out, void *stream);

/* -------------------------------------------------------------------------------------
 * FBGMM components: device image of `GaussianComponentsFixedVar`
 * (gaussian_components_fixedvar.py:20-126) or `GaussianComponentsDiag`
 * (gaussian_components_diag.py:19-120) plus the mixture hyper-parameters of `FBGMM`
 * (fbgmm.py:57-64).  All statistics are float64 as in the reference.
 *   cov_type 0 "fixed": prior_a = precision (1/var), prior_b = mu_0, prior_c = precision_0;
 *        stat_a = mu_N_numerators, stat_b = precision_Ns, log_prod = log_prod_precision_preds,
 *        pred = precision_preds
 *   cov_type 1 "diag":  prior_a = S_0, prior_b = m_0, (prior_c unused), k_0, v_0;
 *        stat_a = m_N_numerators, stat_b = S_N_partials, log_prod = log_prod_vars,
 *        pred = inv_vars
 * ------------------------------------------------------------------------------------- */
typedef struct segk_fbgmm {
    int32_t cov_type;
    int32_t K_max;
    double alpha;              /* Dirichlet concentration (fbgmm.py:59)                       */
    double lms;                /* language-model scaling factor (fbgmm.py:62)                 */
    double k_0, v_0;           /* NIW scalars (diag only)                                     */
    const double *prior_a;     /* [dev] [D]                                                   */
    const double *prior_b;     /* [dev] [D]                                                   */
    const double *prior_c;     /* [dev] [D]                                                   */
    double *stat_a;            /* [dev] [K_max, D]                                            */
    double *stat_b;            /* [dev] [K_max, D]                                            */
    double *log_prod;          /* [dev] [K_max]                                               */
    double *pred;              /* [dev] [K_max, D]                                            */
    int64_t *counts;           /* [dev] [K_max]                                               */
    int32_t *assignments;      /* [dev] [n_emb]                                               */
    int32_t *K;                /* [dev] [1]                                                   */
    /* optional bigram language model tied to the components (BigramSmoothLM, bigram_lms.py:17-47;
     * passed to GaussianComponentsFixedVar as `lm`, gaussian_components_fixedvar.py:90): when
     * lm_unigram != NULL the assignment prior of the Dirichlet-multinomial is replaced by the LM
     * (bigram_acoustic_wordseg.py:314-384) and del_component rewires the LM counts
     * (gaussian_components_fixedvar.py:204-221). */
    int64_t *lm_unigram;       /* [dev] [K_max] or NULL                                       */
    int64_t *lm_bigram;        /* [dev] [K_max, K_max]: (j, i) = count of i following j       */
    double lm_lambda, lm_a, lm_b;
    double *kconst;            /* [dev] [K_max + 1] derived: x-independent constant of each component's
                                * predictive; entry K_max = that of the prior predictive        */
} segk_fbgmm;

/* Record metrics of one sweep on the device: out [dev] double [4] =
 *   { log_prob_z, log_prob_X_given_z, K, number of assigned rows } of the sequential-mode state in `f`:
 *   log_prob_z            FBGMM.log_prob_z, fbgmm.py:208-225 (urn = 0: Dirichlet-multinomial with f->alpha), or
 *                         BigramAcousticWordseg.log_prob_z, bigram_acoustic_wordseg.py:287-305 (urn = 1, urn_a =
 *                         lm.a: the reference's loop never advances j_prev, so it is the Polya-urn probability of
 *                         the tokens under the smoothed unigram model -- a function of the counts);
 *   log_prob_X_given_z    components.log_marg(): gaussian_components_fixedvar.py:261-296 (sums of x and x^2 per
 *                         component in the dtype of X, rows ascending, as numpy's axis-0 reduction) or
 *                         gaussian_components_diag.py:271-303.
 * Values agree with the reference's to ~1e-12 relative (contract for record values: 1e-8); workspace: the context's. */
int32_t segk_fbgmm_record_metrics(segk_ctx *ctx, const segk_corpus *c, const struct segk_fbgmm *f,
                                  int32_t urn, double urn_a, double *out, void *stream);

/* Components __init__ from `assignments` (fixedvar:110-120 / diag:114-120): statistics summed in
 * the order of the reference's add_item loop (k ascending, rows ascending), counts, K. */
int32_t segk_fbgmm_init_stats(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, void *stream);

/* A11 for FBGMM components, reference order and arithmetic, one workgroup:
 *   op 0: del_item for every segment of the CURRENT segmentation of utterance `utt`
 *         (unigram_acoustic_wordseg.py:270-273; boundaries [dev] uint8 [n_utt, N_max])
 *   op 1: add_item(item, k)  (fixedvar:153-170 / diag:162-177)
 *   op 2: del_item(item)     (fixedvar:172-188 / diag:179-194; deletes the component when it
 *         empties: del_component fixedvar:190-221 / diag:196-213, swap-last compaction)
 *   op 4: del_component(k)
 *   op 5 / op 6: lm.remove_counts_from_utterance / lm.counts_from_utterance for the current
 *         transcript of utterance `utt` (bigram_lms.py:98-114; bigram_acoustic_wordseg.py:410,496);
 *         utt < 0: every utterance (set_lm_counts, bigram_acoustic_wordseg.py:271-276)          */
int32_t segk_fbgmm_update(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, int32_t op,
                          int32_t utt, int64_t item, int32_t k, const uint8_t *boundaries,
                          void *stream);

/* A2/A3/A4 -- FBGMM.log_marg_i (fbgmm.py:256-285) = logsumexp over K_max of
 * lms*(log(alpha/K_max + counts) - log(sum counts + alpha)) + log_post_pred (k < K,
 * fixedvar:242-253 / diag:237-259) or log_prior (k >= K, fixedvar:224-231 / diag:215-222), for
 * rows ids[0..n) (row0..row0+n-1 when NULL; -1 skipped); out [dev] double [n_emb] indexed by row. */
int32_t segk_fbgmm_score(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                         const int32_t *ids, int64_t row0, int64_t n, double *out, void *stream);

/* A2/A3 vector API: out [dev] double [K_max + 1]: out[k] = log_post_pred(row)[k] for k < K
 * (fixedvar:242-253 / diag:237-259), out[K_max] = log_prior(row) (fixedvar:224-231 / diag:215-222). */
int32_t segk_fbgmm_pred_vector(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                               int64_t row, double *out, void *stream);

/* A5 + A6/A7 for one utterance -- get_vec_embed_log_probs (unigram_acoustic_wordseg.py:474-511)
 * from the per-row scores, then forward_backward (:653-756) or forward_backward_viterbi
 * (:759-864).  Backward sampling consumes ustream[*ucursor ...] (one value per emitted segment,
 * the reference's random.random() calls) and advances the device-resident cursor.
 *   new_tok [dev] int32 [n_utt, N_max], n_new [dev] int32 [n_utt], out_logprob [dev] double [n_utt]
 *   status bits: 8 uniform stream exhausted, 16 log_prob == -inf (the reference asserts, :753)
 * viterbi == 2: `assignments_only` of bigram_acoustic_wordseg.py:386-387 -- boundaries are kept,
 * new_tok / n_new list the current segments and out_logprob[utt] = 0.                         */
int32_t segk_unigram_segment(segk_ctx *ctx, const segk_corpus *c, int32_t utt, int32_t viterbi,
                             int32_t n_slices_min, int32_t n_slices_max, double wip,
                             double time_power_term, double log_p_continue, double anneal_temp,
                             const double *score, const double *ustream, int64_t *ucursor,
                             int64_t ucap, uint8_t *boundaries, int32_t *new_tok, int32_t *n_new,
                             double *out_logprob, int32_t *status, void *stream);

/* A10 for the new segments of one utterance, in order, statistics updated between segments:
 * gibbs_sample_inside_loop_i (fbgmm.py:422-463; one uniform per segment, utils.draw forward
 * order, `k > K -> K`) or map_assign_i (:465-494) when map_assign != 0.
 * With an LM attached (f->lm_unigram != NULL): log_marg_i_embed_unigram and
 * gibbs_sample_inside_loop_i_embed (bigram_acoustic_wordseg.py:314-384): the prior of the first
 * segment is lms*lm.log_prob_vec_i(), of every later one lms*log(lm.prob_vec_given_j(k_prev)).
 * j_prev: component of the segment preceding new_tok[utt][0], -1 = none (read only with an LM). */
int32_t segk_fbgmm_assign(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, int32_t utt,
                          int32_t map_assign, int32_t j_prev, double anneal_temp, const int32_t *new_tok,
                          const int32_t *n_new, const double *ustream, int64_t *ucursor,
                          int64_t ucap, int32_t *status, void *stream);

/* A12: UnigramAcousticWordseg.gibbs_sample_i (unigram_acoustic_wordseg.py:252-360) for every utterance of order[0 .. n_order)
 * [host] in turn -- remove its segments, score its spans (segk_fbgmm_score), sample or maximise its boundaries
 * (segk_unigram_segment with viterbi 0 / 1), assign the new segments (segk_fbgmm_assign, map_assign) -- by ONE persistent
 * kernel per stretch of utterances between two emptied components (ABI 6): every workgroup keeps the whole model in LDS and
 * replays every update, only the span scores are shared out, one grid barrier per utterance.  Same device functions, same
 * order of operations, same uniforms (ustream / ucursor as in the per-utterance calls): the same bits as the four calls per
 * utterance.  With a language model attached (f->lm_unigram): BigramAcousticWordseg.gibbs_sample_i
 * (bigram_acoustic_wordseg.py:386-551) -- the counts of the utterance's transcript removed first and those of the new one
 * added last (segk_fbgmm_update op 5 / 6), spans scored and segments assigned under the language model; every workgroup
 * keeps its own copy of the bigram counts (K_max^2 int64 in global memory, owned by the context).
 * row_start [dev] int32 [n_utt + 1]: first row of every utterance (an utterance's rows are contiguous);
 * score [dev] double [n_emb] scratch.  Returns SEGK_ERR_UNSUPPORTED, with nothing enqueued, where the kernel does not apply
 * (more than 64 landmarks, 3 K_max D doubles beyond a workgroup's LDS, an utterance listed twice, SEGK_FB_CHAIN=0): the
 * caller then makes the calls per utterance.  SYNCHRONISES the stream after every launch.                                  */
int32_t segk_fbgmm_sequential_sweep(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, const int32_t *order, int32_t n_order,
                                    const int32_t *row_start, int32_t viterbi, int32_t map_assign, int32_t n_slices_min,
                                    int32_t n_slices_max, double wip, double time_power_term, double log_p_continue,
                                    double anneal_temp_fb, double anneal_temp_am, double *score, const double *ustream,
                                    int64_t *ucursor, int64_t ucap, uint8_t *boundaries, int32_t *new_tok, int32_t *n_new,
                                    double *out_logprob, int32_t *status, void *stream);

/* Inner loop of FBGMM.gibbs_sample (fbgmm.py:352-405) over the rows ids[0..n) (ids == NULL: rows
 * 0..n-1) in order: cache_component_stats, del_item, logits (:364-372), annealing, utils.draw with
 * one uniform of the stream per considered row, then restore_component_from_stats when the row
 * went back to its component and no component was deleted, add_item otherwise.  Rows whose
 * assignment is -1 are skipped unless consider_unassigned != 0.                                */
int32_t segk_fbgmm_gibbs_items(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f, const int32_t *ids,
                               int64_t n, int32_t consider_unassigned, double anneal_temp,
                               const double *ustream, int64_t *ucursor, int64_t ucap, int32_t *status,
                               void *stream);

/* -------------------------------------------------------------------------------------
 * Batch-synchronous ("blocked parallel Gibbs") sweep of the FBGMM / bigram samplers.  The
 * reference's samplers (unigram_acoustic_wordseg.py:252-472, bigram_acoustic_wordseg.py:386-671)
 * are serial chains with no parallel mode; this family implements the sampler specified in
 * oracle/np_fbgmm_batch.py from the same building blocks: the utterances are cut into n_slices
 * contiguous slices (unit of GPU ownership and of the fixed summation order) x n_blocks blocks;
 * Gibbs step b resamples block b of every slice in parallel conditioned on all other blocks.
 * Component labels are "slots" 0..K_max-1 that are not renumbered during a sweep (count 0 = empty
 * component = prior predictive); segk_fbb_canonical gives the reference's contiguous labelling.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_slices, n_blocks;   /* S (<= 16), B (>= 2)                                          */
    int32_t u_max;                /* max utterances of a (slice, block); 0 without an LM store    */
    int32_t fast_dp;              /* boundary sampler of the batch steps: 0 = fp64 library exp / log (bit-exact against the
                                   * specification), 1 = hardware v_exp_f32 / v_log_f32 (tolerance modes f32 / f16)  */
    const int32_t *utt_range;     /* [dev] [S, B, 2] utterances [lo, hi) of (slice, block)        */
    const int32_t *row_range;     /* [dev] [S, B, 2] embedding rows [lo, hi) of (slice, block)    */
    double *partials;             /* [dev] [B, S, K_max*(2D+1)]: counts [K_max] (as doubles), sum x
                                   * [K_max, D], sum x^2 [K_max, D] of the tokens of (block, slice) */
    double *cnt;                  /* [dev] [K_max]   counts of the current exclusion              */
    double *mean_t, *q_t;         /* [dev] [D, K_max] predictive mean / scale, slot-contiguous    */
    double *lconst, *zconst, *half; /* [dev] [K_max] likelihood constant, + assignment prior, multiplier */
    double *scal;                 /* [dev] [2]: total count, number of occupied slots             */
    int32_t *slot;                /* [dev] [n_emb]   slot of every embedding row, -1 = unassigned */
    int32_t *lm_tok;              /* [dev] [B, S, u_max, N_max] slots of every utterance's segments,
                                   * -1 padded (replicated on every rank); NULL without an LM     */
    uint64_t seed;                /* of the counter-based uniforms u01(seed, sweep, utt, j)       */
    /* optional fp32 matrix-core span score of fixed-variance components (segk_fbb_score_f32):      */
    float *y;                     /* [dev] [n_emb, ldy] rows [x_0^2, x_0, x_1^2, x_1, ...] (segk_fbb_make_y) */
    int64_t ldy;                  /* 2D rounded up to a multiple of 4                             */
    float *tiles32;               /* [dev] segk_kmeans_tiles_floats(K_max + 1, 2D) floats: operand
                                   * image of the per-slot [-pp/2, pp*mu] rows, written by
                                   * segk_fbb_prepare when non-NULL                               */
    /* optional fp16x2 form of the same score (2D <= 208): both operands as two fp16 pieces on the
     * 16-bit matrix pipe, like the k-means filter; needs y and, when non-NULL, supersedes tiles32:  */
    void *y16;                    /* [dev] segk_corpus_b3_bytes(n_emb, 2D) bytes (segk_fbb_make_y)  */
    float *tiles16;               /* [dev] segk_kmeans_tiles_b3_floats(K_max + 1, 2D) floats        */
    float *rows32;                /* [dev] [(K_max + 1), 2D] scratch: the per-slot rows in float32   */
    double *consts16;             /* [dev] [2 (K_max + 2) + 32] scratch: per-column constants; [K_max + 1] = max |row|^2; behind
                                     them the int32 column maps of the packed image (segk_fbb_prepare writes, the score calls read) */
    /* optional: the prior predictive of every embedding row (an empty slot's likelihood) -- a constant of corpus and prior, so
     * the score and assignment kernels of every Gibbs step need not evaluate its D logarithms per row again
     * (segk_fbb_prior_rows once; NULL: evaluated in the kernels; the values are the same either way)                           */
    const double *prior_rows;     /* [dev] [n_emb] or NULL                                         */
} segk_fbatch;

/* token lists of all utterances from the boundaries: new_tok [n_utt, N_max], n_new [n_utt]     */
int32_t segk_fbb_collect(segk_ctx *ctx, const segk_corpus *c, const uint8_t *boundaries,
                         int32_t *new_tok, int32_t *n_new, void *stream);
/* partials[b][s] for the slices s_lo .. s_lo+s_n-1 from their token lists and bt->slot: per slot
 * sequential in token order (utterance, then segment); x^2 is the square in the dtype of X
 * (gaussian_components_diag.py:125).  SIDE EFFECT: bt->scal (the totals segk_fbb_prepare accumulates into) is left
 * zeroed, so that a segk_fbb_prepare enqueued next on the same stream need not clear it; the totals are therefore only
 * valid between a segk_fbb_prepare and the next segk_fbb_partials -- call segk_fbb_prepare before reading them (or before
 * segk_fbb_score / _assign / _token_scores) after any segk_fbb_partials.                                              */
int32_t segk_fbb_partials(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                          const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                          const int32_t *new_tok, const int32_t *n_new, void *stream);
/* statistics of all tokens outside block b (b = -1: all): per slice the partials of the blocks
 * b' != b in increasing b', slices combined by the balanced tree of the specification; then the
 * predictive parameters of every slot (fixedvar:153-170,317-325 / diag:162-177,332-345).        */
int32_t segk_fbb_prepare(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                         const segk_fbatch *bt, int32_t b, void *stream);
/* log_marg_i (fbgmm.py:256-285; with an LM bigram_acoustic_wordseg.py:314-329) of every row of
 * block b of the local slices under the prepared statistics -> score[row].
 * n_rows [host] [s_n]: rows of (s_lo + i, b).                                                    */
int32_t segk_fbb_score(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                       const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                       const int32_t *n_rows, double *score, void *stream);
/* The same scores within the 1e-4 tolerance of the path, on the fp32 matrix cores (fixed-variance
 * components only): the logit of slot k is a contraction of [x^2, x] with [-pp_k/2, pp_k*mu_k] plus
 * a constant, all empty slots together are one more pseudo-component (the prior predictive, weight
 * = their number); log-sum-exp accumulated online in the MFMA kernel's epilogue.  Needs bt->y
 * (segk_fbb_make_y once) and bt->tiles32 (filled by segk_fbb_prepare).                            */
int32_t segk_fbb_make_y(segk_ctx *ctx, const segk_corpus *c, const segk_fbatch *bt, void *stream);
/* out[row] = log prior predictive of X[row] (gaussian_components_fixedvar.py:224-231 / gaussian_components_diag.py:215-222), the
 * value the score / assignment kernels use for an empty slot: for segk_fbatch.prior_rows.  Valid while X and the prior of `f`
 * stay what they are.                                                                                                          */
int32_t segk_fbb_prior_rows(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, double *out, void *stream);
int32_t segk_fbb_score_f32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                           const segk_fbatch *bt, const int32_t *rows, int64_t n, double *score,
                           void *stream);   /* rows [dev] [n]: the embedding rows to score (a block) */
/* Roofline calibration of segk_fbb_score_diag32 (no reference counterpart): Student-t terms per second of a kernel
 * that does nothing but the score kernel's inner term (subtract, multiply, multiply-add, v_log_f32, accumulate) from
 * registers on every vector ALU of the chip.  out_terms_per_s [host] double; synchronises `stream`.          */
int32_t segk_calibrate_vlog(segk_ctx *ctx, double *out_terms_per_s, void *stream);
/* segk_fbb_score for diagonal (Student-t) components in float32 with the hardware logarithm (opt-in,
 * `score_precision="f32"`): same arguments and result buffer; the reference expression is
 * gaussian_components_diag.py:237-259, 347-360 inside FBGMM.log_marg_i (fbgmm.py:256-285).  Error against the fp64
 * kernel <= 1e-4 relative to max(|log_marg_i|, 1) (the contract of the path; measured ~2e-6).                */
int32_t segk_fbb_score_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                              const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                              const int32_t *n_rows, double *score, void *stream);
/* get_vec_embed_log_probs + forward_backward (unigram...:474-511, 653-756) for every utterance of
 * block b of the local slices with the uniforms u01(seed, sweep, utt, 0, 1, ...); the slots of the
 * old segments are cleared.  n_utts [host] [s_n].  status bit 16: log_prob == -inf.            */
int32_t segk_fbb_segment(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                         const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                         const int32_t *n_utts, uint64_t sweep, int32_t n_slices_min,
                         int32_t n_slices_max, double wip, double time_power_term,
                         double anneal_temp, const double *score, uint8_t *boundaries,
                         int32_t *new_tok, int32_t *n_new, double *out_logprob, int32_t *status,
                         void *stream);
/* slot of every new segment: softmax of the logits (fbgmm.py:436-457; with an LM
 * bigram_acoustic_wordseg.py:332-384 chained over the utterance's segments), utils.draw with
 * u01(seed, sweep, utt, N_max + position); no `k > K` clamp -- slots are drawn as such.        */
int32_t segk_fbb_assign(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                        const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                        const int32_t *n_utts, uint64_t sweep, double anneal_temp,
                        const int32_t *new_tok, const int32_t *n_new, const float *ll_mat,
                        int64_t ll_ld, void *stream);
/* The same with the token likelihoods of diagonal (Student-t) components in float32 -- v_log_f32 terms as in
 * segk_fbb_score_diag32, z and the draws in fp64 as above (`score_precision="f32"`: log-likelihoods within 1e-4
 * relative of the fp64 form; the fp64 software logarithm of K_max * D terms per token is otherwise most of the
 * assignment step).                                                                                        */
int32_t segk_fbb_assign_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                               const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                               const int32_t *n_utts, uint64_t sweep, double anneal_temp,
                               const int32_t *new_tok, const int32_t *n_new, void *stream);
/* One Gibbs step of the diagonal (Student-t) sampler in float32 terms as ONE launch (ABI 8): segk_fbb_score_diag32,
 * segk_fbb_segment and segk_fbb_assign_diag32 of block b fused -- the workgroup that owns an utterance scores its spans,
 * samples its boundaries (the same uniforms, unigram_acoustic_wordseg.py:653-864) and draws the new segments' slots from the
 * logits it already holds (fbgmm.py:422-463); `score` [dev] double [n_emb] still receives the span scores, bit for bit those
 * of segk_fbb_score_diag32, and the boundaries are those segk_fbb_segment samples from them; the token likelihoods are
 * float32 terms with one multiply-add where segk_fbb_assign_diag32 multiplies and adds (both within the 1e-4 contract).
 * Needs bt->prior_rows; honours segk_fbb_set_probe.  Returns SEGK_ERR_UNSUPPORTED, with nothing enqueued, where it does not
 * apply (a language model, K_max > 256, tables + logits beyond a workgroup's LDS, SEGK_FBB_FUSED=0): the caller then makes
 * the three calls.                                                                                                        */
int32_t segk_fbb_step_diag32(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f, const segk_fbatch *bt,
                             int32_t s_lo, int32_t s_n, int32_t b, const int32_t *n_utts, uint64_t sweep,
                             int32_t n_slices_min, int32_t n_slices_max, double wip, double time_power_term,
                             double anneal_temp_fb, double anneal_temp_am, double *score, uint8_t *boundaries,
                             int32_t *new_tok, int32_t *n_new, double *out_logprob, int32_t *status, void *stream);
/* ll_mat of segk_fbb_assign (optional; fixed-variance components with the fp16x2 images): the token
 * likelihoods come from the matrix-core contraction instead of the fp64 VALU loop.  Row j*N_max + t of
 * ll_mat [n, ll_ld] belongs to segment t of the j-th utterance of the block (local slices in order);
 * segk_fbb_token_scores fills it for the row list tok_rows[j*N_max + t] = new_tok[utt_j][t].  The columns
 * are those of the packed operand image of the step's segk_fbb_prepare (occupied slots in slot order, then the
 * pseudo-component of the empty ones), which segk_fbb_assign of the same step resolves; not a per-slot table. */
int32_t segk_fbb_token_scores(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                              const segk_fbatch *bt, const int32_t *tok_rows, int64_t n,
                              float *ll_mat, int64_t ll_ld, void *stream);
/* Diagnostic probes of the tolerance modes (no reference counterpart; used by the parity tests to hold the float32 /
 * fp16 token likelihoods and the hardware-exp/log forward filter to the 1e-4 contract directly, not through the draws
 * they feed).  While set, on this context:
 *   segk_fbb_segment also writes the forward filter's alpha[t] (unigram_acoustic_wordseg.py:691-703), t < N, of every
 *     utterance it samples to alpha_out[utt * N_max + t]                     (alpha_out [dev] double [n_utt, N_max]);
 *   segk_fbb_assign / segk_fbb_assign_diag32 also write, for token t of utterance utt and every slot k < K_max, the
 *     log predictive density of the token under slot k that enters its logits (posterior predictive of an occupied
 *     slot, prior predictive of an empty one: oracle/np_fbgmm_batch.py `loglik`) to
 *     ll_out[(utt * N_max + t) * ll_ld + k]                                 (ll_out [dev] double [n_utt * N_max, ll_ld]).
 * NULL pointers (the default) switch a probe off.                                                                  */
int32_t segk_fbb_set_probe(segk_ctx *ctx, double *alpha_out, double *ll_out, int64_t ll_ld);
/* bigram table += sign * (transcripts of block b, all slices) from bt->lm_tok                  */
int32_t segk_fbb_lm_apply(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                          const segk_fbatch *bt, int32_t b, int32_t sign, void *stream);
/* bt->lm_tok[b][s] <- slots of the new tokens of the local slices                              */
int32_t segk_fbb_lm_fill(segk_ctx *ctx, const segk_corpus *c, const segk_fbgmm *f,
                         const segk_fbatch *bt, int32_t s_lo, int32_t s_n, int32_t b,
                         const int32_t *n_utts, const int32_t *new_tok, const int32_t *n_new,
                         void *stream);
/* after segk_fbb_prepare(b = -1): remap [dev] [K_max] occupied slots -> 0..K-1 in increasing slot
 * order (-1 for empty), f->assignments[row] = remap[slot[row]], *f->K = K.                      */
int32_t segk_fbb_canonical(segk_ctx *ctx, const segk_corpus *c, segk_fbgmm *f,
                           const segk_fbatch *bt, int32_t *remap, void *stream);

/* -------------------------------------------------------------------------------------
 * A6 / A7 / A8 on caller-supplied score vectors -- drop-in for the module-level functions
 *   kind 0: forward_backward_kmeans_viterbi   kmeans_acoustic_wordseg.py:449-555
 *   kind 1: forward_backward_viterbi          unigram_acoustic_wordseg.py:759-864
 *   kind 2: forward_backward                  unigram_acoustic_wordseg.py:653-756
 * n_prob independent problems; problem p has Ns[p] landmarks and its triangular vector
 * (N(N+1)/2 doubles, entry t(t-1)/2+s = span [s,t)) starts at vecs + offs[p].  kind 2
 * consumes uniforms[p*u_stride + j] in place of the j-th random.random() call
 * (_cython_utils.pyx:83) and reports the number consumed in n_draws[p]; status[p] = 1 where
 * the reference would `assert False` (log_prob == -inf, :753).
 *   bounds [dev] uint8 [n_prob, b_stride], totals [dev] double [n_prob],
 *   work [dev] double [n_prob, w_stride], w_stride >= 3*N_max + 2.
 * ------------------------------------------------------------------------------------- */
int32_t segk_dp_tri(segk_ctx *ctx, int32_t kind, const double *vecs, const int32_t *Ns,
                    const int64_t *offs, int32_t n_prob, int32_t n_slices_min, int32_t n_slices_max,
                    double log_p_continue, double anneal_temp, const double *uniforms,
                    int64_t u_stride, uint8_t *bounds, int64_t b_stride, double *totals,
                    int32_t *n_draws, int32_t *status, double *work, int64_t w_stride, void *stream);

/* -------------------------------------------------------------------------------------
 * A9 host shims -- _cython_utils.pyx:13-25 (logsumexp), :75-89 (draw, uniform supplied by
 * the caller instead of random.random()), :30-70 (sums).  Pure host functions on [host]
 * buffers, kept so that `segmentalist_amd._cython_utils` is a drop-in module.
 * ------------------------------------------------------------------------------------- */
double segk_logsumexp(const double *a, int64_t n);
int32_t segk_draw(const double *p_k, int64_t n, double u);
double segk_sum_doubles(const double *y, int64_t n);
int64_t segk_sum_ints(const int64_t *y, int64_t n);
double segk_sum_log(const double *y, int64_t n);
double segk_sum_square_a_times_b(const double *a, const double *b, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* SEGK_H */
