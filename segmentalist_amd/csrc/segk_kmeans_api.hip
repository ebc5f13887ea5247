// segk_kmeans_api.hip -- C ABI of the score stage: filter selection (hinted / pre-filter / split precision / fp32 MFMA) and the sequential sweep
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include <vector>
#include "segk_kmeans_dev.h"

extern "C" {

int32_t segk_kmeans_clear_queue(segk_ctx *ctx, const segk_cand *cand, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(cand && cand->count, "cand");
    SEGK_CHECK_HIP(hipMemsetAsync(cand->count, 0, sizeof(int32_t), (hipStream_t)stream));
    return SEGK_OK;
}

// launch arguments of the filter stage for rows ids[0..n) / row0..row0+n-1 (the split-precision members are filled when
// the fp16x2 / bf16x3 images are in use: segk_use_b3)
static ScoreArgs make_score_args(const segk_corpus *c, const segk_kmeans *m, const int32_t *ids, int64_t row0, int64_t n,
                                 const segk_cand *cand, bool b3)
{
    ScoreArgs A{};
    A.X32 = c->X32; A.ld32 = c->ld32; A.ids = ids; A.row0 = row0; A.n = n;
    A.tiles = m->tiles; A.n_tiles = segk_n_tiles(m->K_max); A.tile_stride = segk_tile_stride(c->D);
    A.G = segk_G(c->D); A.D = c->D;
    A.fuse_exact = (c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128) ? 1 : 0;
    A.is_f64 = c->x_dtype == SEGK_F64;
    A.dbg = segk_dev_env("SEGK_SCORE_DBG");
    A.xnorm = c->xnorm; A.mnorm2 = m->mnorm_max; A.cand = *cand; A.amb_cap = (int)c->n_emb;
    A.n_chunks = 0; A.tiles_per_split = 0; A.part_k = nullptr; A.part_f = nullptr;
    if (b3) {
        A.xrows32 = c->X32;
        A.X32 = (const float *)c->Xb3;
        A.tiles = m->tiles_b3;
        A.tile_stride = segk_sp_tile_stride(c->D, c->sp_pieces);
        A.means32 = (const float *)m->means;
        A.fuse_exact = (c->D % 4 == 0) ? 1 : 0;
        A.K_max = m->K_max;
        A.xerr = (const float *)((const unsigned char *)c->Xb3 + SEGK_SP_HEADER + c->n_emb * 2 * (int64_t)segk_b3_kp(c->D) * 2);
    }
    return A;
}

int32_t segk_kmeans_filter(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                           int64_t row0, int64_t n, const segk_cand *cand, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = score_checks(c, m, ids, row0, n, cand);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    // 4-wave workgroups, two per CU: the two waves sharing a SIMD belong to DIFFERENT workgroups
    // and drift apart, covering each other's barrier/staging gaps.  (Measured: an 8-wave
    // workgroup, one barrier for all eight waves, locks the SIMD partners in step and is 15 %
    // slower although it halves the staging instructions per wave.)
    if (segk_use_b3(c, m)) {
        const ScoreArgs A = make_score_args(c, m, ids, row0, n, cand, true);
        // one-product pre-filter in front (two-piece images, D % 4 == 0).  Its three extra launches -- and the
        // second stage's fixed cost, one workgroup's pass over every tile with all three products (~45 us)
        // -- pay once the split-precision kernel alone would need more than four rounds of the chip
        // (estimated break-even near 130 k rows; 1 M rows: 0.58 ms against 0.77 ms).
        // SEGK_SCORE_PRE=0 disables it, =1 forces it at every size (tests).
        const char *pre_env = getenv("SEGK_SCORE_PRE");
        const int pre_mode = pre_env ? atoi(pre_env) : -1;
        // (round 2: from 384 rows per CU on -- 98 304 -- with the second stage split over component ranges for small
        // queues: a 1 250-utterance shard 5 636 against 5 329 sweeps/s, 2 500 utterances 4 493 against 4 278; at 625
        // utterances the split-precision kernel alone still wins, 7 154 against 6 273)
        const bool small_table = m->K_max < (1 << 24) && (int64_t)m->K_max * c->D * 4 < ((int64_t)1 << 32);      // the exact stage's 32-bit offsets
        if (c->sp_pieces == 2 && A.fuse_exact && small_table && pre_mode != 0 && (pre_mode == 1 || n > 384 * (int64_t)ctx->n_cu) &&
            n < (int64_t)1 << 30)
            return segk_dispatch_score_pre(ctx, A, segk_b3_kp(c->D) / 16, st);
        segk_flush_deferred_zero(ctx, st);
        return segk_dispatch_score_sp(ctx, A, segk_b3_kp(c->D) / 16, c->sp_pieces, st);
    }
    const ScoreArgs A = make_score_args(c, m, ids, row0, n, cand, false);
    segk_flush_deferred_zero(ctx, st);
    return segk_dispatch_score_f32(ctx, c, m, A, st);
}

// Where the hinted path (segk_score_hint.hip) applies: float32 data with the fp16x2 images, D % 4 == 0, enough rows to
// fill the chip (the threshold of the pre-filter path), SEGK_SCORE_HINT not 0 (=1: at every size, tests)
static bool hinted_path_applies(const segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int64_t n)
{
    const char *he = getenv("SEGK_SCORE_HINT");
    const int mode = he ? atoi(he) : -1;
    if (mode == 0 || ctx->capturing) return false;
    if (!segk_use_b3(c, m) || c->sp_pieces != 2 || c->D % 4 != 0) return false;
    if (n >= (int64_t)1 << 30) return false;
    // K1 keeps at most four LDS ranges of tile images
    {
        const int64_t map_bytes = (int64_t)((m->K_max + 3) & ~3) * 4;        // the hint waves' label map shares K1's LDS
        if (map_bytes + ((segk_b3_kp(c->D) / 16) * 256 + 32) * 4 > 160 * 1024) return false;
        int max_tiles = (int)((160 * 1024 - map_bytes) / (((segk_b3_kp(c->D) / 16) * 256 + 32) * sizeof(float)));
        if (max_tiles > 32) max_tiles = 32;              // SEGK_HINT_MAX_TPR
        if (segk_n_tiles(m->K_max) > 4 * max_tiles) return false;
    }
    return mode == 1 || n > 384 * (int64_t)ctx->n_cu;
}

int32_t segk_kmeans_score_hinted(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                                 int64_t row0, int64_t n, const segk_cand *cand, const int32_t *hint_remap,
                                 int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = score_checks(c, m, ids, row0, n, cand);
    if (rc) return rc;
    if (n <= 0) return SEGK_OK;
    if (!hinted_path_applies(ctx, c, m, n)) return segk_kmeans_score(ctx, c, m, ids, row0, n, cand, status, stream);
    hipStream_t st = (hipStream_t)stream;
    ctx->defer_zero = cand->count;                     // cleared by the path's first kernel, with its own queue length
    const ScoreArgs A = make_score_args(c, m, ids, row0, n, cand, true);
    rc = segk_dispatch_score_hint(ctx, A, hint_remap, c->n_emb, segk_b3_kp(c->D) / 16, st);
    segk_flush_deferred_zero(ctx, st);                 // (an early error return: nothing was launched)
    if (rc) return rc;
    return segk_resolve_on(ctx, c, m, ids, row0, n, cand, status, stream);
}

int32_t segk_kmeans_score(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, const int32_t *ids,
                          int64_t row0, int64_t n, const segk_cand *cand, int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc;
    if (ctx->pre_queue && cand && cand->count) {
        // the caller's queue length and the pre-filter's are cleared by the first kernel of the path the filter takes
        // (the pre-filter's own start-up kernel, or one tiny kernel instead of two 4-byte memsets)
        ctx->defer_zero = cand->count;
        rc = SEGK_OK;
    } else {
        rc = segk_kmeans_clear_queue(ctx, cand, stream);
    }
    if (rc) return rc;
    rc = segk_kmeans_filter(ctx, c, m, ids, row0, n, cand, stream);
    ctx->pre_zeroed = 0;
    segk_flush_deferred_zero(ctx, (hipStream_t)stream);        // (an early error return of the filter: nothing was launched)
    return rc ? rc : segk_resolve_on(ctx, c, m, ids, row0, n, cand, status, stream);
}

int32_t segk_kmeans_sequential_sweep(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, const int32_t *order, int32_t n_order,
                                     int32_t n_slices_min, int32_t n_slices_max, double wip, const segk_cand *cand,
                                     uint64_t *keys_scratch, uint8_t *boundaries, int32_t *old_tok, int32_t *new_tok,
                                     int32_t *new_k, int32_t *n_old, int32_t *n_new, int32_t *n_flag, double *out_total,
                                     int32_t *status, void *stream)
{
    SEGK_REQUIRE(ctx, "ctx");
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(order && n_order >= 0 && cand && cand->k && cand->s && keys_scratch, "sequential sweep operands");
    SEGK_REQUIRE(c->vec_ids && c->lengths && c->n_utt > 0, "corpus without utterances");
    if (c->x_dtype != SEGK_F32) {
        segk_set_error("segk_kmeans_sequential_sweep: float32 data only (use the per-utterance calls for float64)");
        return SEGK_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int32_t q = 0; q < n_order; q++) SEGK_REQUIRE(order[q] >= 0 && order[q] < c->n_utt, "utterance index out of range");
    // an utterance listed twice: the persistent kernel stages utterance order[q + 1] while order[q] is still being updated
    // (old boundary mask, old tokens' labels), so a repeat would read the state from before its earlier visit -- those
    // orders take the three launches per utterance, which see every update
    bool repeats = false;
    {
        std::vector<uint8_t> seen((size_t)c->n_utt, 0);
        for (int32_t q = 0; q < n_order && !repeats; q++) {
            repeats = seen[order[q]] != 0;
            seen[order[q]] = 1;
        }
    }
    // the whole sweep in one persistent kernel (segk_seq_chain.hip) where the configuration allows; SEGK_SEQ_CHAIN=0: three
    // launches per utterance
    const char *che = getenv("SEGK_SEQ_CHAIN");
    if (!(che && atoi(che) == 0) && !repeats && !ctx->capturing && (n_slices_min == 0 || n_slices_min == 1)) {
        rc = segk_launch_seq_chain(ctx, c, m, order, n_order, n_slices_max, wip, boundaries, old_tok, new_tok, new_k, n_old, n_new,
                                   n_flag, out_total, status, st);
        if (rc == SEGK_OK) return segk_kmeans_prepare(ctx, c, m, stream);
        if (rc != SEGK_ERR_UNSUPPORTED) return rc;
    }
    for (int32_t q = 0; q < n_order; q++) {
        const int32_t u = order[q];
        rc = segk_launch_seq_score(c, m, u, cand, (unsigned long long *)keys_scratch, st);
        if (rc) return rc;
        rc = segk_kmeans_segment(ctx, c, m, nullptr, u, 1, n_slices_min, n_slices_max, wip, cand, boundaries, old_tok, new_tok,
                                 new_k, n_old, n_new, n_flag, out_total, status, stream);
        if (rc) return rc;
        rc = segk_launch_update_utt(c, m, u, old_tok, new_tok, new_k, n_old, n_new, status, st);
        if (rc) return rc;
    }
    SEGK_LAUNCH_CHECK();
    // the operand images of the filters, once per sweep instead of once per utterance
    return segk_kmeans_prepare(ctx, c, m, stream);
}

}  // extern "C"
