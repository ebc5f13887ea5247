// segk_host.cpp -- context, error reporting and the A9 host shims of the C ABI.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>

#include "segk_internal.h"

static thread_local char g_err[512] = "";

void segk_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char *segk_last_error(void) { return g_err; }

int32_t segk_abi_version(void) { return SEGK_ABI_VERSION; }

int32_t segk_destroy(segk_ctx *ctx);

int32_t segk_create(int32_t device_id, segk_ctx **out_ctx)
{
    if (!out_ctx) {
        segk_set_error("segk_create: out_ctx is NULL");
        return SEGK_ERR_ARG;
    }
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        segk_set_error("segk_create: no HIP device visible (%s); libsegk has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return SEGK_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        segk_set_error("segk_create: device_id %d out of range [0,%d)", device_id, n);
        return SEGK_ERR_ARG;
    }
    hipDeviceProp_t prop;
    SEGK_CHECK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        segk_set_error("segk_create: device %d is %s; this library carries gfx950 (MI355X) code only",
                       device_id, prop.gcnArchName);
        return SEGK_ERR_NO_DEVICE;
    }
    segk_ctx *c = (segk_ctx *)calloc(1, sizeof(segk_ctx));
    if (!c) {
        segk_set_error("segk_create: out of host memory");
        return SEGK_ERR_ARG;
    }
    c->device_id = device_id;
    c->n_cu = prop.multiProcessorCount;
    strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    // one exit path: on any failure the partial allocations are released and the caller's current device
    // is restored before the error is returned
    int prev = -1;
    hipError_t err = hipGetDevice(&prev);
    const char *what = "hipGetDevice";
    if (err == hipSuccess) { what = "hipSetDevice"; err = hipSetDevice(device_id); }
    if (err == hipSuccess) { what = "hipMalloc(ws_k)"; err = hipMalloc((void **)&c->ws_k, sizeof(int32_t) * SEGK_WS_ENTRIES); }
    if (err == hipSuccess) { what = "hipMalloc(ws_f)"; err = hipMalloc((void **)&c->ws_f, sizeof(float) * 2 * SEGK_WS_ENTRIES); }
    if (err == hipSuccess) { what = "hipMalloc(ws_u64)"; err = hipMalloc((void **)&c->ws_u64, sizeof(unsigned long long) * SEGK_WS_ENTRIES); }
    if (err == hipSuccess) { what = "hipMemset(ws_u64)"; err = hipMemset(c->ws_u64, 0, sizeof(unsigned long long) * SEGK_WS_ENTRIES); }
    // (a device memset is not ordered before kernels of the caller's non-blocking streams: the workspace must read zero at its first use)
    if (err == hipSuccess) { what = "hipDeviceSynchronize"; err = hipDeviceSynchronize(); }
    if (err == hipSuccess) {
        // the feedback word of the hinted score path: pinned host memory the device writes directly (optional: without it
        // segk_kmeans_hint_feedback reports that nothing has arrived)
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            memset(hp, 0, 64);
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                c->miss_host = (volatile unsigned long long *)hp;
                c->miss_dev = (unsigned long long *)dp;
            } else {
                (void)hipHostFree(hp);
            }
        }
        (void)hipGetLastError();
    }
    if (prev >= 0) {
        const hipError_t back = hipSetDevice(prev);
        if (err == hipSuccess && back != hipSuccess) { what = "hipSetDevice(previous)"; err = back; }
    }
    if (err != hipSuccess) {
        segk_set_error("segk_create: %s -> %s", what, hipGetErrorString(err));
        segk_destroy(c);
        return SEGK_ERR_HIP;
    }
    *out_ctx = c;
    return SEGK_OK;
}

int32_t segk_destroy(segk_ctx *ctx)
{
    if (ctx) {
        if (ctx->ws_k) (void)hipFree(ctx->ws_k);
        if (ctx->ws_f) (void)hipFree(ctx->ws_f);
        if (ctx->sp2_part) (void)hipFree(ctx->sp2_part);
        if (ctx->sp2_ticket) (void)hipFree(ctx->sp2_ticket);
        if (ctx->chain_buf) (void)hipFree(ctx->chain_buf);
        if (ctx->fbchain_buf) (void)hipFree(ctx->fbchain_buf);
        if (ctx->fbchain_lm) (void)hipFree(ctx->fbchain_lm);
        if (ctx->fbchain_terms) (void)hipFree(ctx->fbchain_terms);
        if (ctx->fb_ktab) (void)hipFree(ctx->fb_ktab);
        if (ctx->fb_ptab) (void)hipFree(ctx->fb_ptab);
        if (ctx->fb_fp_dev) (void)hipFree(ctx->fb_fp_dev);
        if (ctx->fbs_buf) (void)hipFree(ctx->fbs_buf);
        if (ctx->flag_ovf) (void)hipFree(ctx->flag_ovf);
        if (ctx->hint_part) (void)hipFree(ctx->hint_part);
        if (ctx->hint_map) (void)hipFree(ctx->hint_map);
        if (ctx->ws_u64) (void)hipFree(ctx->ws_u64);
        if (ctx->hint_fb) (void)hipFree(ctx->hint_fb);
        if (ctx->pre_thr) (void)hipFree(ctx->pre_thr);
        if (ctx->band_mask) (void)hipFree(ctx->band_mask);
        if (ctx->miss_host) (void)hipHostFree((void *)ctx->miss_host);
        if (ctx->brute_ws) (void)hipFree(ctx->brute_ws);
        if (ctx->pre_queue) (void)hipFree(ctx->pre_queue);
        if (ctx->row_hash) (void)hipFree(ctx->row_hash);
        if (ctx->rb_sorted) (void)hipFree(ctx->rb_sorted);
        if (ctx->rb_koff) (void)hipFree(ctx->rb_koff);
        if (ctx->rb_misc) (void)hipFree(ctx->rb_misc);
        if (ctx->rb_term) (void)hipFree(ctx->rb_term);
        for (int i = 0; i < SEGK_PROF_SLOTS; i++)
            for (int j = 0; j < 2; j++)
                if (ctx->prof_ev[i][j]) (void)hipEventDestroy(ctx->prof_ev[i][j]);
    }
    free(ctx);
    return SEGK_OK;
}

// Timing of the main launch of the MFMA score kernel with HIP events recorded on its launch stream
// (bench.py: roofline.achieved).  While enabled every `on`-th timed launch records one pair.
int32_t segk_profile_enable(segk_ctx *ctx, int32_t on)
{
    SEGK_REQUIRE(ctx, "ctx");
    if (on)
        for (int i = 0; i < SEGK_PROF_SLOTS; i++)
            for (int j = 0; j < 2; j++)
                if (!ctx->prof_ev[i][j]) SEGK_CHECK_HIP(hipEventCreate(&ctx->prof_ev[i][j]));
    ctx->prof_on = on > 0 ? on : 0;
    ctx->prof_calls = 0;
    ctx->prof_n = 0;
    if (on) ctx->prof_kind = -1;
    return SEGK_OK;
}

// Synchronises the recorded events; ms_out / rows_out [max]: duration and row count of the main
// score launch of the most recent calls (oldest first).  Returns the number written, < 0 on error.
int32_t segk_profile_read(segk_ctx *ctx, float *ms_out, int64_t *rows_out, int32_t max)
{
    SEGK_REQUIRE(ctx && ms_out && rows_out, "arguments");
    const int have = ctx->prof_n < SEGK_PROF_SLOTS ? ctx->prof_n : SEGK_PROF_SLOTS;
    const int n = have < max ? have : max;
    for (int i = 0; i < n; i++) {
        const int slot = (ctx->prof_n - n + i) % SEGK_PROF_SLOTS;
        SEGK_CHECK_HIP(hipEventSynchronize(ctx->prof_ev[slot][1]));
        SEGK_CHECK_HIP(hipEventElapsedTime(&ms_out[i], ctx->prof_ev[slot][0], ctx->prof_ev[slot][1]));
        rows_out[i] = ctx->prof_rows[slot];
    }
    return n;
}

int32_t segk_profile_last_kind(segk_ctx *ctx)
{
    return ctx ? ctx->prof_kind : -1;
}

int32_t segk_profile_last_launches(segk_ctx *ctx)
{
    return ctx && ctx->prof_launches > 0 ? ctx->prof_launches : 1;
}

// ----------------------------------------------------------------------------------------
// hipGraph capture of a launch sequence (a whole batch sweep: a dozen kernels).  Everything the library enqueues between
// begin and end on `stream` becomes one executable graph;
// replaying it costs one host call instead of a dozen launches.  Lazy initialisation (workspace growth,
// stream / event creation, function attributes) must have happened before: run the sequence once, then
// capture the second run.  The stream must not be the legacy default stream.
// ----------------------------------------------------------------------------------------
int32_t segk_graph_begin(segk_ctx *ctx, void *stream)
{
    SEGK_REQUIRE(ctx && stream, "segk_graph_begin needs a context and a non-default stream");
    SEGK_REQUIRE(!ctx->capturing, "a capture is already open on this context");
    SEGK_REQUIRE(!ctx->prof_on, "segk_profile_enable(1) records events between launches: not inside a capture");
    SEGK_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = 1;
    return SEGK_OK;
}

int32_t segk_graph_end(segk_ctx *ctx, void *stream, void **exec_out)
{
    SEGK_REQUIRE(ctx && stream && exec_out, "arguments");
    SEGK_REQUIRE(ctx->capturing, "no capture open");
    ctx->capturing = 0;
    *exec_out = nullptr;
    hipGraph_t graph = nullptr;
    SEGK_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        segk_set_error("segk_graph_end: hipGraphInstantiate -> %s", hipGetErrorString(e));
        return SEGK_ERR_HIP;
    }
    *exec_out = (void *)exec;
    return SEGK_OK;
}

int32_t segk_graph_launch(segk_ctx *ctx, void *exec, void *stream)
{
    (void)ctx;
    SEGK_REQUIRE(exec, "exec");
    SEGK_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
    return SEGK_OK;
}

int32_t segk_graph_destroy(segk_ctx *ctx, void *exec)
{
    (void)ctx;
    if (exec) SEGK_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
    return SEGK_OK;
}

// Queue lengths of the most recent segk_kmeans_score on this context: out[0] = rows the one-product
// pre-filter passed to its second stage (-1 when the pre-filter has never run), out[1] = rows in the
// caller's ambiguity queue (full scan).  Synchronises `stream`.  Diagnostics / tests.
int32_t segk_kmeans_stage_counts(segk_ctx *ctx, const segk_cand *cand, int32_t *out, void *stream)
{
    SEGK_REQUIRE(ctx && cand && cand->count && out, "arguments");
    hipStream_t st = (hipStream_t)stream;
    out[0] = -1;
    int32_t hdr[16] = {0};                        // one counter per chunk of the pre-filter pipeline
    if (ctx->pre_queue) SEGK_CHECK_HIP(hipMemcpyAsync(hdr, ctx->pre_queue, sizeof(hdr), hipMemcpyDeviceToHost, st));
    SEGK_CHECK_HIP(hipMemcpyAsync(&out[1], cand->count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SEGK_CHECK_HIP(hipStreamSynchronize(st));
    if (ctx->pre_queue) {
        out[0] = 0;
        for (int i = 0; i < 16; i++) out[0] += hdr[i];
    }
    return SEGK_OK;
}

// How many rows the hinted score path's certificate left undecided, WITHOUT touching the stream: the band kernel of call number
// q writes (q << 32) | permille into pinned host memory; *launched = number of the last hinted call enqueued on this context,
// *seen = number of the latest call whose figure has arrived (0: none yet), *permille = that figure.  Lets a driver that
// enqueues sweeps asynchronously stop passing hints while most of them are wrong (early sweeps of a chain) -- results are the
// same either way, only the time differs.
int32_t segk_kmeans_hint_feedback(segk_ctx *ctx, uint32_t *launched, uint32_t *seen, int32_t *permille)
{
    SEGK_REQUIRE(ctx && launched && seen && permille, "arguments");
    *launched = ctx->miss_seq;
    *seen = 0;
    *permille = 0;
    if (ctx->miss_host) {
        const unsigned long long w = *ctx->miss_host;
        *seen = (uint32_t)(w >> 32);
        *permille = (int32_t)(w & 0xffffffffull);
    }
    return SEGK_OK;
}

// ----------------------------------------------------------------------------------------
// A9 host shims (_cython_utils.pyx).  Same loops, same order, libm exp/log.
// ----------------------------------------------------------------------------------------
double segk_logsumexp(const double *a, int64_t n)   // _cython_utils.pyx:13-25
{
    double mx = a[0], s = 0.0;
    for (int64_t j = 1; j < n; j++)
        if (a[j] > mx) mx = a[j];
    for (int64_t j = 0; j < n; j++) s += exp(a[j] - mx);
    return log(s) + mx;
}

int32_t segk_draw(const double *p_k, int64_t n, double u)   // _cython_utils.pyx:75-89
{
    for (int64_t i = 0; i < n; i++) {
        u = u - p_k[i];
        if (u < 0) return (int32_t)i;
    }
    return (int32_t)(n - 1);
}

double segk_sum_doubles(const double *y, int64_t n)   // :30-36
{
    double s = y[0];
    for (int64_t i = 1; i < n; i++) s += y[i];
    return s;
}

int64_t segk_sum_ints(const int64_t *y, int64_t n)   // :41-47
{
    int64_t s = y[0];
    for (int64_t i = 1; i < n; i++) s += y[i];
    return s;
}

double segk_sum_log(const double *y, int64_t n)   // :52-58
{
    double s = log(y[0]);
    for (int64_t i = 1; i < n; i++) s += log(y[i]);
    return s;
}

double segk_sum_square_a_times_b(const double *a, const double *b, int64_t n)   // :63-70
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += a[i] * a[i] * b[i];
    return s;
}

}  // extern "C"


// ---- per-device launch attributes (segk_internal.h)
static std::mutex g_attr_mu;
static std::map<std::pair<int, const void *>, size_t> g_dyn_lds;
static std::map<std::tuple<int, const void *, int, size_t>, int> g_occupancy;

hipError_t segk_dyn_lds(const void *fn, size_t lds)
{
    if (lds <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    size_t &have = g_dyn_lds[std::make_pair(dev, fn)];
    if (have >= lds) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have = lds;
    return e;
}

hipError_t segk_occupancy(const void *fn, int threads, size_t lds, int *wg_per_cu)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(g_attr_mu);
        auto it = g_occupancy.find(std::make_tuple(dev, fn, threads, lds));
        if (it != g_occupancy.end()) {
            *wg_per_cu = it->second;
            return hipSuccess;
        }
    }
    e = segk_dyn_lds(fn, lds);
    if (e != hipSuccess) return e;
    int occ = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, threads, lds);
    if (e != hipSuccess) return e;
    *wg_per_cu = occ > 0 ? occ : 1;
    std::lock_guard<std::mutex> lock(g_attr_mu);
    g_occupancy[std::make_tuple(dev, fn, threads, lds)] = *wg_per_cu;
    return hipSuccess;
}
