// segk_prepare.hip -- corpus / means -> operand images (fp32 tiles, split-precision pieces), duplicate marking
// (one of the translation units of the k-means path; shared helpers: segk_kmeans_dev.h)
#include "segk_kmeans_dev.h"

// ======================================================================================
// corpus prepare: X (f32/f64, ldx) -> X32 [n_emb, ld32] zero padded, xnorm upper bound
// ======================================================================================
template <typename XT>
__global__ void k_corpus_prepare(const XT *X, int64_t ldx, int64_t n_emb, int D, int64_t ld32,
                                 float *X32, float *xnorm)
{
    int64_t e = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (e >= n_emb) return;
    int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int d = lane; d < (int)ld32; d += 64) {
        float v = 0.f;
        if (d < D) {
            XT xv = X[e * ldx + d];
            v = (float)xv;
            s += (double)xv * (double)xv;
        }
        if (X32) X32[e * ld32 + d] = v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) xnorm[e] = (float)(sqrt(s) * (1.0 + 1e-6)) + 1e-30f;
}

// ======================================================================================
// means -> tiles
// ======================================================================================
template <typename XT>
__global__ void k_kmeans_prepare(const XT *means, int K_max, int D, float *tiles,
                                 unsigned long long *mnorm2_bits, unsigned int *zero_slot, unsigned long long *row_hash)
{
    dev_prepare_tile<XT>(means, K_max, D, tiles, mnorm2_bits, zero_slot, row_hash, (int)blockIdx.x);
}

// header of the row image: int32 {pieces, exponent a, bits of max |x_d|}
__global__ void k_corpus_maxabs(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned int *hdr)
{
    __shared__ float part[4];
    float v = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_emb * D; idx += (int64_t)gridDim.x * blockDim.x)
        v = fmaxf(v, fabsf(X[(idx / D) * ldx + (idx % D)]));
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (v > 0.f) atomicMax(hdr + 2, __float_as_uint(v));
    }
}

template <int P>
__global__ void k_corpus_split_sp(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned char *img)
{
    typedef typename SegkPiece<P>::T T;
    const int KP = segk_b3_kp(D);
    int *hdr = (int *)img;
    const int ea = P == 2 ? sp_exponent(__uint_as_float(((unsigned int *)img)[2])) : 0;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) { hdr[0] = P; hdr[1] = ea; *reinterpret_cast<int64_t *>(img + 16) = n_emb * KP; }
    if (idx >= n_emb * KP) return;
    const int64_t e = idx / KP;
    const int pos = (int)(idx - e * KP), d = segk_b3_dim(pos);
    const float x = d < D ? ldexpf(X[e * ldx + d], ea) : 0.f;
    T pc[P];
    split_sp<P>(x, pc);
    // one PLANE per piece ([P][n_emb][KP]): the one-product pre-filter streams the leading pieces alone, 2 KP contiguous
    // bytes per row (interleaved, its 224 bytes shared 128-byte lines with the second piece: 369 MB fetched for 235)
    T *row = (T *)(img + SEGK_SP_HEADER) + e * KP;
#pragma unroll
    for (int q = 0; q < P; q++) row[q * (n_emb * KP) + pos] = pc[q];
}

// the zero vector as a "mean": neg_sqd_exact against it is -|x|^2 in the reference's own float32 summation order
struct ZeroRow {
    __device__ __forceinline__ float operator[](int) const { return 0.f; }
    __device__ __forceinline__ ZeroRow operator+(int) const { return ZeroRow{}; }
};

// per row, behind the two piece planes (the image is sized for three): float [n_emb] |x - x1| (the residual norm of the
// pre-filter's margin), then float [n_emb] -|x|^2 summed exactly like the reference's -(deltas * deltas).sum() with a zero
// mean (kmeans_components.py:225-226) -- the hinted exact stage turns a reference-arithmetic score s = -|x - m|^2 into
// the filter's quantity x.m - |m|^2/2 = (s + |x|^2) / 2 with it (segk_score_hint.hip)
__global__ void k_corpus_resid_sp(const float *X, int64_t ldx, int64_t n_emb, int D, unsigned char *img)
{
    const int ea = ((const int *)img)[1];
    const int KP = segk_b3_kp(D);
    float *xerr = (float *)(img + SEGK_SP_HEADER + n_emb * 2 * (int64_t)KP * 2);
    float *nxx = xerr + n_emb;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_emb) return;
    double s = 0.0;
    for (int d = 0; d < D; d++) s += sp_resid2(ldexpf(X[e * ldx + d], ea));
    xerr[e] = (float)(ldexp(sqrt(s), -ea) * (1.0 + 1e-6)) + 1e-37f;
    nxx[e] = neg_sqd_exact<float>(ZeroRow{}, X + e * ldx, D);
}

template <int P>
__global__ void k_kmeans_prepare_sp(const float *means, int K_max, int D, float *tiles, const double *mnorm2,
                                    const unsigned char *ximg, const double *consts)
{
    dev_prepare_sp_tile<P>(means, K_max, D, tiles, mnorm2, ximg, consts, (int)blockIdx.x);
}

// ---- split-precision images of arbitrary float32 matrices (internal; used by segk_fbbatch.hip) ----
int segk_sp_prepare_rows(const float *Y, int64_t ldy, int64_t n, int D2, void *img, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(img, 0, SEGK_SP_HEADER, st));
    const int64_t nx = n * D2, tot = n * segk_b3_kp(D2);
    const int64_t blocks = (nx + 255) / 256;
    hipLaunchKernelGGL(k_corpus_maxabs, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, Y, ldy, n, D2,
                       (unsigned int *)img);
    hipLaunchKernelGGL(k_corpus_split_sp<2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Y, ldy, n, D2,
                       (unsigned char *)img);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int segk_sp_prepare_tiles(const float *rows, const double *consts, const double *rowmax2, int K, int D2, float *tiles_sp,
                          const void *ximg, void *stream)
{
    hipLaunchKernelGGL(k_kmeans_prepare_sp<2>, dim3(segk_n_tiles(K)), dim3(256), 0, (hipStream_t)stream, rows, K, D2, tiles_sp,
                       rowmax2, (const unsigned char *)ximg, consts);
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

// Exact duplicates among the rows of `means` (clean_components leaves the moved component's old row behind,
// inactive rows hold copies): a duplicate with the HIGHER index can never be np.argmax -- its score is the
// lower one's bit for bit and the first maximum wins -- but it makes every row near the pair a tie that only
// the full scan resolves (2 100 of the 2 200 queued rows of a 1 250-utterance shard had exactly these two
// contenders).  One workgroup: value hashes of all rows (8 lanes per row), then every row looks for an
// earlier row with its hash, verifies equality element by element, and if it finds one writes the
// "absent" constant (-3e38, what the padding components carry) over its accumulator seed in both tile
// images.  The full scan does not read those constants, so its first-maximum rule is untouched.
template <typename XT>
__global__ __launch_bounds__(1024) void k_kmeans_mark_dups(const XT *means, int K_max, int D, float *tiles, int stride32, int G,
                                                           float *tiles_sp, int stride_sp, int sp_const_off, int32_t *n_marked,
                                                           const unsigned long long *row_hash)
{
    // open-addressing table in LDS: key = row hash, value = the lowest row index with that hash
    constexpr int TB = 4096;                                      // slots (K_max <= 2048: load factor <= 1/2; 48 KiB)
    __shared__ unsigned long long keys[TB];
    __shared__ int32_t first[TB];
    const int tid = threadIdx.x, sub = tid & 7;
    for (int i = tid; i < TB; i += blockDim.x) { keys[i] = 0ull; first[i] = 0x7fffffff; }
    __syncthreads();
    for (int k = tid; k < K_max; k += blockDim.x) {
        const unsigned long long h = row_hash[k];
        for (unsigned slot = (unsigned)(h >> 20) & (TB - 1);; slot = (slot + 1) & (TB - 1)) {
            const unsigned long long prev = atomicCAS(&keys[slot], 0ull, h);
            if (prev == 0ull || prev == h) { atomicMin(&first[slot], k); break; }
        }
    }
    __syncthreads();
    // every row: the first row with its hash; if that is an earlier one, verify element by element (8 lanes per
    // row, all loads of a lane in flight together) and mark
    int marked = 0;
    for (int k0 = 0; k0 < K_max; k0 += 128) {
        const int k = k0 + (tid >> 3);
        int i = -1;
        if (k < K_max) {
            const unsigned long long h = row_hash[k];
            unsigned slot = (unsigned)(h >> 20) & (TB - 1);
            while (keys[slot] != h) slot = (slot + 1) & (TB - 1);
            i = first[slot];
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
            tiles[(int64_t)(k >> 5) * stride32 + G * 128 + (k & 31)] = -3.0e38f;
            if (tiles_sp) tiles_sp[1024 + (int64_t)(k >> 5) * stride_sp + sp_const_off + (k & 31)] = -3.0e38f;
            marked++;
        }
    }
    if (n_marked && marked) atomicAdd(n_marked, marked);
}

extern "C" {

int64_t segk_kmeans_tiles_floats(int32_t K_max, int32_t D)
{
    return (int64_t)segk_n_tiles(K_max) * segk_tile_stride(D);
}

int32_t segk_corpus_prepare(segk_ctx *ctx, const segk_corpus *c, float *X32_out, float *xnorm_out, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(xnorm_out != nullptr, "xnorm_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    int64_t nblk = (c->n_emb + 3) / 4;
    DISPATCH_XT(c, hipLaunchKernelGGL(k_corpus_prepare<XT>, dim3((unsigned)nblk), dim3(256), 0, st,
                                       (const XT *)c->X, c->ldx, c->n_emb, c->D, c->ld32, X32_out, xnorm_out););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}


int32_t segk_kmeans_prepare(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream)
{
    return segk_kmeans_prepare_impl(ctx, c, m, stream, false);
}

}  // extern "C"

int segk_kmeans_prepare_impl(segk_ctx *ctx, const segk_corpus *c, segk_kmeans *m, void *stream, bool mnorm_zeroed)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && m->mnorm_max, "kmeans tiles/mnorm_max");
    hipStream_t st = (hipStream_t)stream;
    // value hashes of the rows, for segk_kmeans_mark_duplicates (context-owned, K_max <= 2048 only)
    unsigned long long *row_hash = nullptr;
    if (ctx && m->K_max <= 2048) {
        if (!ctx->row_hash) SEGK_CHECK_HIP(hipMalloc((void **)&ctx->row_hash, 2048 * sizeof(unsigned long long)));
        row_hash = ctx->row_hash;
        ctx->row_hash_means = m->means;
    }
    // mnorm_max = max_k |m_k|^2, maintained by atomicMax on the bit pattern (non-negative doubles)
    if (!mnorm_zeroed) SEGK_CHECK_HIP(hipMemsetAsync(m->mnorm_max, 0, sizeof(double), st));
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_prepare<XT>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st,
                                       (const XT *)m->means, m->K_max, c->D, m->tiles,
                                       (unsigned long long *)m->mnorm_max, m->tiles_b3 ? (unsigned int *)m->tiles_b3 + 1 : nullptr,
                                       row_hash););
    if (m->tiles_b3 && c->Xb3 && c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128) {
        if (c->sp_pieces == 2)
            hipLaunchKernelGGL(k_kmeans_prepare_sp<2>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st, (const float *)m->means,
                               m->K_max, c->D, m->tiles_b3, m->mnorm_max, (const unsigned char *)c->Xb3, (const double *)nullptr);
        else if (c->sp_pieces == 3)
            hipLaunchKernelGGL(k_kmeans_prepare_sp<3>, dim3(segk_n_tiles(m->K_max)), dim3(256), 0, st, (const float *)m->means,
                               m->K_max, c->D, m->tiles_b3, m->mnorm_max, (const unsigned char *)c->Xb3, (const double *)nullptr);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

extern "C" {

int32_t segk_kmeans_mark_duplicates(segk_ctx *ctx, const segk_corpus *c, const segk_kmeans *m, int32_t *n_marked, void *stream)
{
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(m && m->tiles && m->means, "kmeans tiles / means");
    // needs the row hashes of the segk_kmeans_prepare that built these images (same context, same means buffer)
    if (!ctx || m->K_max > 2048 || !ctx->row_hash || ctx->row_hash_means != m->means) return SEGK_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool sp = m->tiles_b3 && c->Xb3 && c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128 && (c->sp_pieces == 2 || c->sp_pieces == 3);
    const int kp = segk_b3_kp(c->D);
    DISPATCH_XT(c, hipLaunchKernelGGL(k_kmeans_mark_dups<XT>, dim3(1), dim3(1024), 0, st,
                                       (const XT *)m->means, m->K_max, c->D, m->tiles, segk_tile_stride(c->D), segk_gmax(c->D),
                                       sp ? m->tiles_b3 : (float *)nullptr, sp ? segk_sp_tile_stride(c->D, c->sp_pieces) : 0,
                                       sp ? (kp / 16) * c->sp_pieces * 256 : 0, n_marked, ctx->row_hash););
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

int64_t segk_kmeans_tiles_b3_floats(int32_t K_max, int32_t D)
{
    return 1024 + (int64_t)segk_n_tiles(K_max) * segk_sp_tile_stride(D, 3);     // sized for either piece count
}

int64_t segk_corpus_b3_bytes(int64_t n_emb, int32_t D)
{
    return SEGK_SP_HEADER + n_emb * 3 * (int64_t)segk_b3_kp(D) * 2;             // sized for either piece count
}

int32_t segk_corpus_prepare_b3(segk_ctx *ctx, const segk_corpus *c, void *Xb3_out, int32_t pieces, void *stream)
{
    (void)ctx;
    int rc = check_corpus(c);
    if (rc) return rc;
    SEGK_REQUIRE(Xb3_out != nullptr, "Xb3_out is NULL");
    SEGK_REQUIRE(pieces == 2 || pieces == 3, "pieces must be 2 (fp16x2) or 3 (bf16x3)");
    SEGK_REQUIRE(c->x_dtype == SEGK_F32 && c->D >= 8 && c->D <= 128, "the split images exist for float32 data with 8 <= D <= 128");
    hipStream_t st = (hipStream_t)stream;
    SEGK_CHECK_HIP(hipMemsetAsync(Xb3_out, 0, SEGK_SP_HEADER, st));
    const int64_t tot = c->n_emb * segk_b3_kp(c->D);
    if (pieces == 2) {
        const int64_t nx = c->n_emb * c->D;
        const int64_t blocks = (nx + 255) / 256;
        hipLaunchKernelGGL(k_corpus_maxabs, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned int *)Xb3_out);
        hipLaunchKernelGGL(k_corpus_split_sp<2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
        hipLaunchKernelGGL(k_corpus_resid_sp, dim3((unsigned)((c->n_emb + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
    } else {
        hipLaunchKernelGGL(k_corpus_split_sp<3>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float *)c->X,
                           c->ldx, c->n_emb, c->D, (unsigned char *)Xb3_out);
    }
    SEGK_LAUNCH_CHECK();
    return SEGK_OK;
}

}  // extern "C"
