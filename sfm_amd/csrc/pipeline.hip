// Device-resident hand-off from matching to RANSAC.
//
// The reference assembles the correspondences of a frame pair on the host: for every shared
// observation it pushes normalizePoints(o1), normalizePoints(o2) (submodules/scene.cpp:216-252,
// normalizePoints scene.cpp:495-501: the first two components of K^-1 (u, v, 1)) and hands the two
// vectors to solve::RANSAC (scene.cpp:261).  Here the match lists stay on the device: one kernel
// gathers the matched pixel positions and normalises them into the CSR layout the batched RANSAC
// entry points read, and a second one draws the sample table (the reference's sampling is a
// time-seeded shuffle, RANSAC.cpp:170,201; any uniform draw of 8 distinct indices is a valid one).
#include "match_types.h"

#pragma clang fp contract(off)

typedef unsigned long long u64;

struct KinvArg {
    double m[9]; // column-major
};

// One block per pair.  x[k] = (Kinv (uA, vA, 1))_{0,1}, y[k] likewise for image B, k over the pair's
// match list in its order (ascending row of image A).
__global__ __launch_bounds__(256) void corr_gather_kernel(const PairRec *__restrict__ recs,
                                                          const int64_t *__restrict__ offsets,
                                                          const int2 *__restrict__ pairs,
                                                          const double2 *__restrict__ pts, const KinvArg Ki,
                                                          double2 *__restrict__ x, double2 *__restrict__ y)
{
    const PairRec pr = recs[blockIdx.x];
    const int64_t lo = offsets[blockIdx.x], hi = offsets[blockIdx.x + 1];
    for (int64_t k = lo + threadIdx.x; k < hi; k += blockDim.x) {
        const int2 m = pairs[k];
        const double2 a = pts[pr.rowA32 + m.x], b = pts[pr.rowB32 + m.y];
        // scene.cpp:499: ray = K_inv * (u, v, 1); the product by 1 is exact
        x[k] = make_double2(__dadd_rn(__dadd_rn(__dmul_rn(Ki.m[0], a.x), __dmul_rn(Ki.m[3], a.y)), Ki.m[6]),
                            __dadd_rn(__dadd_rn(__dmul_rn(Ki.m[1], a.x), __dmul_rn(Ki.m[4], a.y)), Ki.m[7]));
        y[k] = make_double2(__dadd_rn(__dadd_rn(__dmul_rn(Ki.m[0], b.x), __dmul_rn(Ki.m[3], b.y)), Ki.m[6]),
                            __dadd_rn(__dadd_rn(__dmul_rn(Ki.m[1], b.x), __dmul_rn(Ki.m[4], b.y)), Ki.m[7]));
    }
}

// splitmix64 finaliser
__host__ __device__ inline u64 mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Row h of pair p: 8 distinct indices of range(n), uniform.  Draw k picks position
// floor(r_k * (n - k) / 2^64) among the indices not taken yet, r_k = mix64(s + (k + 1) * golden),
// s = mix64(seed ^ mix64(p * 2^32 + h)).  Pairs with n < 8 get rows of zeros (never read).
__global__ __launch_bounds__(256) void sample_table_kernel(const int64_t *__restrict__ offsets, int n_hyp, u64 seed,
                                                           int32_t *__restrict__ samples)
{
    const int p = blockIdx.y;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hyp) return;
    const int64_t n = offsets[p + 1] - offsets[p];
    int32_t pick[8];
    if (n < 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) pick[k] = 0;
    } else {
        const u64 s = mix64(seed ^ mix64(((u64)p << 32) | (u64)(unsigned)h));
        int32_t sorted[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u64 r = mix64(s + (u64)(k + 1) * 0x9E3779B97F4A7C15ull);
            int32_t v = (int32_t)__umul64hi(r, (u64)(n - k)); // position among the free indices
            // step over the indices already taken, in ascending order
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (t < k && sorted[t] <= v) v++;
            pick[k] = v;
            // insert v into sorted[0..k)
            int32_t carry = v;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                if (t < k && sorted[t] > carry) {
                    const int32_t tmp = sorted[t];
                    sorted[t] = carry;
                    carry = tmp;
                }
            }
            sorted[k] = carry;
        }
    }
    int32_t *dst = samples + ((size_t)p * n_hyp + h) * 8;
    *reinterpret_cast<int4 *>(dst) = make_int4(pick[0], pick[1], pick[2], pick[3]);
    *reinterpret_cast<int4 *>(dst + 4) = make_int4(pick[4], pick[5], pick[6], pick[7]);
}

void sfm_launch_sample_table(const int64_t *offsets_device, int n_pairs, int n_hyp, uint64_t seed, int32_t *samples_device,
                             hipStream_t st)
{
    if (n_pairs <= 0 || n_hyp <= 0) return;
    hipLaunchKernelGGL(sample_table_kernel, dim3((n_hyp + 255) / 256, n_pairs), dim3(256), 0, st, offsets_device, n_hyp,
                       (u64)seed, samples_device);
}

struct sfm_corr {
    sfm_ctx *ctx = nullptr;
    int n_pairs = 0;
    int n_hyp = 0;
    int64_t total = 0;
    std::vector<int64_t> offsets; // host copy, n_pairs + 1
    DevBuf offsetsDev, x, y, samples;
};

extern "C" void sfm_corr_destroy(sfm_corr *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    c->offsetsDev.release();
    c->x.release();
    c->y.release();
    c->samples.release();
    sfm_ctx *owner = c->ctx;
    delete c;
    sfm_ctx_release(owner);
}

extern "C" int sfm_corr_from_matches(sfm_ctx *ctx, const sfm_imageset *set, sfm_match_result *res,
                                     const double *Kinv, sfm_corr **corr_inout)
{
    if (!ctx || !set || !res || !corr_inout) return SFM_ERR_INVALID_ARG;
    if (!Kinv) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "correspondences: null inverse intrinsics");
    if (set->ctx != ctx || res->ctx != ctx)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "correspondences: image set or match result belongs to another context");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_corr *c = *corr_inout;
    if (!c) {
        c = new (std::nothrow) sfm_corr();
        if (!c) SFM_FAIL(ctx, SFM_ERR_OOM, "correspondences: host allocation failed");
        sfm_ctx_retain(ctx);
        c->ctx = ctx;
        *corr_inout = c;
    } else if (c->ctx != ctx) {
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "correspondences: object belongs to another context");
    }
    const int P = res->n_pairs;
    c->n_pairs = P;
    c->n_hyp = 0;
    c->offsets.assign((size_t)P + 1, 0);
    hipStream_t st = ctx->stream;
    SFM_HIP(ctx, hipMemcpyAsync(c->offsets.data(), res->offsets.p, (size_t)(P + 1) * sizeof(int64_t),
                                hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    c->total = c->offsets[P];
    SFM_HIP(ctx, c->x.reserve((size_t)(c->total ? c->total : 1) * sizeof(double2)));
    SFM_HIP(ctx, c->y.reserve((size_t)(c->total ? c->total : 1) * sizeof(double2)));
    if (P > 0 && c->total > 0) {
        KinvArg Ki;
        for (int i = 0; i < 9; i++) Ki.m[i] = Kinv[i];
        hipLaunchKernelGGL(corr_gather_kernel, dim3(P), dim3(256), 0, st, res->pairRecs.as<PairRec>(),
                           res->offsets.as<int64_t>(), res->pairs.as<int2>(), set->pts.as<double2>(), Ki,
                           c->x.as<double2>(), c->y.as<double2>());
        SFM_HIP(ctx, hipGetLastError());
    }
    return SFM_OK;
}

extern "C" int sfm_corr_offsets(sfm_corr *c, int64_t *offsets_out)
{
    if (!c || !offsets_out) return SFM_ERR_INVALID_ARG;
    memcpy(offsets_out, c->offsets.data(), c->offsets.size() * sizeof(int64_t));
    return SFM_OK;
}

extern "C" int sfm_corr_device(sfm_corr *c, const double **x_device, const double **y_device)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    if (x_device) *x_device = c->x.as<double>();
    if (y_device) *y_device = c->y.as<double>();
    return SFM_OK;
}

extern "C" int sfm_corr_fetch(sfm_corr *c, double *x_out, double *y_out)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = c->ctx;
    if (c->total == 0) return SFM_OK;
    if (!x_out || !y_out) return SFM_ERR_INVALID_ARG;
    SFM_HIP(ctx, hipMemcpyAsync(x_out, c->x.p, (size_t)c->total * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipMemcpyAsync(y_out, c->y.p, (size_t)c->total * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_corr_sample_table(sfm_corr *c, int n_hyp, uint64_t seed, const int32_t **samples_device,
                                     int32_t *samples_out)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = c->ctx;
    if (n_hyp < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "sample table: negative row count");
    if (c->n_pairs > 65535) SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "sample table: more than 65535 pairs per call");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cells = (size_t)c->n_pairs * (size_t)n_hyp * 8;
    SFM_HIP(ctx, c->samples.reserve((cells ? cells : 1) * sizeof(int32_t)));
    SFM_HIP(ctx, c->offsetsDev.reserve(c->offsets.size() * sizeof(int64_t)));
    c->n_hyp = n_hyp;
    if (cells > 0) {
        SFM_HIP(ctx, hipMemcpyAsync(c->offsetsDev.p, c->offsets.data(), c->offsets.size() * sizeof(int64_t),
                                    hipMemcpyHostToDevice, ctx->stream));
        sfm_launch_sample_table(c->offsetsDev.as<int64_t>(), c->n_pairs, n_hyp, seed, c->samples.as<int32_t>(), ctx->stream);
        SFM_HIP(ctx, hipGetLastError());
        if (samples_out) {
            SFM_HIP(ctx, hipMemcpyAsync(samples_out, c->samples.p, cells * sizeof(int32_t), hipMemcpyDeviceToHost,
                                        ctx->stream));
            SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if (samples_device) *samples_device = c->samples.as<int32_t>();
    return SFM_OK;
}

extern "C" int sfm_ransac_batch_run_corr(sfm_ctx *ctx, sfm_corr *c, const double *K, int n_hyp, uint64_t seed,
                                         double max_squared_error, double success_prob, int adaptive, int loss,
                                         sfm_ransac_batch **batch_inout)
{
    if (!ctx || !c || !batch_inout) return SFM_ERR_INVALID_ARG;
    if (c->ctx != ctx) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac: correspondences belong to another context");
    const int32_t *samples = nullptr;
    int rc = sfm_corr_sample_table(c, n_hyp, seed, &samples, nullptr);
    if (rc != SFM_OK) return rc;
    return sfm_ransac_batch_run_device(ctx, c->n_pairs, c->offsets.data(), c->x.as<double>(), c->y.as<double>(), K,
                                       samples, n_hyp, max_squared_error, success_prob, adaptive, loss, batch_inout);
}
