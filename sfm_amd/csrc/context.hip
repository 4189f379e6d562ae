// Context lifetime, options, error text and kernel timing spans.
#include "common.h"

extern "C" const char *sfm_version(void) { return "sfm_amd 0.1 (gfx950)"; }

extern "C" int sfm_ctx_create(int device, sfm_ctx **ctx_out)
{
    if (!ctx_out) return SFM_ERR_INVALID_ARG;
    *ctx_out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SFM_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SFM_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return SFM_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SFM_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "sfm_amd: device %d is %s; this library carries gfx950 code only\n", device,
                prop.gcnArchName);
        return SFM_ERR_NO_DEVICE;
    }
    sfm_ctx *c = new (std::nothrow) sfm_ctx();
    if (!c) return SFM_ERR_OOM;
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k < SFM_CTX_EVENTS; k++) ok = hipEventCreateWithFlags(&c->ev[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        sfm_ctx_destroy(c);
        return SFM_ERR_HIP;
    }
    const int rc = sfm_ransac_static_setup(device);
    if (rc != SFM_OK) {
        sfm_ctx_destroy(c);
        return rc;
    }
    *ctx_out = c;
    return SFM_OK;
}

void sfm_ctx_retain(sfm_ctx *ctx) { ctx->refs++; }

static void ctx_teardown(sfm_ctx *ctx)
{
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    // events first: some were recorded on the streams that go next
    for (int k = 0; k < SFM_CTX_EVENTS; k++)
        if (ctx->ev[k]) (void)hipEventDestroy(ctx->ev[k]);
    for (int w = 0; w < 2; w++)
        for (auto &s : ctx->span_pool[w]) {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    delete ctx;
}

void sfm_ctx_release(sfm_ctx *ctx)
{
    if (--ctx->refs == 0) ctx_teardown(ctx);
}

// The caller is done with the context.  Objects made on it that are still alive keep its streams alive
// (they synchronise on them when they are destroyed) and may still be read; the context goes with the last
// of them.  A second sfm_ctx_destroy of the same handle is an error of the caller, as free() twice is.
extern "C" void sfm_ctx_destroy(sfm_ctx *ctx)
{
    if (!ctx || ctx->closed) return;
    ctx->closed = true;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    // the objects the context made for its own use (each holds a reference)
    sfm_chain_scratch_free(ctx);
    sfm_pair_batch_free(ctx);
    sfm_pair_match_free(ctx);
    sfm_ctx_release(ctx);
}

extern "C" int sfm_ctx_synchronize(sfm_ctx *ctx)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" const char *sfm_last_error(const sfm_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

extern "C" int sfm_ctx_set_option(sfm_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key) return SFM_ERR_INVALID_ARG;
    if (!strcmp(key, "cand_capacity")) {
        if (value < 0 || value > 0x7FFFFFFF) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "cand_capacity must be in [0, 2^31)");
        ctx->cand_capacity = value;
    } else if (!strcmp(key, "pairs_per_launch")) {
        // (a chunk's pairs are one grid dimension of several kernels)
        if (value < 1 || value > 65535) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "pairs_per_launch must be in [1, 65535]");
        ctx->pairs_per_launch = value;
    } else if (!strcmp(key, "hyps_per_launch")) {
        if (value < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hyps_per_launch must be >= 0");
        ctx->hyps_per_launch = value;
    } else if (!strcmp(key, "chain_table_entries")) {
        if (value < 2) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "chain_table_entries must be >= 2");
        ctx->chain_table_entries = value;
    } else if (!strcmp(key, "hamming_mfma")) {
        ctx->hamming_mfma = value != 0;
    } else if (!strcmp(key, "span_accumulate")) {
        ctx->span_accumulate = value != 0;
        ctx->span_used[0] = ctx->span_used[1] = 0;
    } else if (!strcmp(key, "match_two_phase")) {
        ctx->match_two_phase = value != 0;
    } else if (!strcmp(key, "match_flood_probe")) {
        ctx->match_flood_probe = value != 0;
    } else if (!strcmp(key, "match_six_bit")) {
        ctx->match_six_bit = value != 0;
    } else if (!strcmp(key, "match_dot_strict")) {
        ctx->match_dot_strict = value != 0;
    } else if (!strcmp(key, "match_pair_reuse")) {
        ctx->match_pair_reuse = value != 0;
    } else if (!strcmp(key, "cand_observe")) {
        ctx->cand_observe = value != 0;
    } else if (!strcmp(key, "match_list_observe")) {
        if (value < 0 || value > 2) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match_list_observe must be 0, 1 or 2");
        ctx->match_list_observe = (int)value;
    } else if (!strcmp(key, "match_overlap")) {
        ctx->match_overlap = value != 0;
    } else if (!strcmp(key, "match_col_split")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16)
            SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match_col_split must be 0 (chosen per call), 1, 2, 4, 8 or 16");
        ctx->match_col_split = (int)value;
    } else if (!strcmp(key, "ransac_split")) {
        ctx->ransac_split = value != 0;
    } else if (!strcmp(key, "ransac_filter")) {
        ctx->ransac_filter = value != 0;
    } else if (!strcmp(key, "ransac_mfma")) {
        ctx->ransac_mfma = value != 0;
    } else if (!strcmp(key, "ransac_defer_mib")) {
        if (value < 0 || value > (1 << 20)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac_defer_mib must be in [0, 1048576]");
        ctx->ransac_defer_mib = (int)value;
    } else if (!strcmp(key, "ransac_mfma_ops_gib")) {
        if (value < 0 || value > 4096) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac_mfma_ops_gib must be in [0, 4096]");
        ctx->ransac_mfma_ops_gib = (int)value;
    } else if (!strcmp(key, "ransac_seed")) {
        ctx->ransac_seed = (uint64_t)value;
    } else if (!strcmp(key, "comm_timeout_ms")) {
        if (value < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "comm_timeout_ms must be >= 0");
        ctx->comm_timeout_ms = value;
    } else if (!strcmp(key, "ransac_max_time_ms")) {
        if (value < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "ransac_max_time_ms must be >= 0");
        ctx->ransac_max_time_ms = value;
    } else if (!strcmp(key, "sim_variant")) {
        if (value < 0 || value >= sfm_num_sim_variants())
            SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "sim_variant must be in [0, %d)", sfm_num_sim_variants());
        ctx->sim_variant = (int)value;
    } else {
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "unknown option '%s'", key);
    }
    return SFM_OK;
}

void sfm_span_reset(sfm_ctx *ctx, int which)
{
    if (!ctx->span_accumulate) ctx->span_used[which] = 0;
}

hipError_t sfm_span_begin(sfm_ctx *ctx, int which, hipStream_t st)
{
    if (ctx->span_used[which] == ctx->span_pool[which].size()) {
        sfm_ctx::Span s;
        hipError_t r = hipEventCreate(&s.a);
        if (r != hipSuccess) return r;
        r = hipEventCreate(&s.b);
        if (r != hipSuccess) return r;
        ctx->span_pool[which].push_back(s);
    }
    return hipEventRecord(ctx->span_pool[which][ctx->span_used[which]].a, st ? st : ctx->stream);
}

hipError_t sfm_span_end(sfm_ctx *ctx, int which, hipStream_t st)
{
    return hipEventRecord(ctx->span_pool[which][ctx->span_used[which]++].b, st ? st : ctx->stream);
}

extern "C" int sfm_ctx_last_kernel_ms(sfm_ctx *ctx, int which, double *ms_out, int *launches_out)
{
    if (!ctx || which < 0 || which > 1) return SFM_ERR_INVALID_ARG;
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0;
    for (size_t k = 0; k < ctx->span_used[which]; k++) {
        float ms = 0;
        SFM_HIP(ctx, hipEventElapsedTime(&ms, ctx->span_pool[which][k].a, ctx->span_pool[which][k].b));
        total += ms;
    }
    if (ms_out) *ms_out = total;
    if (launches_out) *launches_out = (int)ctx->span_used[which];
    if (ctx->span_accumulate) ctx->span_used[which] = 0; // read: the next calls start a new sum
    return SFM_OK;
}

// Page-locked host memory for the caller's tables: the device reads it directly, so sfm_match_pair and
// sfm_ransac_pair upload from it at the link's rate (the runtime pins pageable memory piece by piece, at
// about half that rate).
extern "C" int sfm_host_alloc(sfm_ctx *ctx, size_t bytes, void **ptr_out)
{
    if (!ctx || !ptr_out) return SFM_ERR_INVALID_ARG;
    *ptr_out = nullptr;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) SFM_FAIL(ctx, SFM_ERR_OOM, "host_alloc: %zu bytes of page-locked memory: %s", bytes, hipGetErrorString(e));
    *ptr_out = p;
    return SFM_OK;
}

extern "C" void sfm_host_free(void *ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}
