// Shared host-side plumbing of libsfm_amd.so: context, error reporting, device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/sfm_amd.h"

struct sfm_ctx {
    // Every object made on a context (image sets, results, batches, communicators, ...) holds a reference to
    // it and reads its stream when it is destroyed.  sfm_ctx_destroy drops the caller's reference; the streams
    // and events go when the last object has gone too, so handles may be destroyed in any order.
    int refs = 1;
    bool closed = false; // sfm_ctx_destroy has been called
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    // tunables
    int64_t cand_capacity = 524288;  // candidate slots per image pair
    int64_t pairs_per_launch = 2048; // image pairs per matcher chunk
    int match_two_phase = 1;         // 0: flagged pairs always take the filtered pass with running row bounds
    int match_flood_probe = 1;       // 0: pairs the int8 bound flooded always try the fp16 strict pass before the filtered ones
    int match_six_bit = 1;           // 0: the first bound pass of the float matcher is the int8 one (no 6-bit table is made)
    int match_dot_strict = 1;        // 0: the dot metric scales the bound by the norms cell by cell for every pair
    int match_pair_reuse = 0;        // 1: the single-pair matcher takes image A from the device when it is the last call's image B (same host addresses and sizes: the caller promises the tables have not changed)
    int cand_observe = 1;            // calls of several chunks size the candidate lists of later chunks from the first one
    int64_t hyps_per_launch = 0;     // rows of the first slice of an adaptive RANSAC batch (0 = sized to fill the device)
    int ransac_filter = 1;           // 0 disables the scoring kernel's epipolar pre-test
    int ransac_mfma = 1;             // 0: the pre-test of unsplit launches stays on the vector ALU (packed fp32)
    int ransac_defer_mib = 2048;     // buffer for the rows a scoring launch hands over to cell-by-cell evaluation (8 B per cell of such a row), MiB; 0: rows are never handed over
    int ransac_mfma_ops_gib = 64;    // largest table of prepared matrix-core operands (128 B per correspondence) a call may hold, GiB; 0: none
    int ransac_split = 1;            // 0: never cut a pair's correspondences across blocks
    int hamming_mfma = 1;            // 0: 256-bit binary descriptors take the popcount kernel only
    int match_col_split = 0;         // column parts per row block in the matcher's strict passes; 0 = chosen per call from the launch size
    int match_overlap = 1;           // 0: the matcher's chunks run one after the other on `stream` only
    int match_list_observe = 1;      // 0: a batch call allocates its match lists for the worst case (min(nA, nB) per pair) up front; 1: calls of more than 32 chunks size them as they go; 2: calls of more than 8 chunks do
    // Two more streams of the batch matcher and the events that order the three.  stream2: what follows a
    // chunk's first bound pass; stream3: the uploads of the next chunk's records and their bounds.  ev[0..2]
    // a chunk's first bound pass is done, ev[3..5] its buffer set is clear again, ev[7..9] its records are on
    // the device (all three by set), ev[6] fork / join.
    hipStream_t stream2 = nullptr, stream3 = nullptr;
#define SFM_CTX_EVENTS 16
    hipEvent_t ev[SFM_CTX_EVENTS] = {};
    int span_accumulate = 0;         // 1: sfm_ctx_last_kernel_ms sums over all calls since it was last read
    int sim_variant = 1;             // index into the similarity kernel's instance table (1 = int8 bound)
    // timing of the dominant kernels (HIP events on `stream`)
    struct Span { hipEvent_t a, b; };
    std::vector<Span> span_pool[2]; // events are created once and reused
    size_t span_used[2] = {0, 0};
    int n_cus = 256;
    // scale chain: work space kept between calls, and the size of its (step, track) table
    struct ChainScratch *chain = nullptr;
    sfm_ransac_batch *pair_batch = nullptr; // device buffers of the single-pair RANSAC entry points
    sfm_imageset *pair_set = nullptr;       // and of the single-pair matcher
    sfm_match_result *pair_result = nullptr;
    struct PairKeep *pair_keep = nullptr;   // image B of the last single-pair matcher call (option match_pair_reuse)
    int64_t chain_table_entries = (int64_t)1 << 26;
    // single-pair RANSAC: seed of the sample table the library draws when the caller passes none (0 = from the
    // clock, as the reference does, RANSAC.cpp:170), and the wall-clock cap of RANSAC_OPTIONS::maxTimeMs
    // (RANSAC.cpp:189-196), checked on the device between slices of sample rows (0 = none)
    uint64_t ransac_seed = 0;
    int64_t ransac_max_time_ms = 0;
    int64_t comm_timeout_ms = 120000; // longest wait for a multi-GPU exchange before SFM_ERR_TIMEOUT (0 = for ever)
};
void sfm_ctx_retain(sfm_ctx *ctx);
void sfm_ctx_release(sfm_ctx *ctx); // tears the context down when the last reference goes
void sfm_chain_scratch_free(sfm_ctx *ctx);
void sfm_pair_batch_free(sfm_ctx *ctx);
void sfm_pair_match_free(sfm_ctx *ctx);
// Per-device set-up of the RANSAC entry point that allocates nothing (address of its static workspace);
// called by sfm_ctx_create with the device current.
int sfm_ransac_static_setup(int device);
// Sample tables drawn on the device (the generator documented with sfm_corr_sample_table): offsets[n_pairs + 1]
// device, samples[n_pairs][n_hyp][8] device.
void sfm_launch_sample_table(const int64_t *offsets_device, int n_pairs, int n_hyp, uint64_t seed, int32_t *samples_device,
                             hipStream_t st);

#define SFM_FAIL(ctx, code, ...)                                   \
    do {                                                           \
        char _buf[512];                                            \
        snprintf(_buf, sizeof _buf, __VA_ARGS__);                  \
        if (ctx) (ctx)->last_error = _buf;                         \
        return (code);                                             \
    } while (0)

#define SFM_HIP(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            int _code = (_e == hipErrorOutOfMemory) ? SFM_ERR_OOM : SFM_ERR_HIP;        \
            SFM_FAIL(ctx, _code, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                               \
        }                                                                               \
    } while (0)

// Grow-only device buffer.  A `fixed` buffer is a slice of memory that somebody else owns (the static
// workspace of the entry point that allocates nothing): it never grows and is never freed.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool fixed = false;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (fixed) return hipErrorOutOfMemory;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p && !fixed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Pinned host staging buffer (grow-only).
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// A reader on another stream (the multi-GPU gather runs on a stream of the communicator's) marks the point up to
// which it reads an object's device buffers; the next call that overwrites them makes its stream wait for it.
struct ReadFence {
    hipEvent_t ev = nullptr;
    bool pending = false;
    hipError_t mark(hipStream_t reader)
    {
        if (!ev) {
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        hipError_t e = hipEventRecord(ev, reader);
        if (e == hipSuccess) pending = true;
        return e;
    }
    hipError_t wait_on(hipStream_t writer)
    {
        if (!pending) return hipSuccess;
        pending = false;
        return hipStreamWaitEvent(writer, ev, 0);
    }
    void destroy()
    {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        pending = false;
    }
};

// What the multi-GPU gather reads of a RANSAC batch after a run (device pointers; see sfm_ransac_gather).
struct RansacGatherView {
    sfm_ctx *ctx;
    int n_pairs;
    const int64_t *inlOff;  // [n_pairs + 1]
    const int32_t *inliers; // CSR, ascending per pair
    const void *packed;     // [n_pairs] records of SFM_RANSAC_RECORD_BYTES
    ReadFence *fence;
};
#define SFM_RANSAC_RECORD_BYTES 120 // {int32 best row, count, iterations, pad; double error sum; double pose12[12]}
int sfm_ransac_batch_gather_view(sfm_ransac_batch *b, RansacGatherView *v);

// Event-pair timing of one kernel launch (on ctx->stream unless a stream is given).
hipError_t sfm_span_begin(sfm_ctx *ctx, int which, hipStream_t st = nullptr);
hipError_t sfm_span_end(sfm_ctx *ctx, int which, hipStream_t st = nullptr);
void sfm_span_reset(sfm_ctx *ctx, int which);
int sfm_num_sim_variants();

// Device-side view of a RANSAC batch after its refit (read by the scale chain).
struct RefitView {
    int n_pairs;
    int64_t total;
    const int64_t *corrOffHost; // [n_pairs + 1], host
    const int64_t *corrOff;     // device from here on
    const int32_t *inlCount;
    const int64_t *inlOff;
    const int32_t *inliers;
    const double *pose16;
    const int32_t *used;
    const double *points;
};
int sfm_ransac_batch_refit_view(sfm_ransac_batch *b, RefitView *v, sfm_ctx **ctx_out);

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
