// Multi-GPU layer of libsfm_amd.so: static partition of an image-pair list over the GPUs of one node
// and the gather of the per-GPU match lists to one of them with RCCL over xGMI.
//
// The reference is a single process that matches one pair per call (submodules/scene.cpp:60); pairs
// read only their own two images, so a pair list shards with no exchange inside the data path.  The one
// exchange is the gather of results.  Match lists have variable length, so it runs in two steps:
//   1. ncclAllGather of the per-pair match counts (every rank pads its counts to the largest share) and
//      of the per-rank totals;
//   2. the lists themselves, point to point: every other rank ncclSend's its list to the root, the root
//      posts all ncclRecv's in one group, each straight into its place in one rank-major buffer.  On an
//      8-GPU MI355X node every peer has its own xGMI link to the root, so the transfers run side by side
//      (no ring).
// The gather runs on a stream that the communicator owns and waits only for the batch call whose result it
// moves: sfm_gathered_wait (or any accessor of the lists) completes it.
// Two ways to drive it: one process per GPU (sfm_comm_create from a broadcast id; sfm_match_gather), or
// one process that owns several devices (sfm_comm_create_local; sfm_match_gather_local /
// sfm_match_batch_multi), where the calls of all ranks are issued inside one RCCL group.
#include "common.h"
#include "match_types.h"

#include <rccl/rccl.h>

#include <chrono>
#include <thread>

#define SHARD_BLOCK 64 // consecutive pairs dealt to one rank (they share an image in an exhaustive list)

// The gather runs on a stream of the communicator's own: it waits for the event that marks the end of the
// batch call whose lists it moves and for nothing enqueued on the context's stream after that, so a caller
// can start the next batch (into another result object) before it gathers the previous one.
struct sfm_comm {
    sfm_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr; // for results that carry no event of their own
    int refs = 1;            // the caller's handle, and one per gathered object that uses this communicator
    bool closed = false;     // sfm_comm_destroy has been called
};

static int comm_streams(sfm_comm *c)
{
    sfm_ctx *ctx = c->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    SFM_HIP(ctx, hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    return SFM_OK;
}

struct sfm_gathered {
    sfm_comm *comm = nullptr;
    int64_t n_total = 0;   // pairs of the whole list
    int64_t max_owned = 0; // largest share of a rank
    int root = 0;
    int elemBytes = 8;     // bytes per list entry: 8 for match pairs (i, j), 4 for inlier indices
    size_t recBytes = 0;   // bytes of the fixed record that travels with every pair (RANSAC winners), 0 for none
    DevBuf countsMine, countsAll, totalsMine, totalsAll, payload, records;
    std::vector<int64_t> totals;  // per rank, host (valid on every rank)
    std::vector<int32_t> countsHost; // [world][max_owned], host, filled on demand
    bool countsFetched = false;
    int64_t total = 0; // sum of totals
};

// One rank's lists as the gather reads them: CSR offsets and entries on the device, optionally a fixed record
// per pair, the event that marks them complete (or NULL: whatever the context's stream holds by now), and the
// fence through which the owner's next call learns how long they are being read.
struct GatherSource {
    sfm_ctx *ctx;
    int64_t n_pairs;
    const int64_t *offsets;
    const void *lists;
    int elemBytes;
    const void *records;
    size_t recBytes;
    hipEvent_t done;
    ReadFence *fence;
};

#define SFM_NCCL(ctx, expr)                                                                       \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess)                                                                    \
            SFM_FAIL(ctx, SFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

// Waits for a stream of the communicator with the context's time limit (option "comm_timeout_ms", 0 = for
// ever): a collective that a peer never joins shows as SFM_ERR_TIMEOUT instead of a process that hangs.
static int stream_wait(sfm_ctx *ctx, hipStream_t st, int64_t timeout_ms)
{
    if (timeout_ms <= 0) {
        SFM_HIP(ctx, hipStreamSynchronize(st));
        return SFM_OK;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (int spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) return SFM_OK;
        if (e != hipErrorNotReady) SFM_FAIL(ctx, SFM_ERR_HIP, "gather: the communicator's stream failed: %s", hipGetErrorString(e));
        if (std::chrono::steady_clock::now() > deadline)
            SFM_FAIL(ctx, SFM_ERR_TIMEOUT, "gather: not finished after %lld ms (a rank that never joined the collective?)", (long long)timeout_ms);
        if (spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// ---- static partition -------------------------------------------------------------------------
// Blocks of SHARD_BLOCK consecutive pairs are dealt round-robin: pair i belongs to rank
// (i / SHARD_BLOCK) % world.  Shares differ by at most one block.
extern "C" int64_t sfm_shard_count(int64_t n_pairs, int rank, int world)
{
    if (n_pairs <= 0 || world <= 0 || rank < 0 || rank >= world) return 0;
    const int64_t blocks = (n_pairs + SHARD_BLOCK - 1) / SHARD_BLOCK;
    const int64_t mine = blocks / world + ((blocks % world) > rank ? 1 : 0); // blocks of this rank
    int64_t n = mine * SHARD_BLOCK;
    if (mine > 0 && (blocks - 1) % world == rank) n -= blocks * SHARD_BLOCK - n_pairs; // the last block may be short
    return n;
}

extern "C" int sfm_shard_indices(int64_t n_pairs, int rank, int world, int64_t *indices_out)
{
    if (n_pairs < 0 || world <= 0 || rank < 0 || rank >= world || (!indices_out && n_pairs > 0)) return SFM_ERR_INVALID_ARG;
    int64_t k = 0;
    for (int64_t b = rank; b * SHARD_BLOCK < n_pairs; b += world)
        for (int64_t i = b * SHARD_BLOCK; i < std::min<int64_t>(n_pairs, (b + 1) * SHARD_BLOCK); i++) indices_out[k++] = i;
    return SFM_OK;
}

// Where the gathered lists lie: the buffer holds rank 0's lists, then rank 1's, ..., each in the order of
// the rank's own pairs.  counts[world][max_owned] are the gathered per-pair counts.  starts_out /
// counts_out (either may be NULL) are indexed by the pair's position in the whole list.
extern "C" int sfm_shard_layout(int64_t n_pairs, int world, const int32_t *counts, int64_t max_owned, int64_t *starts_out,
                                int32_t *counts_out)
{
    if (n_pairs < 0 || world <= 0 || (n_pairs > 0 && !counts)) return SFM_ERR_INVALID_ARG;
    int64_t run = 0;
    for (int r = 0; r < world; r++) {
        int64_t k = 0;
        for (int64_t b = r; b * SHARD_BLOCK < n_pairs; b += world)
            for (int64_t i = b * SHARD_BLOCK; i < std::min<int64_t>(n_pairs, (b + 1) * SHARD_BLOCK); i++, k++) {
                if (k >= max_owned) return SFM_ERR_INVALID_ARG;
                const int32_t c = counts[(size_t)r * max_owned + k];
                if (starts_out) starts_out[i] = run;
                if (counts_out) counts_out[i] = c;
                run += c;
            }
    }
    return SFM_OK;
}

// The rank-major buffers of a RANSAC gather, put into the order of the whole pair list (host only): records
// [world][max_owned] of SFM_RANSAC_RECORD_BYTES, counts [world][max_owned], lists = rank 0's inlier lists, then
// rank 1's, ...  Every output may be NULL.
extern "C" int sfm_shard_unpack_ransac(int64_t n_pairs, int world, const int32_t *counts, int64_t max_owned, const void *records,
                                       const int32_t *lists, int32_t *best_hyp_out, int32_t *best_count_out, double *best_toterr_out,
                                       int32_t *iterations_out, double *pose12_out, int64_t *inlier_offsets_out, int32_t *inliers_out)
{
    if (n_pairs < 0 || world <= 0 || (n_pairs > 0 && (!counts || !records))) return SFM_ERR_INVALID_ARG;
    std::vector<int64_t> starts((size_t)std::max<int64_t>(n_pairs, 1));
    std::vector<int32_t> cnt((size_t)std::max<int64_t>(n_pairs, 1));
    const int rc = sfm_shard_layout(n_pairs, world, counts, max_owned, starts.data(), cnt.data());
    if (rc != SFM_OK) return rc;
    int64_t run = 0;
    std::vector<int64_t> k((size_t)world, 0); // next pair of every rank
    for (int64_t i = 0; i < n_pairs; i++) {
        const int r = (int)((i / SHARD_BLOCK) % world);
        const char *src = (const char *)records + ((size_t)r * (size_t)max_owned + (size_t)k[r]++) * SFM_RANSAC_RECORD_BYTES;
        int32_t head[4];
        double err, pose[12];
        memcpy(head, src, sizeof head);
        memcpy(&err, src + 16, sizeof err);
        memcpy(pose, src + 24, sizeof pose);
        if (best_hyp_out) best_hyp_out[i] = head[0];
        if (best_count_out) best_count_out[i] = head[1];
        if (iterations_out) iterations_out[i] = head[2];
        if (best_toterr_out) best_toterr_out[i] = err;
        if (pose12_out) memcpy(pose12_out + 12 * i, pose, sizeof pose);
        if (inlier_offsets_out) inlier_offsets_out[i] = run;
        if (inliers_out && cnt[i] > 0) {
            if (!lists) return SFM_ERR_INVALID_ARG;
            memcpy(inliers_out + run, lists + starts[i], (size_t)cnt[i] * sizeof(int32_t));
        }
        run += cnt[i];
    }
    if (inlier_offsets_out) inlier_offsets_out[n_pairs] = run;
    return SFM_OK;
}

// ---- communicators ------------------------------------------------------------------------------
extern "C" int sfm_comm_unique_id(void *id128_out)
{
    if (!id128_out) return SFM_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "the id crosses the C-ABI as 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SFM_ERR_HIP;
    memcpy(id128_out, &id, sizeof id);
    return SFM_OK;
}

extern "C" int sfm_comm_create(sfm_ctx *ctx, int rank, int world, const void *id128, sfm_comm **comm_out)
{
    if (!ctx || !comm_out || !id128 || world < 1 || rank < 0 || rank >= world) return SFM_ERR_INVALID_ARG;
    *comm_out = nullptr;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_comm *c = new (std::nothrow) sfm_comm();
    if (!c) SFM_FAIL(ctx, SFM_ERR_OOM, "comm: host allocation failed");
    sfm_ctx_retain(ctx);
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        sfm_ctx_release(ctx);
        SFM_FAIL(ctx, SFM_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, ncclGetErrorString(r));
    }
    const int rc = comm_streams(c);
    if (rc != SFM_OK) {
        sfm_comm_destroy(c);
        return rc;
    }
    *comm_out = c;
    return SFM_OK;
}

extern "C" int sfm_comm_create_local(int n, sfm_ctx *const *ctxs, sfm_comm **comms_out)
{
    if (n < 1 || !ctxs || !comms_out) return SFM_ERR_INVALID_ARG;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return SFM_ERR_INVALID_ARG;
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) SFM_FAIL(ctxs[0], SFM_ERR_INVALID_ARG, "comm: device %d is listed twice", devs[i]);
        comms_out[i] = nullptr;
    }
    SFM_NCCL(ctxs[0], ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) {
        sfm_comm *c = new (std::nothrow) sfm_comm();
        if (!c) {
            for (int j = 0; j < n; j++) {
                if (j < i) {
                    delete comms_out[j];
                    sfm_ctx_release(ctxs[j]);
                }
                (void)ncclCommDestroy(comms[j]);
            }
            SFM_FAIL(ctxs[0], SFM_ERR_OOM, "comm: host allocation failed");
        }
        sfm_ctx_retain(ctxs[i]);
        c->ctx = ctxs[i];
        c->comm = comms[i];
        c->rank = i;
        c->world = n;
        comms_out[i] = c;
    }
    for (int i = 0; i < n; i++) {
        const int rc = comm_streams(comms_out[i]);
        if (rc != SFM_OK) { // none of the n is handed out
            for (int j = 0; j < n; j++) {
                sfm_comm_destroy(comms_out[j]);
                comms_out[j] = nullptr;
            }
            return rc;
        }
    }
    return SFM_OK;
}

static void comm_release(sfm_comm *c);

// Gathered objects made with the communicator keep it (and its stream) alive until they are destroyed too.
extern "C" void sfm_comm_destroy(sfm_comm *c)
{
    if (!c || c->closed) return;
    c->closed = true;
    comm_release(c);
}

static void comm_release(sfm_comm *c)
{
    if (--c->refs > 0) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    sfm_ctx *owner = c->ctx;
    delete c;
    sfm_ctx_release(owner);
}

extern "C" int sfm_comm_rank(const sfm_comm *c, int *rank_out, int *world_out)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    if (rank_out) *rank_out = c->rank;
    if (world_out) *world_out = c->world;
    return SFM_OK;
}

// ---- gather ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_counts_kernel(const int64_t *__restrict__ off, int64_t nOwned, int64_t padded,
                                                            int32_t *__restrict__ counts, int64_t *__restrict__ total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < padded) counts[i] = i < nOwned ? (int32_t)(off[i + 1] - off[i]) : 0;
    if (i == 0) *total = nOwned > 0 ? off[nOwned] : 0;
}

// Before anything is enqueued: the arguments of one rank are checked and its buffers made.  Everything that
// can fail without a peer happens here, so that a group of collectives is only opened when every rank of the
// process will go through with it.
static int gather_prepare(sfm_comm *c, const GatherSource &src, int64_t n_total, int root, sfm_gathered **g_inout)
{
    sfm_ctx *ctx = c->ctx;
    if (!g_inout || n_total < 0 || root < 0 || root >= c->world) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: bad argument");
    if (src.ctx != ctx) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: the result belongs to another context");
    const int64_t nOwned = sfm_shard_count(n_total, c->rank, c->world);
    if (src.n_pairs != nOwned)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: rank %d of %d owns %lld of %lld pairs, its result holds %lld", c->rank, c->world,
                 (long long)nOwned, (long long)n_total, (long long)src.n_pairs);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_gathered *g = *g_inout;
    if (!g) {
        g = new (std::nothrow) sfm_gathered();
        if (!g) SFM_FAIL(ctx, SFM_ERR_OOM, "gather: host allocation failed");
        *g_inout = g;
    }
    if (g->comm != c) {
        if (g->comm) comm_release(g->comm);
        c->refs++;
    }
    g->comm = c;
    g->n_total = n_total;
    g->root = root;
    g->elemBytes = src.elemBytes;
    g->recBytes = src.recBytes;
    g->countsFetched = false;
    g->total = 0;
    g->max_owned = 0;
    for (int r = 0; r < c->world; r++) g->max_owned = std::max(g->max_owned, sfm_shard_count(n_total, r, c->world));
    const size_t padded = (size_t)std::max<int64_t>(g->max_owned, 1);
    SFM_HIP(ctx, g->countsMine.reserve(padded * sizeof(int32_t)));
    SFM_HIP(ctx, g->countsAll.reserve(padded * c->world * sizeof(int32_t)));
    SFM_HIP(ctx, g->totalsMine.reserve(sizeof(int64_t)));
    SFM_HIP(ctx, g->totalsAll.reserve((size_t)c->world * sizeof(int64_t)));
    if (src.recBytes && c->rank == root) SFM_HIP(ctx, g->records.reserve(padded * c->world * src.recBytes));
    g->totals.assign(c->world, 0);
    return SFM_OK;
}

// Step 1 for one rank: counts and total of the local lists, and their all-gathers (enqueued; the caller
// brackets the calls of all its ranks in one RCCL group).
static int gather_enqueue_counts(sfm_gathered *g, const GatherSource &src)
{
    sfm_comm *c = g->comm;
    sfm_ctx *ctx = c->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t padded = (size_t)std::max<int64_t>(g->max_owned, 1);
    if (src.done) {
        SFM_HIP(ctx, hipStreamWaitEvent(c->stream, src.done, 0));
    } else { // whatever the context's stream holds by now
        SFM_HIP(ctx, hipEventRecord(c->ev, ctx->stream));
        SFM_HIP(ctx, hipStreamWaitEvent(c->stream, c->ev, 0));
    }
    hipLaunchKernelGGL(gather_counts_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, c->stream, src.offsets, src.n_pairs,
                       (int64_t)padded, g->countsMine.as<int32_t>(), g->totalsMine.as<int64_t>());
    SFM_HIP(ctx, hipGetLastError());
    SFM_NCCL(ctx, ncclAllGather(g->countsMine.p, g->countsAll.p, padded, ncclInt32, c->comm, c->stream));
    SFM_NCCL(ctx, ncclAllGather(g->totalsMine.p, g->totalsAll.p, 1, ncclInt64, c->comm, c->stream));
    return SFM_OK;
}

// Between the steps: the totals on the host (the root sizes its buffer and its receives from them).
static int gather_fetch_totals(sfm_gathered *g)
{
    sfm_comm *c = g->comm;
    sfm_ctx *ctx = c->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemcpyAsync(g->totals.data(), g->totalsAll.p, (size_t)c->world * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    const int rc = stream_wait(ctx, c->stream, ctx->comm_timeout_ms);
    if (rc != SFM_OK) return rc;
    g->total = 0;
    for (int r = 0; r < c->world; r++) {
        if (g->totals[r] < 0) SFM_FAIL(ctx, SFM_ERR_HIP, "gather: rank %d reports a negative total", r);
        g->total += g->totals[r];
    }
    if (c->rank == g->root) SFM_HIP(ctx, g->payload.reserve((size_t)std::max<int64_t>(g->total, 1) * (size_t)g->elemBytes));
    return SFM_OK;
}

// Step 2 for one rank: the root receives every other rank's lists (and records) into their places, the others
// send.  The fence marks the point on the communicator's stream after which the local lists are no longer read.
static int gather_enqueue_payload(sfm_gathered *g, const GatherSource &src)
{
    sfm_comm *c = g->comm;
    sfm_ctx *ctx = c->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t eb = (size_t)g->elemBytes, rb = g->recBytes;
    const size_t padded = (size_t)std::max<int64_t>(g->max_owned, 1);
    if (c->rank == g->root) {
        int64_t base = 0;
        for (int r = 0; r < c->world; r++) {
            char *dst = g->payload.as<char>() + (size_t)base * eb;
            const size_t owned = (size_t)sfm_shard_count(g->n_total, r, c->world);
            char *rdst = rb ? g->records.as<char>() + (size_t)r * padded * rb : nullptr;
            if (r == c->rank) {
                if (g->totals[r] > 0)
                    SFM_HIP(ctx, hipMemcpyAsync(dst, src.lists, (size_t)g->totals[r] * eb, hipMemcpyDeviceToDevice, c->stream));
                if (rb && owned > 0) SFM_HIP(ctx, hipMemcpyAsync(rdst, src.records, owned * rb, hipMemcpyDeviceToDevice, c->stream));
            } else {
                if (g->totals[r] > 0) SFM_NCCL(ctx, ncclRecv(dst, (size_t)g->totals[r] * eb, ncclChar, r, c->comm, c->stream));
                if (rb && owned > 0) SFM_NCCL(ctx, ncclRecv(rdst, owned * rb, ncclChar, r, c->comm, c->stream));
            }
            base += g->totals[r];
        }
    } else {
        if (g->totals[c->rank] > 0)
            SFM_NCCL(ctx, ncclSend(src.lists, (size_t)g->totals[c->rank] * eb, ncclChar, g->root, c->comm, c->stream));
        if (rb && src.n_pairs > 0) SFM_NCCL(ctx, ncclSend(src.records, (size_t)src.n_pairs * rb, ncclChar, g->root, c->comm, c->stream));
    }
    return SFM_OK;
}

// The three steps for the n ranks this process drives (n = 1 for one process per GPU).
static int gather_run(int n, sfm_comm *const *comms, const GatherSource *srcs, int64_t n_total, int root, sfm_gathered **gathered_inout)
{
    sfm_ctx *ctx0 = comms[0]->ctx;
    int rc = SFM_OK;
    for (int i = 0; i < n; i++)
        if ((rc = gather_prepare(comms[i], srcs[i], n_total, root, &gathered_inout[i])) != SFM_OK) return rc;
    SFM_NCCL(ctx0, ncclGroupStart());
    for (int i = 0; i < n && rc == SFM_OK; i++) rc = gather_enqueue_counts(gathered_inout[i], srcs[i]);
    ncclResult_t e = ncclGroupEnd();
    if (rc != SFM_OK) return rc;
    SFM_NCCL(ctx0, e);
    for (int i = 0; i < n; i++)
        if ((rc = gather_fetch_totals(gathered_inout[i])) != SFM_OK) return rc;
    SFM_NCCL(ctx0, ncclGroupStart());
    for (int i = 0; i < n && rc == SFM_OK; i++) rc = gather_enqueue_payload(gathered_inout[i], srcs[i]);
    e = ncclGroupEnd();
    if (rc != SFM_OK) return rc;
    SFM_NCCL(ctx0, e);
    for (int i = 0; i < n; i++) // (after the group: inside it the transfers are not on the stream yet)
        if (srcs[i].fence) SFM_HIP(comms[i]->ctx, srcs[i].fence->mark(comms[i]->stream));
    return SFM_OK;
}

static int check_local_comms(int n, sfm_comm *const *comms)
{
    if (n < 1 || !comms || !comms[0]) return SFM_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++)
        if (!comms[i] || comms[i]->world != n || comms[i]->rank != i)
            SFM_FAIL(comms[0]->ctx, SFM_ERR_INVALID_ARG, "gather: comms must be the n ranks of one local communicator, in rank order");
    return SFM_OK;
}

static GatherSource match_source(sfm_match_result *r)
{
    GatherSource s;
    s.ctx = r->ctx;
    s.n_pairs = r->n_pairs;
    s.offsets = r->offsets.as<int64_t>();
    s.lists = r->pairs.p;
    s.elemBytes = (int)sizeof(int2);
    s.records = nullptr;
    s.recBytes = 0;
    s.done = r->done;
    s.fence = &r->readFence;
    return s;
}

extern "C" int sfm_match_gather(sfm_comm *c, sfm_match_result *local, int64_t n_pairs_total, int root, sfm_gathered **gathered_inout)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    if (!local || !gathered_inout) SFM_FAIL(c->ctx, SFM_ERR_INVALID_ARG, "gather: bad argument");
    const GatherSource src = match_source(local);
    return gather_run(1, &c, &src, n_pairs_total, root, gathered_inout);
}

extern "C" int sfm_match_gather_local(int n, sfm_comm *const *comms, sfm_match_result *const *locals, int64_t n_pairs_total, int root,
                                      sfm_gathered **gathered_inout)
{
    int rc = check_local_comms(n, comms);
    if (rc != SFM_OK) return rc;
    if (!locals || !gathered_inout) return SFM_ERR_INVALID_ARG;
    std::vector<GatherSource> srcs((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!locals[i]) SFM_FAIL(comms[0]->ctx, SFM_ERR_INVALID_ARG, "gather: rank %d has no result", i);
        srcs[i] = match_source(locals[i]);
    }
    return gather_run(n, comms, srcs.data(), n_pairs_total, root, gathered_inout);
}

// RANSAC results (the batch of this rank's share of the pair list, in sfm_shard_indices order): the winners'
// records travel as fixed-size records, the inlier lists as the variable-length part.
static int ransac_source(sfm_ransac_batch *b, GatherSource *s)
{
    RansacGatherView v;
    const int rc = sfm_ransac_batch_gather_view(b, &v);
    if (rc != SFM_OK) return rc;
    s->ctx = v.ctx;
    s->n_pairs = v.n_pairs;
    s->offsets = v.inlOff;
    s->lists = v.inliers;
    s->elemBytes = (int)sizeof(int32_t);
    s->records = v.packed;
    s->recBytes = SFM_RANSAC_RECORD_BYTES;
    s->done = nullptr; // a batch run ends on the context's stream
    s->fence = v.fence;
    return SFM_OK;
}

extern "C" int sfm_ransac_gather(sfm_comm *c, sfm_ransac_batch *local, int64_t n_pairs_total, int root, sfm_gathered **gathered_inout)
{
    if (!c) return SFM_ERR_INVALID_ARG;
    if (!local || !gathered_inout) SFM_FAIL(c->ctx, SFM_ERR_INVALID_ARG, "gather: bad argument");
    GatherSource src;
    const int rc = ransac_source(local, &src);
    if (rc != SFM_OK) return rc;
    return gather_run(1, &c, &src, n_pairs_total, root, gathered_inout);
}

extern "C" int sfm_ransac_gather_local(int n, sfm_comm *const *comms, sfm_ransac_batch *const *locals, int64_t n_pairs_total, int root,
                                       sfm_gathered **gathered_inout)
{
    int rc = check_local_comms(n, comms);
    if (rc != SFM_OK) return rc;
    if (!locals || !gathered_inout) return SFM_ERR_INVALID_ARG;
    std::vector<GatherSource> srcs((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!locals[i]) SFM_FAIL(comms[0]->ctx, SFM_ERR_INVALID_ARG, "gather: rank %d has no batch", i);
        if ((rc = ransac_source(locals[i], &srcs[i])) != SFM_OK) return rc;
    }
    return gather_run(n, comms, srcs.data(), n_pairs_total, root, gathered_inout);
}

// ---- accessors --------------------------------------------------------------------------------------
extern "C" int sfm_gathered_total(sfm_gathered *g, int64_t *total_out, int64_t *per_rank_totals_out)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    if (total_out) *total_out = g->total;
    if (per_rank_totals_out)
        for (int r = 0; r < g->comm->world; r++) per_rank_totals_out[r] = g->totals[r];
    return SFM_OK;
}

static int gathered_fetch_counts(sfm_gathered *g)
{
    if (g->countsFetched) return SFM_OK;
    sfm_ctx *ctx = g->comm->ctx;
    const size_t padded = (size_t)std::max<int64_t>(g->max_owned, 1);
    g->countsHost.resize(padded * g->comm->world);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    SFM_HIP(ctx, hipMemcpyAsync(g->countsHost.data(), g->countsAll.p, g->countsHost.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g->comm->stream));
    const int rc = stream_wait(ctx, g->comm->stream, ctx->comm_timeout_ms);
    if (rc != SFM_OK) return rc;
    g->countsFetched = true;
    return SFM_OK;
}

// Per pair of the whole list (valid on every rank): where its list starts in the rank-major buffer, and
// its length.  Either output may be NULL.
extern "C" int sfm_gathered_layout(sfm_gathered *g, int64_t *starts_out, int32_t *counts_out)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    int rc = gathered_fetch_counts(g);
    if (rc != SFM_OK) return rc;
    return sfm_shard_layout(g->n_total, g->comm->world, g->countsHost.data(), std::max<int64_t>(g->max_owned, 1), starts_out, counts_out);
}

// The gathered lists (root only): pairs_out[2 * total] int32 (i, j), rank-major.  Synchronises.
extern "C" int sfm_gathered_pairs(sfm_gathered *g, int32_t *pairs_out)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = g->comm->ctx;
    if (g->comm->rank != g->root) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: only the root holds the lists");
    if (g->recBytes) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: this object holds RANSAC results (sfm_gathered_ransac reads them)");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    if (g->total > 0) {
        if (!pairs_out) return SFM_ERR_INVALID_ARG;
        SFM_HIP(ctx, hipMemcpyAsync(pairs_out, g->payload.p, (size_t)g->total * sizeof(int2), hipMemcpyDeviceToHost, g->comm->stream));
    }
    return stream_wait(ctx, g->comm->stream, ctx->comm_timeout_ms);
}

// Gathered RANSAC results on the root, in the order of the whole pair list (every output may be NULL):
// winner, its inlier count, error sum, iterations and pose (R row-major, then T), and the ascending inlier
// lists as CSR (inlier_offsets_out[n_total + 1], inliers_out[total]).  Synchronises.
extern "C" int sfm_gathered_ransac(sfm_gathered *g, int32_t *best_hyp_out, int32_t *best_count_out, double *best_toterr_out,
                                   int32_t *iterations_out, double *pose12_out, int64_t *inlier_offsets_out, int32_t *inliers_out)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    sfm_comm *c = g->comm;
    sfm_ctx *ctx = c->ctx;
    if (c->rank != g->root) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: only the root holds the results");
    if (g->recBytes != SFM_RANSAC_RECORD_BYTES) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "gather: this object holds match lists");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    const size_t padded = (size_t)std::max<int64_t>(g->max_owned, 1);
    std::vector<char> rec(padded * c->world * g->recBytes);
    std::vector<int32_t> lists((size_t)std::max<int64_t>(g->total, 1));
    SFM_HIP(ctx, hipMemcpyAsync(rec.data(), g->records.p, rec.size(), hipMemcpyDeviceToHost, c->stream));
    if (g->total > 0 && inliers_out)
        SFM_HIP(ctx, hipMemcpyAsync(lists.data(), g->payload.p, (size_t)g->total * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    int rc = gathered_fetch_counts(g); // (waits for the stream)
    if (rc != SFM_OK) return rc;
    return sfm_shard_unpack_ransac(g->n_total, c->world, g->countsHost.data(), (int64_t)padded, rec.data(), lists.data(), best_hyp_out,
                                   best_count_out, best_toterr_out, iterations_out, pose12_out, inlier_offsets_out, inliers_out);
}

extern "C" int sfm_gathered_device(sfm_gathered *g, const int32_t **pairs_device, const int32_t **counts_device)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    if (pairs_device) *pairs_device = g->comm->rank == g->root ? g->payload.as<int32_t>() : nullptr;
    if (counts_device) *counts_device = g->countsAll.as<int32_t>();
    return SFM_OK;
}

// Blocks until the gather has finished on this rank: its lists are in place on the root, and the local
// result it read may be overwritten.
extern "C" int sfm_gathered_wait(sfm_gathered *g)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = g->comm->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    return stream_wait(ctx, g->comm->stream, ctx->comm_timeout_ms);
}

// The same with a time limit of the caller's (milliseconds; <= 0: for ever): SFM_ERR_TIMEOUT when the gather
// has not finished by then (it is still in flight; the call may be repeated).
extern "C" int sfm_gathered_wait_timeout(sfm_gathered *g, int64_t timeout_ms)
{
    if (!g || !g->comm) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = g->comm->ctx;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    return stream_wait(ctx, g->comm->stream, timeout_ms);
}

extern "C" void sfm_gathered_destroy(sfm_gathered *g)
{
    if (!g) return;
    if (g->comm) {
        (void)hipSetDevice(g->comm->ctx->device);
        (void)hipStreamSynchronize(g->comm->stream);
    }
    DevBuf *bufs[] = {&g->countsMine, &g->countsAll, &g->totalsMine, &g->totalsAll, &g->payload, &g->records};
    for (DevBuf *d : bufs) d->release();
    if (g->comm) comm_release(g->comm);
    delete g;
}

// ---- one process, several devices --------------------------------------------------------------------
// The whole multi-GPU matcher for a caller that owns n devices (the reference is one process): the pair
// list is cut by sfm_shard_indices, every device matches its share against its own copy of the image set
// (all launches are asynchronous, so the devices work side by side), and the lists are gathered to
// ctxs[root]'s device.  results_inout[n] and gathered_inout[n] are reused across calls (pass NULL entries
// the first time).
extern "C" int sfm_match_batch_multi(int n, sfm_comm *const *comms, const sfm_imageset *const *sets, const int32_t *pair_list,
                                     int64_t n_pairs, int metric, float threshold, float max_distance_px_squared, int root,
                                     sfm_match_result **results_inout, sfm_gathered **gathered_inout)
{
    if (n < 1 || !comms || !sets || !results_inout || !gathered_inout || !comms[0] || n_pairs < 0 || (n_pairs > 0 && !pair_list))
        return SFM_ERR_INVALID_ARG;
    const int rcc = check_local_comms(n, comms);
    if (rcc != SFM_OK) return rcc;
    std::vector<int64_t> idx;
    std::vector<int32_t> mine;
    for (int i = 0; i < n; i++) {
        if (!comms[i] || !sets[i]) return SFM_ERR_INVALID_ARG;
        sfm_ctx *ctx = comms[i]->ctx;
        const int64_t cnt = sfm_shard_count(n_pairs, i, n);
        if (cnt > 0x7FFFFFFF) SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "multi: more than 2^31 pairs per device");
        idx.resize((size_t)cnt);
        mine.resize((size_t)cnt * 2);
        int rc = sfm_shard_indices(n_pairs, i, n, idx.data());
        if (rc != SFM_OK) return rc;
        for (int64_t k = 0; k < cnt; k++) {
            mine[2 * k] = pair_list[2 * idx[k]];
            mine[2 * k + 1] = pair_list[2 * idx[k] + 1];
        }
        // (sfm_match_batch copies the list before it returns)
        rc = sfm_match_batch(ctx, sets[i], mine.data(), (int)cnt, metric, threshold, max_distance_px_squared, &results_inout[i]);
        if (rc != SFM_OK) return rc;
    }
    return sfm_match_gather_local(n, comms, results_inout, n_pairs, root, gathered_inout);
}
