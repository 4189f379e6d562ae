// Binary descriptors: mutual nearest neighbours under the Hamming distance.
//
// The reference has no binary descriptors (Keypoint::descriptor is a float vector, SfM.hpp:132), so
// this path has no counterpart there; it applies the structure of matchTwoSided (match.cpp:24-94) to
// the distance popcount(a xor b): a cell is admissible when its distance is at most max_distance and
// the keypoints lie inside the pixel gate; every row and every column takes its nearest admissible
// cell (lowest index on ties, as the strict scans of match.cpp:49,72 do); mutual pairs are kept in
// ascending row order.  All integer work: results are bit-exact against the CPU restatement.
//
// 256-bit descriptors take the matrix cores first: every descriptor is also kept as 256 signs (+1 for a
// set bit, -1 otherwise, one int8 each), whose dot product is 256 - 2 * distance exactly, so the strict
// int8 instance of the float matcher's similarity kernel (v_mfma_i32_16x16x64_i8, A-stationary, B through
// LDS) finds exactly the cells within max_distance; those few are then scored by popcount, gated and
// folded into the row and column maxima.  Pairs whose candidate list overflows (a loose max_distance),
// and the other descriptor sizes, take the popcount kernel below.
//
// Popcount kernel: a thread keeps one descriptor of image A in registers (W 32-bit words), image B streams
// through LDS 256 descriptors at a time and is read as broadcasts, so a cell costs W xor and W
// popcount-accumulate instructions (v_bcnt_u32_b32 adds as it counts) and one minimum.  A row's best is
// a register; a column's best is an atomic maximum on a packed key, reached only by the few cells
// within max_distance.
#include "csr_kernels.h"

struct sfm_binset {
    sfm_ctx *ctx = nullptr;
    int n_images = 0;
    int words = 0; // 32-bit words per descriptor
    std::vector<int32_t> counts;
    std::vector<int64_t> row; // first row of every image
    int64_t rows = 0;
    DevBuf bits, pts;
    // 256-bit descriptors only: the sign table (rows of 256 int8, every image padded to SIGN_ROW_PAD rows
    // so that a block of the similarity kernel may read past an image's end) and each image's first row
    std::vector<int64_t> signRow;
    DevBuf signs;
};

#define SIGN_ROW_PAD 1024 // at least the similarity kernel's rows per block (256) and columns per step (64)
#define SIM_ROWS_PER_BLOCK 256
#define SIM_WAVES_PER_BLOCK 4

// One thread per 16 bits -> 16 signs.
__global__ __launch_bounds__(256) void signs_kernel(const unsigned *__restrict__ bits, const int64_t *__restrict__ rowOf,
                                                   const int64_t *__restrict__ signRowOf, int n_images,
                                                   signed char *__restrict__ signs)
{
    const int img = blockIdx.y;
    const int64_t r0 = rowOf[img], n = rowOf[img + 1] - r0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; // (row, half word)
    if (t >= n * 16) return;
    const int64_t r = t >> 4;
    const int hw = (int)(t & 15);
    const unsigned w = bits[(size_t)(r0 + r) * 8 + (hw >> 1)] >> ((hw & 1) * 16);
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) v |= (((w >> (4 * q + b)) & 1u) ? 0x01u : 0xFFu) << (8 * b);
        out[q] = v;
    }
    *reinterpret_cast<uint4 *>(signs + (size_t)(signRowOf[img] + r) * 256 + hw * 16) = make_uint4(out[0], out[1], out[2], out[3]);
}

struct HamArgs {
    const int *onlyFlagged; // non-null: only the pairs whose entry is not 0
    const PairRec *pairs;
    const unsigned *bits;
    const double2 *pts;
    u64 *best;
    int maxDistance;
    float maxDistSq;
    int bitsPerDesc;
};

#define HAM_TILE 256

// Key of a cell for the row / column maxima: larger = nearer, then lower index; never 0.
__device__ __forceinline__ u64 ham_key(int bitsPerDesc, int d, int other)
{
    return ((u64)(unsigned)(bitsPerDesc - d + 1) << 32) | (u64)(0xFFFFFFFFu - (unsigned)other);
}

// grid (row blocks of 256, pairs of the chunk)
template <int W> __global__ __launch_bounds__(256) void hamming_kernel(const HamArgs g)
{
    __shared__ __attribute__((aligned(16))) unsigned tile[HAM_TILE * W];
    if (g.onlyFlagged && g.onlyFlagged[blockIdx.y] == 0) return;
    const PairRec pr = g.pairs[blockIdx.y];
    const int rowBlock = blockIdx.x * 256;
    if (rowBlock >= pr.nA) return;
    const int tid = threadIdx.x;
    const int i = rowBlock + tid;
    const bool live = i < pr.nA;
    unsigned a[W];
    {
        const unsigned *src = g.bits + (size_t)(pr.rowA32 + (live ? i : pr.nA - 1)) * W;
#pragma unroll
        for (int w = 0; w < W; w++) a[w] = src[w];
    }
    const double2 pa = g.pts[pr.rowA32 + (live ? i : pr.nA - 1)];
    int bestD = 0x7FFFFFFF, bestJ = -1;
    const int thr = g.maxDistance;

    for (int j0 = 0; j0 < pr.nB; j0 += HAM_TILE) {
        const int m = min(HAM_TILE, pr.nB - j0);
        __syncthreads();
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(g.bits + (size_t)(pr.rowB32 + j0) * W);
            uint4 *dst = reinterpret_cast<uint4 *>(tile);
            for (int q = tid; q < m * (W / 4); q += 256) dst[q] = src[q];
        }
        __syncthreads();
        // eight columns at a time: distances first, one test for the group
        for (int jj = 0; jj < m; jj += 8) {
            int d[8];
            int near = 0x7FFFFFFF;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int col = min(jj + u, m - 1); // the last group of a tile repeats its last column
                const uint4 *b = reinterpret_cast<const uint4 *>(tile + col * W);
                unsigned acc = 0;
#pragma unroll
                for (int w4 = 0; w4 < W / 4; w4++) {
                    const uint4 bv = b[w4];
                    acc = __builtin_popcount(a[4 * w4] ^ bv.x) + acc;
                    acc = __builtin_popcount(a[4 * w4 + 1] ^ bv.y) + acc;
                    acc = __builtin_popcount(a[4 * w4 + 2] ^ bv.z) + acc;
                    acc = __builtin_popcount(a[4 * w4 + 3] ^ bv.w) + acc;
                }
                d[u] = (int)acc;
                near = min(near, d[u]);
            }
            if (__builtin_amdgcn_ballot_w64(live && near <= thr) == 0) continue;
            // rare: some lane has a cell within max_distance
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + jj + u;
                if (!live || jj + u >= m || d[u] > thr) continue;
                const double2 pb = g.pts[pr.rowB32 + j];
                const double dx = pa.x - pb.x, dy = pa.y - pb.y;
                if (!(dx * dx + dy * dy < (double)g.maxDistSq)) continue;
                if (d[u] < bestD) { // columns come in ascending order: the first of equals stays
                    bestD = d[u];
                    bestJ = j;
                }
                atomicMax(&g.best[pr.bestCol + j], ham_key(g.bitsPerDesc, d[u], i));
            }
        }
    }
    if (live) g.best[pr.bestRow + i] = bestJ >= 0 ? ham_key(g.bitsPerDesc, bestD, bestJ) : 0ull;
}

// Candidates of the sign pass (all within max_distance): popcount distance, pixel gate, maxima.  One wave
// per slice of a pair's list.  256-bit descriptors.
__global__ __launch_bounds__(256) void hamming_cand_kernel(const HamArgs g, const MatchArgs m)
{
    const int p = blockIdx.y;
    if (m.pairOvf[p] != 0) return; // the popcount kernel redoes the pair
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const PairRec pr = g.pairs[p];
    const int nQueues = ((pr.nA + SIM_ROWS_PER_BLOCK - 1) / SIM_ROWS_PER_BLOCK) * SIM_WAVES_PER_BLOCK;
    if (q >= nQueues) return;
    const int cap = m.candCap / nQueues;
    const int n = m.candCount[p * m.maxQueues + q];
    const uint2 *list = m.cand + (size_t)p * m.candCap + (size_t)q * cap;
    for (int k = lane; k < n; k += 64) {
        const uint2 c = list[k];
        const int i = (int)c.x, j = (int)c.y;
        const uint4 *a = reinterpret_cast<const uint4 *>(g.bits + (size_t)(pr.rowA32 + i) * 8);
        const uint4 *b = reinterpret_cast<const uint4 *>(g.bits + (size_t)(pr.rowB32 + j) * 8);
        const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
        const int d = __builtin_popcount(a0.x ^ b0.x) + __builtin_popcount(a0.y ^ b0.y) + __builtin_popcount(a0.z ^ b0.z) +
                      __builtin_popcount(a0.w ^ b0.w) + __builtin_popcount(a1.x ^ b1.x) + __builtin_popcount(a1.y ^ b1.y) +
                      __builtin_popcount(a1.z ^ b1.z) + __builtin_popcount(a1.w ^ b1.w);
        if (d > g.maxDistance) continue;
        const double2 pa = g.pts[pr.rowA32 + i], pb = g.pts[pr.rowB32 + j];
        const double dx = pa.x - pb.x, dy = pa.y - pb.y;
        if (!(dx * dx + dy * dy < (double)g.maxDistSq)) continue;
        atomicMax(&g.best[pr.bestRow + i], ham_key(g.bitsPerDesc, d, j));
        atomicMax(&g.best[pr.bestCol + j], ham_key(g.bitsPerDesc, d, i));
    }
}

extern "C" void sfm_binset_destroy(sfm_binset *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    s->bits.release();
    s->pts.release();
    s->signs.release();
    sfm_ctx *owner = s->ctx;
    delete s;
    sfm_ctx_release(owner);
}

extern "C" int sfm_binset_create(sfm_ctx *ctx, int n_images, const int32_t *counts, const uint8_t *desc,
                                 const double *pts, int bytes_per_desc, sfm_binset **set_out)
{
    if (!ctx || !set_out) return SFM_ERR_INVALID_ARG;
    *set_out = nullptr;
    if (n_images < 0 || (n_images > 0 && !counts)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "binary set: bad image count");
    if (bytes_per_desc != 16 && bytes_per_desc != 32 && bytes_per_desc != 64)
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "binary set: %d bytes per descriptor has no kernel (16, 32 or 64)", bytes_per_desc);
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_binset *s = new (std::nothrow) sfm_binset();
    if (!s) SFM_FAIL(ctx, SFM_ERR_OOM, "binary set: host allocation failed");
    sfm_ctx_retain(ctx);
    s->ctx = ctx;
    s->n_images = n_images;
    s->words = bytes_per_desc / 4;
    s->counts.assign(counts, counts + n_images);
    s->row.resize((size_t)n_images + 1, 0);
    for (int i = 0; i < n_images; i++) {
        if (counts[i] < 0) {
            sfm_binset_destroy(s);
            SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "binary set: counts[%d] < 0", i);
        }
        s->row[(size_t)i + 1] = s->row[i] + counts[i];
    }
    s->rows = s->row[n_images];
    if (s->rows > 0 && (!desc || !pts)) {
        sfm_binset_destroy(s);
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "binary set: null descriptor or point table");
    }
    const size_t bBits = (size_t)s->rows * bytes_per_desc, bPts = (size_t)s->rows * 2 * sizeof(double);
    hipError_t e = s->bits.reserve(bBits ? bBits : 16);
    if (e == hipSuccess) e = s->pts.reserve(bPts ? bPts : 16);
    if (e == hipSuccess && bBits) e = hipMemcpyAsync(s->bits.p, desc, bBits, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && bPts) e = hipMemcpyAsync(s->pts.p, pts, bPts, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && s->words == 8) { // sign table for the matrix-core pass
        s->signRow.resize((size_t)n_images + 1, 0);
        int maxN = 0;
        for (int i = 0; i < n_images; i++) {
            s->signRow[(size_t)i + 1] = s->signRow[i] + round_up(counts[i], SIGN_ROW_PAD);
            maxN = std::max(maxN, counts[i]);
        }
        const size_t bSigns = (size_t)(s->signRow[n_images] + SIGN_ROW_PAD) * 256;
        DevBuf rowDev;
        e = s->signs.reserve(bSigns);
        if (e == hipSuccess) e = rowDev.reserve((size_t)(n_images + 1) * 2 * sizeof(int64_t));
        if (e == hipSuccess) e = hipMemsetAsync(s->signs.p, 0, bSigns, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(rowDev.p, s->row.data(), (size_t)(n_images + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(rowDev.as<int64_t>() + n_images + 1, s->signRow.data(), (size_t)(n_images + 1) * sizeof(int64_t),
                               hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && maxN > 0) {
            hipLaunchKernelGGL(signs_kernel, dim3((unsigned)(((int64_t)maxN * 16 + 255) / 256), n_images), dim3(256), 0, ctx->stream,
                               s->bits.as<unsigned>(), rowDev.as<int64_t>(), rowDev.as<int64_t>() + n_images + 1, n_images,
                               s->signs.as<signed char>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        rowDev.release();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        sfm_binset_destroy(s);
        SFM_FAIL(ctx, e == hipErrorOutOfMemory ? SFM_ERR_OOM : SFM_ERR_HIP, "binary set: %s", hipGetErrorString(e));
    }
    *set_out = s;
    return SFM_OK;
}

extern "C" int sfm_match_batch_hamming(sfm_ctx *ctx, const sfm_binset *set, const int32_t *pair_list, int n_pairs,
                                       int max_distance, float max_distance_px_squared, sfm_match_result **result_inout)
{
    if (!ctx || !set || !result_inout) return SFM_ERR_INVALID_ARG;
    if (set->ctx != ctx) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hamming: image set belongs to another context");
    if (n_pairs < 0 || (n_pairs > 0 && !pair_list)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hamming: bad pair list");
    const int bitsPerDesc = set->words * 32;
    if (max_distance < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hamming: max_distance must be >= 0 (was %d)", max_distance);
    if (max_distance > bitsPerDesc) max_distance = bitsPerDesc;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    sfm_match_result *res = *result_inout;
    if (!res) {
        res = new (std::nothrow) sfm_match_result();
        if (!res) SFM_FAIL(ctx, SFM_ERR_OOM, "hamming: host allocation failed");
        sfm_ctx_retain(ctx);
        res->ctx = ctx;
        *result_inout = res;
    } else if (res->ctx != ctx) {
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hamming: result belongs to another context");
    } else {
        // The staging buffers of a reused result may still feed copies in flight, and a float call that used it
        // before spreads its work over the context's three streams: wait for all of them.
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stream2) SFM_HIP(ctx, hipStreamSynchronize(ctx->stream2));
        if (ctx->stream3) SFM_HIP(ctx, hipStreamSynchronize(ctx->stream3));
    }
    // This call uploads from the result's staging buffers on the context's stream and records no upload marker:
    // the next call on this result, of either kind, waits for the streams instead of an event of an older call.
    res->upValid = false;
    SFM_HIP(ctx, res->readFence.wait_on(ctx->stream)); // a gather of the call before may still read the lists
    res->n_pairs = n_pairs;
    res->total = -1;
    hipStream_t st = ctx->stream;
    const int chunkPairs = (int)std::min<int64_t>(ctx->pairs_per_launch, n_pairs > 0 ? n_pairs : 1);

    SFM_HIP(ctx, res->hostRecs.reserve((size_t)(n_pairs ? n_pairs : 1) * sizeof(PairRec)));
    PairRec *recs = res->hostRecs.as<PairRec>();
    int64_t outCap = 0, maxBest = 0;
    int maxNA = 0;
    for (int c0 = 0; c0 < n_pairs; c0 += chunkPairs) {
        const int c1 = std::min(n_pairs, c0 + chunkPairs);
        int64_t bestOff = 0;
        for (int p = c0; p < c1; p++) {
            const int ia = pair_list[2 * p], ib = pair_list[2 * p + 1];
            if (ia < 0 || ia >= set->n_images || ib < 0 || ib >= set->n_images)
                SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "hamming: pair %d names image %d/%d outside the set", p, ia, ib);
            PairRec &r = recs[p];
            memset(&r, 0, sizeof r);
            r.nA = set->counts[ia];
            r.nB = set->counts[ib];
            r.rowA32 = set->row[ia];
            r.rowB32 = set->row[ib];
            r.bestRow = bestOff;
            r.bestCol = bestOff + r.nA;
            r.chunkLocal = p - c0;
            bestOff += (int64_t)r.nA + r.nB;
            outCap += std::min(r.nA, r.nB);
            maxNA = std::max(maxNA, r.nA);
        }
        maxBest = std::max(maxBest, bestOff);
    }
    // 256-bit descriptors: matrix-core pass first, unless the caller turned it off or max_distance admits
    // every cell anyway
    const bool signPass = set->words == 8 && ctx->hamming_mfma && max_distance < 128 && ctx->cand_capacity > 0;
    const int candCap = (int)ctx->cand_capacity;
    const int maxQueues = std::max(1, (maxNA + SIM_ROWS_PER_BLOCK - 1) / SIM_ROWS_PER_BLOCK) * SIM_WAVES_PER_BLOCK;
    std::vector<int64_t> workBase;
    std::vector<int> workCount;
    int64_t workTotal = 0;
    if (signPass) {
        for (int c0 = 0; c0 < n_pairs; c0 += chunkPairs) {
            const int c1 = std::min(n_pairs, c0 + chunkPairs);
            int64_t len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int p = c0; p < c1; p++) {
                recs[p].rowA16 = set->signRow[pair_list[2 * p]];
                recs[p].rowB16 = set->signRow[pair_list[2 * p + 1]];
                recs[p].thr_i = 256 - 2 * max_distance - 1; // sum > thr_i  <=>  distance <= max_distance
                if (recs[p].nA > 0 && recs[p].nB > 0)
                    len[(p - c0) & 7] += (recs[p].nA + SIM_ROWS_PER_BLOCK - 1) / SIM_ROWS_PER_BLOCK;
            }
            int64_t qmax = 0;
            for (int x = 0; x < 8; x++) qmax = std::max(qmax, len[x]);
            workBase.push_back(workTotal);
            workCount.push_back((int)(qmax * 8));
            workTotal += qmax * 8;
        }
        // Blocks b and b + 8 run on the same XCD (round-robin dispatch): a pair's row blocks go to one
        // residue class mod 8 and share that XCD's L2 copy of the streamed image.
        SFM_HIP(ctx, res->hostWork.reserve((size_t)(workTotal ? workTotal : 1) * sizeof(int2)));
        int2 *work = res->hostWork.as<int2>();
        size_t ci = 0;
        for (int c0 = 0; c0 < n_pairs; c0 += chunkPairs, ci++) {
            const int c1 = std::min(n_pairs, c0 + chunkPairs);
            int2 *wl = work + workBase[ci];
            for (int k = 0; k < workCount[ci]; k++) wl[k] = make_int2(-1, 0);
            int64_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int p = c0; p < c1; p++) {
                if (recs[p].nA <= 0 || recs[p].nB <= 0) continue;
                const int x = (p - c0) & 7;
                const int nb = (recs[p].nA + SIM_ROWS_PER_BLOCK - 1) / SIM_ROWS_PER_BLOCK;
                for (int rb = 0; rb < nb; rb++) wl[(pos[x]++) * 8 + x] = make_int2(p - c0, rb);
            }
        }
        SFM_HIP(ctx, res->work.reserve((size_t)(workTotal ? workTotal : 1) * sizeof(int2)));
        if (workTotal)
            SFM_HIP(ctx, hipMemcpyAsync(res->work.p, work, (size_t)workTotal * sizeof(int2), hipMemcpyHostToDevice, st));
        SFM_HIP(ctx, res->cand.reserve((size_t)chunkPairs * candCap * sizeof(uint2)));
        SFM_HIP(ctx, res->candCount.reserve((size_t)chunkPairs * maxQueues * sizeof(int)));
    }
    SFM_HIP(ctx, res->pairOvf.reserve((size_t)(chunkPairs + 4) * sizeof(int)));
    SFM_HIP(ctx, res->pairRecs.reserve((size_t)(n_pairs ? n_pairs : 1) * sizeof(PairRec)));
    SFM_HIP(ctx, res->offsets.reserve((size_t)(n_pairs + 1) * sizeof(int64_t)));
    SFM_HIP(ctx, res->pairs.reserve((size_t)(outCap ? outCap : 1) * sizeof(int2)));
    SFM_HIP(ctx, res->best.reserve((size_t)(maxBest ? maxBest : 1) * sizeof(u64)));
    SFM_HIP(ctx, res->counts.reserve((size_t)chunkPairs * sizeof(int)));
    SFM_HIP(ctx, res->stats.reserve(4 * sizeof(u64)));
    SFM_HIP(ctx, hipMemsetAsync(res->offsets.p, 0, sizeof(int64_t), st));
    SFM_HIP(ctx, hipMemsetAsync(res->stats.p, 0, 4 * sizeof(u64), st));
    if (n_pairs > 0)
        SFM_HIP(ctx, hipMemcpyAsync(res->pairRecs.p, recs, (size_t)n_pairs * sizeof(PairRec), hipMemcpyHostToDevice, st));

    HamArgs g;
    g.bits = set->bits.as<unsigned>();
    g.pts = set->pts.as<double2>();
    g.best = res->best.as<u64>();
    g.maxDistance = max_distance;
    g.maxDistSq = max_distance_px_squared;
    g.bitsPerDesc = bitsPerDesc;
    g.onlyFlagged = nullptr;
    MatchArgs m;
    memset(&m, 0, sizeof m);
    if (signPass) {
        m.desc8 = set->signs.as<signed char>();
        m.cand = res->cand.as<uint2>();
        m.candCount = res->candCount.as<int>();
        m.pairOvf = res->pairOvf.as<int>();
        m.flagCount = res->pairOvf.as<int>() + chunkPairs;
        m.maxQueues = maxQueues;
        m.candCap = candCap;
        m.blockRows = SIM_ROWS_PER_BLOCK;
        m.wavesPerBlock = SIM_WAVES_PER_BLOCK;
        m.dim = 256;
    }
    sfm_span_reset(ctx, 0);
    size_t ci = 0;
    for (int c0 = 0; c0 < n_pairs; c0 += chunkPairs, ci++) {
        const int c1 = std::min(n_pairs, c0 + chunkPairs);
        const int np = c1 - c0;
        g.pairs = res->pairRecs.as<PairRec>() + c0;
        const int64_t bestLen = recs[c1 - 1].bestCol + recs[c1 - 1].nB;
        if (bestLen > 0) SFM_HIP(ctx, hipMemsetAsync(res->best.p, 0, (size_t)bestLen * sizeof(u64), st));
        const dim3 grid((maxNA + 255) / 256, np);
        if (signPass && workCount[ci] > 0) {
            m.pairs = g.pairs;
            m.work = res->work.as<int2>() + workBase[ci];
            SFM_HIP(ctx, hipMemsetAsync(res->candCount.p, 0, (size_t)np * maxQueues * sizeof(int), st));
            SFM_HIP(ctx, hipMemsetAsync(res->pairOvf.p, 0, (size_t)(chunkPairs + 4) * sizeof(int), st));
            SFM_HIP(ctx, sfm_span_begin(ctx, 0));
            sfm_launch_sim_signs(m, workCount[ci], st);
            SFM_HIP(ctx, sfm_span_end(ctx, 0));
            hipLaunchKernelGGL(hamming_cand_kernel, dim3((maxQueues + 3) / 4, np), dim3(256), 0, st, g, m);
            g.onlyFlagged = res->pairOvf.as<int>(); // overflowed lists: the popcount kernel takes the pair
            hipLaunchKernelGGL(hamming_kernel<8>, grid, dim3(256), 0, st, g);
        } else if (maxNA > 0) {
            SFM_HIP(ctx, sfm_span_begin(ctx, 0));
            if (set->words == 4) hipLaunchKernelGGL(hamming_kernel<4>, grid, dim3(256), 0, st, g);
            else if (set->words == 8) hipLaunchKernelGGL(hamming_kernel<8>, grid, dim3(256), 0, st, g);
            else hipLaunchKernelGGL(hamming_kernel<16>, grid, dim3(256), 0, st, g);
            SFM_HIP(ctx, sfm_span_end(ctx, 0));
        }
        hipLaunchKernelGGL((match_mutual_kernel<false>), dim3(np), dim3(256), 0, st, g.pairs, g.best,
                           res->counts.as<int>(), (const int64_t *)nullptr, (int2 *)nullptr, (int32_t *)nullptr,
                           (int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr);
        hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, res->counts.as<int>(), np,
                           res->offsets.as<int64_t>(), (int64_t)c0);
        hipLaunchKernelGGL((match_mutual_kernel<true>), dim3(np), dim3(256), 0, st, g.pairs, g.best, (int *)nullptr,
                           res->offsets.as<int64_t>() + c0, res->pairs.as<int2>(), (int32_t *)nullptr,
                           (int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr);
        SFM_HIP(ctx, hipGetLastError());
    }
    // the point on the context's stream where this call's lists are complete (the multi-GPU gather waits for it)
    return sfm_match_result_mark_done(ctx, res);
}
