// Scale chaining and the running average of the 3-D points, on the device.
//
// The reference closes every frame step by matching the scale of the new two-view solution to the
// chain so far and folding the triangulated points into per-track running means
// (submodules/scene.cpp:301-421).  Only two things in that are sequential: the accumulated scale and
// pose (a handful of 4x4 operations per step) and, per track, its running mean.  Everything else
// is spread out:
//   chain_index_kernel   a table slot[step][track] = the step's point on that track (all points at once)
//   chain_ratio_kernel   one block per step: depth ratios against the previous step through the table,
//                        exact median by radix selection on the ratios' bit patterns
//   chain_walk_kernel    one lane: scale, view matrix, translation clamp, pose accumulation, step by step
//   chain_points_kernel  one thread per track: walks the steps through the table and updates its mean
// The colour average (scene.cpp:394-407) runs beside the mean when the caller hands over the colour of every
// observation (scene.cpp:377 samples it from the image: pixel_bilinear_kernel restates that sampler).
//
// Every floating-point operation is written in the order the CPU restatement uses (contraction off),
// and the median is an exact selection, so the two agree bit for bit.
#include "common.h"

#pragma clang fp contract(off)

typedef unsigned long long u64;

#define CHAIN_THREADS 256

struct ChainArgs {
    int nSteps;
    int nTracks;
    const int64_t *offsets; // [nSteps + 1] row base of step s in tracks / points
    const int32_t *tracks;  // track id per correspondence row (distinct inside a step's point set)
    const double *points;   // [rows][3]; the k-th point of step s at row offsets[s] + k
    const double *poses;    // [nSteps][16] column-major
    // optional selection (the refit's point set): point k of step s belongs to correspondence
    // sel[selOff[s] + k] when selCount[s] >= 8, to correspondence k otherwise; used[s] points exist
    const int32_t *used;
    const int32_t *selCount;
    const int64_t *selOff;
    const int32_t *sel;
    double maxTranslation;
    double start[16];
    double *extrinsics; // [nSteps + 1][16]
    double *points3d;   // [nTracks][3]
    int32_t *counts;    // [nTracks]
    double *accScale;   // [nSteps]
    const unsigned char *colors; // optional [rows][3]: the colour of the observation behind every point
    unsigned char *colors3d;     // [nTracks][3] running average of a track's colours (with colors)
    // work space
    int32_t *slot;      // [rows of the window][nTracks]: the point of a step that carries a track, or -1
    u64 *keys;          // per correspondence row: bit pattern of the point's depth ratio, 0 = none
    double *rel;        // [nSteps] median ratio of the step, 0 = leave the scale
    double *state;      // accumulated pose (16) and scale carried from window to window
    int *error;         // [2] first offending step + 1, kind
    // window of steps [s0, s1); slot row r holds step s0 - 1 + r
    int s0, s1;
};

// (P * p.homogeneous()).head<3>(), scene.cpp:326, :376
__device__ __forceinline__ void pose_apply(const double *P, double x, double y, double z, double out[3])
{
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = ((P[r] * x + P[r + 4] * y) + P[r + 8] * z) + P[r + 12];
}

// Inverse of a column-major 4x4 by adjugate / determinant over 2x2 sub-determinants.
__device__ void mat4_inverse(const double *m, double *b)
{
#define M(r, c) m[(r) + 4 * (c)]
#define B_(r, c) b[(r) + 4 * (c)]
    const double s0 = M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1);
    const double s1 = M(0, 0) * M(1, 2) - M(1, 0) * M(0, 2);
    const double s2 = M(0, 0) * M(1, 3) - M(1, 0) * M(0, 3);
    const double s3 = M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2);
    const double s4 = M(0, 1) * M(1, 3) - M(1, 1) * M(0, 3);
    const double s5 = M(0, 2) * M(1, 3) - M(1, 2) * M(0, 3);
    const double c5 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
    const double c4 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
    const double c3 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
    const double c2 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
    const double c1 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
    const double c0 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
    const double det = ((((s0 * c5 - s1 * c4) + s2 * c3) + s3 * c2) - s4 * c1) + s5 * c0;
    const double id = 1.0 / det;
    B_(0, 0) = ((M(1, 1) * c5 - M(1, 2) * c4) + M(1, 3) * c3) * id;
    B_(0, 1) = ((-M(0, 1) * c5 + M(0, 2) * c4) - M(0, 3) * c3) * id;
    B_(0, 2) = ((M(3, 1) * s5 - M(3, 2) * s4) + M(3, 3) * s3) * id;
    B_(0, 3) = ((-M(2, 1) * s5 + M(2, 2) * s4) - M(2, 3) * s3) * id;
    B_(1, 0) = ((-M(1, 0) * c5 + M(1, 2) * c2) - M(1, 3) * c1) * id;
    B_(1, 1) = ((M(0, 0) * c5 - M(0, 2) * c2) + M(0, 3) * c1) * id;
    B_(1, 2) = ((-M(3, 0) * s5 + M(3, 2) * s2) - M(3, 3) * s1) * id;
    B_(1, 3) = ((M(2, 0) * s5 - M(2, 2) * s2) + M(2, 3) * s1) * id;
    B_(2, 0) = ((M(1, 0) * c4 - M(1, 1) * c2) + M(1, 3) * c0) * id;
    B_(2, 1) = ((-M(0, 0) * c4 + M(0, 1) * c2) - M(0, 3) * c0) * id;
    B_(2, 2) = ((M(3, 0) * s4 - M(3, 1) * s2) + M(3, 3) * s0) * id;
    B_(2, 3) = ((-M(2, 0) * s4 + M(2, 1) * s2) - M(2, 3) * s0) * id;
    B_(3, 0) = ((-M(1, 0) * c3 + M(1, 1) * c1) - M(1, 2) * c0) * id;
    B_(3, 1) = ((M(0, 0) * c3 - M(0, 1) * c1) + M(0, 2) * c0) * id;
    B_(3, 2) = ((-M(3, 0) * s3 + M(3, 1) * s1) - M(3, 2) * s0) * id;
    B_(3, 3) = ((M(2, 0) * s3 - M(2, 1) * s1) + M(2, 2) * s0) * id;
#undef B_
#undef M
}

struct StepView {
    int64_t base;
    int n;
    const int32_t *list; // null: point k is correspondence k
};

__device__ __forceinline__ StepView step_view(const ChainArgs &g, int s)
{
    StepView v;
    v.base = g.offsets[s];
    const int nAll = (int)(g.offsets[s + 1] - v.base);
    v.n = g.used ? g.used[s] : nAll;
    v.list = (g.sel && g.selCount[s] >= 8) ? g.sel + g.selOff[s] : nullptr;
    return v;
}

__device__ __forceinline__ int step_track(const ChainArgs &g, const StepView &v, int k)
{
    return g.tracks[v.base + (v.list ? v.list[k] : k)];
}

__device__ __forceinline__ int32_t *slot_row(const ChainArgs &g, int s)
{
    return g.slot + (size_t)(s - (g.s0 - 1)) * (size_t)g.nTracks;
}

enum { CHAIN_ERR_RANGE = 1, CHAIN_ERR_TWICE = 2, CHAIN_ERR_COUNT = 3 };

// grid (blocks per step, steps of the window + 1): blockIdx.y = 0 is step s0 - 1
__global__ __launch_bounds__(CHAIN_THREADS) void chain_index_kernel(const ChainArgs g)
{
    const int s = g.s0 - 1 + (int)blockIdx.y;
    if (s < 0) return;
    const StepView v = step_view(g, s);
    if (v.n < 0 || v.n > (int)(g.offsets[s + 1] - v.base)) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && atomicCAS(&g.error[0], 0, s + 1) == 0) g.error[1] = CHAIN_ERR_COUNT;
        return;
    }
    int32_t *row = slot_row(g, s);
    for (int k = blockIdx.x * CHAIN_THREADS + threadIdx.x; k < v.n; k += gridDim.x * CHAIN_THREADS) {
        const int t = step_track(g, v, k);
        int kind = 0;
        if (t < 0 || t >= g.nTracks) kind = CHAIN_ERR_RANGE;
        else if (atomicExch(&row[t], k) != -1) kind = CHAIN_ERR_TWICE;
        if (kind && atomicCAS(&g.error[0], 0, s + 1) == 0) g.error[1] = kind;
    }
}

// One block per step s in [max(s0, 1), s1): rel[s] = the median depth ratio, or 0.
__global__ __launch_bounds__(CHAIN_THREADS) void chain_ratio_kernel(const ChainArgs g)
{
    __shared__ unsigned sHist[256];
    __shared__ unsigned sWave[4];
    __shared__ u64 sPrefix;
    __shared__ unsigned sRank, sTotal;

    const int s = (g.s0 > 1 ? g.s0 : 1) + (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (g.error[0]) return; // uniform: written by the kernel before this one
    const StepView cur = step_view(g, s), old = step_view(g, s - 1);
    const int32_t *prevRow = slot_row(g, s - 1);
    const double *pose12 = g.poses + 16 * (size_t)(s - 1);
    u64 *keys = g.keys + cur.base;

    // scene.cpp:301-338.  The reference walks both ascending id lists in step; the table finds the same
    // pairs, and the median does not depend on their order.
    for (int k = tid; k < cur.n; k += CHAIN_THREADS) {
        const int k12 = prevRow[step_track(g, cur, k)];
        u64 key = 0;
        if (k12 >= 0) {
            const double *p12 = g.points + 3 * (size_t)(old.base + k12);
            const double *p23 = g.points + 3 * (size_t)(cur.base + k);
            double q[3];
            pose_apply(pose12, p12[0], p12[1], p12[2], q);
            const double dist12 = sqrt((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
            const double dist23 = sqrt((p23[0] * p23[0] + p23[1] * p23[1]) + p23[2] * p23[2]);
            if (dist23 > 0.1 && dist12 > 0.1 && dist23 < 100.0 && dist12 < 100.0)
                key = (u64)__double_as_longlong(dist12 / dist23); // positive: bit patterns order like values
        }
        keys[k] = key;
    }
    if (tid == 0) sPrefix = 0;
    __syncthreads();

    // getMedian, scene.cpp:486-493: the element of rank n / 2.  Radix selection, 8 bits a pass; the first
    // pass also counts the ratios.
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        const u64 highMask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        sHist[tid] = 0;
        __syncthreads();
        const u64 prefix = sPrefix;
        for (int k = tid; k < cur.n; k += CHAIN_THREADS) {
            const u64 key = keys[k];
            if (key != 0 && ((key ^ prefix) & highMask) == 0) atomicAdd(&sHist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 bins
        const unsigned mine = sHist[tid];
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        unsigned before = 0;
        for (int w = 0; w < wave; w++) before += sWave[w];
        incl += before;
        if (pass == 0) {
            if (tid == CHAIN_THREADS - 1) {
                sTotal = incl;
                sRank = incl / 2;
            }
            __syncthreads();
            if (sTotal < 5) { // scene.cpp:340
                if (tid == 0) g.rel[s] = 0.0;
                return;
            }
        }
        const unsigned rank = sRank;
        __syncthreads();
        if (rank < incl && rank >= incl - mine) { // exactly one bin
            sRank = rank - (incl - mine);
            sPrefix = prefix | ((u64)tid << shift);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double rel = __longlong_as_double((long long)sPrefix);
        g.rel[s] = (rel >= 0.1 && rel < 10) ? rel : 0.0; // scene.cpp:344
    }
}

// One lane walks the steps of the window: scene.cpp:346, :356-365, :421.
__global__ __launch_bounds__(64) void chain_walk_kernel(const ChainArgs g)
{
    if (threadIdx.x != 0 || g.error[0]) return;
    double acc[16], scale;
    if (g.s0 == 0) {
        for (int e = 0; e < 16; e++) {
            acc[e] = g.start[e];
            g.extrinsics[e] = g.start[e];
        }
        scale = 1.0;
    } else {
        for (int e = 0; e < 16; e++) acc[e] = g.state[e];
        scale = g.state[16];
    }
    for (int s = g.s0; s < g.s1; s++) {
        if (s > 0) {
            const double rel = g.rel[s];
            if (rel != 0.0) scale = scale * rel;
        }
        g.accScale[s] = scale;
        double view[16], inv[16], next[16];
        const double *pose23 = g.poses + 16 * (size_t)s;
        for (int e = 0; e < 16; e++) view[e] = pose23[e];
        for (int r = 0; r < 3; r++) view[12 + r] *= scale;
        const double len = sqrt((view[12] * view[12] + view[13] * view[13]) + view[14] * view[14]);
        if (len > g.maxTranslation) {
            const double f = g.maxTranslation / len;
            for (int r = 0; r < 3; r++) view[12 + r] *= f;
        }
        mat4_inverse(view, inv);
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++)
                next[r + 4 * c] = ((acc[r] * inv[4 * c] + acc[r + 4] * inv[1 + 4 * c]) + acc[r + 8] * inv[2 + 4 * c]) +
                                  acc[r + 12] * inv[3 + 4 * c];
        for (int e = 0; e < 16; e++) {
            acc[e] = next[e];
            g.extrinsics[16 * (size_t)(s + 1) + e] = next[e];
        }
    }
    for (int e = 0; e < 16; e++) g.state[e] = acc[e];
    g.state[16] = scale;
}

// One thread per track: scene.cpp:367-411 for the steps of the window, in order.  Step s places its
// points with the pose and scale before the step's own update: extrinsics[s], accScale[s].
__global__ __launch_bounds__(CHAIN_THREADS) void chain_points_kernel(const ChainArgs g)
{
    const int t = blockIdx.x * CHAIN_THREADS + threadIdx.x;
    if (t >= g.nTracks || g.error[0]) return;
    double m0 = 0.0, m1 = 0.0, m2 = 0.0;
    unsigned char col[3] = {0, 0, 0};
    int N = 0;
    if (g.s0 > 0) {
        m0 = g.points3d[3 * (size_t)t];
        m1 = g.points3d[3 * (size_t)t + 1];
        m2 = g.points3d[3 * (size_t)t + 2];
        N = g.counts[t];
        if (g.colors)
            for (int c = 0; c < 3; c++) col[c] = g.colors3d[3 * (size_t)t + c];
    }
    for (int s = g.s0; s < g.s1; s++) {
        const int k = slot_row(g, s)[t];
        if (k < 0) continue;
        const double *p = g.points + 3 * (size_t)(g.offsets[s] + k);
        const double px = p[0], py = p[1], pz = p[2];
        if (pz < 0) continue;
        const double scale = g.accScale[s];
        double gl[3];
        pose_apply(g.extrinsics + 16 * (size_t)s, scale * px, scale * py, scale * pz, gl);
        const unsigned char *nc = g.colors ? g.colors + 3 * (size_t)(g.offsets[s] + k) : nullptr;
        if (N == 0) {
            m0 = gl[0];
            m1 = gl[1];
            m2 = gl[2];
            if (nc)
                for (int c = 0; c < 3; c++) col[c] = nc[c];
            N = 1;
        } else {
            const double d0 = m0 - gl[0], d1 = m1 - gl[1], d2 = m2 - gl[2];
            if (sqrt((d0 * d0 + d1 * d1) + d2 * d2) < 50.0) {
                const double N1 = (double)(N + 1);
                m0 = m0 + (gl[0] - m0) / N1;
                m1 = m1 + (gl[1] - m1) / N1;
                m2 = m2 + (gl[2] - m2) / N1;
                if (nc) { // scene.cpp:394-407: in float, clamped, cut off towards zero
                    for (int c = 0; c < 3; c++) {
                        const float cNew = (float)nc[c], cOld = (float)col[c];
                        float res = cOld + (cNew - cOld) / (float)(N + 1);
                        res = res < 0.f ? 0.f : (res > 255.f ? 255.f : res);
                        col[c] = (unsigned char)res;
                    }
                }
                N = N + 1;
            }
        }
    }
    g.points3d[3 * (size_t)t] = m0;
    g.points3d[3 * (size_t)t + 1] = m1;
    g.points3d[3 * (size_t)t + 2] = m2;
    g.counts[t] = N;
    if (g.colors)
        for (int c = 0; c < 3; c++) g.colors3d[3 * (size_t)t + c] = col[c];
}

// util::getPixelBilinearUchar (util.cpp:40-78) for n positions of one image: BGR bytes in, RGB bytes out; the
// weights and the weighted sum in double, in the reference's order; cv::saturate_cast<uchar>(double) rounds to
// the nearest integer, ties to even, and clamps.
__global__ __launch_bounds__(256) void pixel_bilinear_kernel(const unsigned char *__restrict__ bgr, int rows, int cols,
                                                            const double2 *__restrict__ pixels, int n,
                                                            unsigned char *__restrict__ rgb)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double u = pixels[i].x, v = pixels[i].y;
    const double fu = floor(u), fv = floor(v);
    // (the cast of a value beyond int's range is undefined in the reference; here it saturates, and NaN goes to 0)
    int x0 = fu >= 2147483647.0 ? 2147483647 : (fu <= -2147483648.0 ? (int)-2147483648ll : (int)fu);
    int y0 = fv >= 2147483647.0 ? 2147483647 : (fv <= -2147483648.0 ? (int)-2147483648ll : (int)fv);
    x0 = min(max(x0, 0), cols - 2);
    y0 = min(max(y0, 0), rows - 2);
    const double dx = u - x0, dy = v - y0;
    const double w00 = (1.0 - dx) * (1.0 - dy), w10 = dx * (1.0 - dy), w01 = (1.0 - dx) * dy, w11 = dx * dy;
    const unsigned char *p00 = bgr + 3 * ((size_t)y0 * cols + x0), *p10 = p00 + 3, *p01 = p00 + 3 * (size_t)cols, *p11 = p01 + 3;
    unsigned char out[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double val = ((w00 * p00[c] + w10 * p10[c]) + w01 * p01[c]) + w11 * p11[c];
        const double r = rint(val); // nearest, ties to even
        out[c] = (unsigned char)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r)); // (NaN: 0)
    }
    rgb[3 * (size_t)i] = out[2]; // BGR -> RGB
    rgb[3 * (size_t)i + 1] = out[1];
    rgb[3 * (size_t)i + 2] = out[0];
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
struct ChainScratch {
    DevBuf offsets, tracks, points, poses, extr, pts3d, counts, scale, slot, keys, rel, state, colors, cols3d, image, pixels;
    PinBuf err;
    void release()
    {
        DevBuf *all[] = {&offsets, &tracks, &points, &poses, &extr, &pts3d, &counts, &scale, &slot, &keys, &rel, &state,
                         &colors, &cols3d, &image, &pixels};
        for (DevBuf *d : all) d->release();
        err.release();
    }
};

void sfm_chain_scratch_free(sfm_ctx *ctx)
{
    if (!ctx->chain) return;
    ctx->chain->release();
    delete ctx->chain;
    ctx->chain = nullptr;
}

static ChainScratch *chain_scratch(sfm_ctx *ctx)
{
    if (!ctx->chain) ctx->chain = new (std::nothrow) ChainScratch();
    return ctx->chain;
}

// Launches the four kernels window by window and fetches the outputs.  a: inputs filled in.
static int chain_run(sfm_ctx *ctx, ChainScratch &w, ChainArgs &a, int64_t total, int maxStep, double *extrinsics_out,
                     double *points3d_out, int32_t *counts_out, double *acc_scale_out, unsigned char *colors_out = nullptr)
{
    hipStream_t st = ctx->stream;
    const size_t S = (size_t)a.nSteps, T = (size_t)a.nTracks;
    // steps per window: the table holds one row per step and one for the step before the window
    int64_t G = ctx->chain_table_entries / (int64_t)(T ? T : 1) - 1;
    G = std::max<int64_t>(1, std::min<int64_t>(G, (int64_t)(S ? S : 1)));
    SFM_HIP(ctx, w.extr.reserve((S + 1) * 16 * sizeof(double)));
    SFM_HIP(ctx, w.pts3d.reserve((T ? T : 1) * 3 * sizeof(double)));
    SFM_HIP(ctx, w.counts.reserve((T ? T : 1) * sizeof(int32_t)));
    if (a.colors) {
        SFM_HIP(ctx, w.cols3d.reserve((T ? T : 1) * 3));
        a.colors3d = w.cols3d.as<unsigned char>();
        if (T) SFM_HIP(ctx, hipMemsetAsync(w.cols3d.p, 0, T * 3, ctx->stream));
    }
    SFM_HIP(ctx, w.scale.reserve((S ? S : 1) * sizeof(double)));
    SFM_HIP(ctx, w.slot.reserve((size_t)(G + 1) * (T ? T : 1) * sizeof(int32_t)));
    SFM_HIP(ctx, w.keys.reserve((size_t)(total ? total : 1) * sizeof(u64)));
    SFM_HIP(ctx, w.rel.reserve((S ? S : 1) * sizeof(double)));
    SFM_HIP(ctx, w.state.reserve(17 * sizeof(double) + 2 * sizeof(int)));
    SFM_HIP(ctx, w.err.reserve(2 * sizeof(int)));
    a.extrinsics = w.extr.as<double>();
    a.points3d = w.pts3d.as<double>();
    a.counts = w.counts.as<int32_t>();
    a.accScale = w.scale.as<double>();
    a.slot = w.slot.as<int32_t>();
    a.keys = w.keys.as<u64>();
    a.rel = w.rel.as<double>();
    a.state = w.state.as<double>();
    a.error = reinterpret_cast<int *>(w.state.as<double>() + 17);
    SFM_HIP(ctx, hipMemsetAsync(a.error, 0, 2 * sizeof(int), st));
    if (S == 0) { // no step: the start pose and an empty cloud
        SFM_HIP(ctx, hipMemcpyAsync(w.extr.p, a.start, 16 * sizeof(double), hipMemcpyHostToDevice, st));
        if (T) {
            SFM_HIP(ctx, hipMemsetAsync(w.pts3d.p, 0, T * 3 * sizeof(double), st));
            SFM_HIP(ctx, hipMemsetAsync(w.counts.p, 0, T * sizeof(int32_t), st));
        }
    }
    const int stepBlocks = std::max(1, std::min(64, (maxStep + CHAIN_THREADS - 1) / CHAIN_THREADS));
    for (int64_t s0 = 0; s0 < (int64_t)S; s0 += G) {
        a.s0 = (int)s0;
        a.s1 = (int)std::min<int64_t>(s0 + G, (int64_t)S);
        const int rows = a.s1 - a.s0 + 1;
        if (T) SFM_HIP(ctx, hipMemsetAsync(w.slot.p, 0xFF, (size_t)rows * T * sizeof(int32_t), st));
        hipLaunchKernelGGL(chain_index_kernel, dim3(stepBlocks, rows), dim3(CHAIN_THREADS), 0, st, a);
        const int firstRatio = a.s0 > 1 ? a.s0 : 1;
        if (a.s1 > firstRatio)
            hipLaunchKernelGGL(chain_ratio_kernel, dim3(a.s1 - firstRatio), dim3(CHAIN_THREADS), 0, st, a);
        hipLaunchKernelGGL(chain_walk_kernel, dim3(1), dim3(64), 0, st, a);
        if (T)
            hipLaunchKernelGGL(chain_points_kernel, dim3((unsigned)((T + CHAIN_THREADS - 1) / CHAIN_THREADS)),
                               dim3(CHAIN_THREADS), 0, st, a);
        SFM_HIP(ctx, hipGetLastError());
    }
    int *err = w.err.as<int>();
    SFM_HIP(ctx, hipMemcpyAsync(err, a.error, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (extrinsics_out)
        SFM_HIP(ctx, hipMemcpyAsync(extrinsics_out, w.extr.p, (S + 1) * 16 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (points3d_out && T)
        SFM_HIP(ctx, hipMemcpyAsync(points3d_out, w.pts3d.p, T * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (counts_out && T)
        SFM_HIP(ctx, hipMemcpyAsync(counts_out, w.counts.p, T * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (colors_out && a.colors && T) SFM_HIP(ctx, hipMemcpyAsync(colors_out, w.cols3d.p, T * 3, hipMemcpyDeviceToHost, st));
    if (acc_scale_out && S)
        SFM_HIP(ctx, hipMemcpyAsync(acc_scale_out, w.scale.p, S * sizeof(double), hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    if (err[0]) {
        const int s = err[0] - 1;
        if (err[1] == CHAIN_ERR_RANGE) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: step %d holds a track id outside [0, %d)", s, a.nTracks);
        if (err[1] == CHAIN_ERR_TWICE) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: a track id occurs twice in step %d", s);
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: step %d has more points than correspondences", s);
    }
    return SFM_OK;
}

static int chain_common_checks(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const double *start_pose16,
                               double max_translation, int n_tracks)
{
    if (n_steps < 0 || n_tracks < 0 || (n_steps > 0 && !offsets)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: bad sizes");
    if (!start_pose16) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: null start pose");
    if (!(max_translation > 0)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: max_translation must be positive");
    if (n_steps > 0 && offsets[0] != 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: offsets must start at 0");
    for (int s = 0; s < n_steps; s++) {
        const int64_t n = offsets[s + 1] - offsets[s];
        if (n < 0 || n > 0x7FFFFFFF) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: offsets not ascending at step %d", s);
    }
    return SFM_OK;
}

static int scale_chain_host(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const int32_t *track_ids, const double *points,
                            const unsigned char *colors, const double *poses16, const double *start_pose16, double max_translation,
                            int n_tracks, double *extrinsics_out, double *points3d_out, unsigned char *colors_out,
                            int32_t *counts_out, double *acc_scale_out)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    int rc = chain_common_checks(ctx, n_steps, offsets, start_pose16, max_translation, n_tracks);
    if (rc != SFM_OK) return rc;
    const int64_t total = n_steps ? offsets[n_steps] : 0;
    if (total > 0 && (!track_ids || !points)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: null track ids or points");
    if (n_steps > 0 && !poses16) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: null poses");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ChainScratch *wp = chain_scratch(ctx);
    if (!wp) SFM_FAIL(ctx, SFM_ERR_OOM, "scale chain: host allocation failed");
    ChainScratch &w = *wp;
    const size_t S = (size_t)n_steps;
    SFM_HIP(ctx, w.offsets.reserve((S + 1) * sizeof(int64_t)));
    SFM_HIP(ctx, w.tracks.reserve((size_t)(total ? total : 1) * sizeof(int32_t)));
    SFM_HIP(ctx, w.points.reserve((size_t)(total ? total : 1) * 3 * sizeof(double)));
    SFM_HIP(ctx, w.poses.reserve((S ? S : 1) * 16 * sizeof(double)));
    if (colors) SFM_HIP(ctx, w.colors.reserve((size_t)(total ? total : 1) * 3));
    if (S) SFM_HIP(ctx, hipMemcpyAsync(w.offsets.p, offsets, (S + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (total) {
        SFM_HIP(ctx, hipMemcpyAsync(w.tracks.p, track_ids, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice, st));
        SFM_HIP(ctx, hipMemcpyAsync(w.points.p, points, (size_t)total * 3 * sizeof(double), hipMemcpyHostToDevice, st));
        if (colors) SFM_HIP(ctx, hipMemcpyAsync(w.colors.p, colors, (size_t)total * 3, hipMemcpyHostToDevice, st));
    }
    if (S) SFM_HIP(ctx, hipMemcpyAsync(w.poses.p, poses16, S * 16 * sizeof(double), hipMemcpyHostToDevice, st));
    int maxStep = 0;
    for (int s = 0; s < n_steps; s++) maxStep = (int)std::max<int64_t>(maxStep, offsets[s + 1] - offsets[s]);
    ChainArgs a{};
    a.nSteps = n_steps;
    a.nTracks = n_tracks;
    a.offsets = w.offsets.as<int64_t>();
    a.tracks = w.tracks.as<int32_t>();
    a.points = w.points.as<double>();
    a.poses = w.poses.as<double>();
    a.colors = colors ? w.colors.as<unsigned char>() : nullptr;
    a.maxTranslation = max_translation;
    memcpy(a.start, start_pose16, sizeof a.start);
    return chain_run(ctx, w, a, total, maxStep, extrinsics_out, points3d_out, counts_out, acc_scale_out, colors_out);
}

extern "C" int sfm_scale_chain(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const int32_t *track_ids,
                               const double *points, const double *poses16, const double *start_pose16,
                               double max_translation, int n_tracks, double *extrinsics_out, double *points3d_out,
                               int32_t *counts_out, double *acc_scale_out)
{
    return scale_chain_host(ctx, n_steps, offsets, track_ids, points, nullptr, poses16, start_pose16, max_translation, n_tracks,
                            extrinsics_out, points3d_out, nullptr, counts_out, acc_scale_out);
}

extern "C" int sfm_scale_chain_colors(sfm_ctx *ctx, int n_steps, const int64_t *offsets, const int32_t *track_ids,
                                      const double *points, const uint8_t *colors, const double *poses16,
                                      const double *start_pose16, double max_translation, int n_tracks,
                                      double *extrinsics_out, double *points3d_out, uint8_t *colors_out,
                                      int32_t *counts_out, double *acc_scale_out)
{
    if (ctx && n_steps > 0 && offsets && offsets[n_steps] > 0 && !colors)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: null colours");
    static const uint8_t none[3] = {0, 0, 0}; // (no observation at all: the colour path still clears its output)
    return scale_chain_host(ctx, n_steps, offsets, track_ids, points, colors ? colors : none, poses16, start_pose16, max_translation,
                            n_tracks, extrinsics_out, points3d_out, colors_out, counts_out, acc_scale_out);
}

extern "C" int sfm_pixel_bilinear_u8(sfm_ctx *ctx, const uint8_t *bgr, int rows, int cols, const double *pixels, int n,
                                     uint8_t *rgb_out)
{
    if (!ctx) return SFM_ERR_INVALID_ARG;
    if (n < 0 || rows < 2 || cols < 2 || !bgr || (n > 0 && (!pixels || !rgb_out)))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "pixel_bilinear: an image of at least 2 x 2 pixels, positions and an output are needed");
    if (n == 0) return SFM_OK;
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ChainScratch *wp = chain_scratch(ctx);
    if (!wp) SFM_FAIL(ctx, SFM_ERR_OOM, "pixel_bilinear: host allocation failed");
    ChainScratch &w = *wp;
    const size_t imgBytes = (size_t)rows * cols * 3;
    SFM_HIP(ctx, w.image.reserve(imgBytes));
    SFM_HIP(ctx, w.pixels.reserve((size_t)n * sizeof(double2)));
    SFM_HIP(ctx, w.colors.reserve((size_t)n * 3));
    SFM_HIP(ctx, hipMemcpyAsync(w.image.p, bgr, imgBytes, hipMemcpyHostToDevice, st));
    SFM_HIP(ctx, hipMemcpyAsync(w.pixels.p, pixels, (size_t)n * sizeof(double2), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pixel_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w.image.as<unsigned char>(), rows,
                       cols, w.pixels.as<double2>(), n, w.colors.as<unsigned char>());
    SFM_HIP(ctx, hipGetLastError());
    SFM_HIP(ctx, hipMemcpyAsync(rgb_out, w.colors.p, (size_t)n * 3, hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    return SFM_OK;
}

extern "C" int sfm_scale_chain_refit(sfm_ransac_batch *b, const int32_t *corr_track_ids, const double *start_pose16,
                                     double max_translation, int n_tracks, double *extrinsics_out,
                                     double *points3d_out, int32_t *counts_out, double *acc_scale_out)
{
    RefitView v;
    sfm_ctx *ctx = nullptr;
    int rc = sfm_ransac_batch_refit_view(b, &v, &ctx);
    if (rc != SFM_OK) return rc;
    rc = chain_common_checks(ctx, v.n_pairs, v.corrOffHost, start_pose16, max_translation, n_tracks);
    if (rc != SFM_OK) return rc;
    if (v.total > 0 && !corr_track_ids) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "scale chain: null track ids");
    SFM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ChainScratch *wp = chain_scratch(ctx);
    if (!wp) SFM_FAIL(ctx, SFM_ERR_OOM, "scale chain: host allocation failed");
    ChainScratch &w = *wp;
    SFM_HIP(ctx, w.tracks.reserve((size_t)(v.total ? v.total : 1) * sizeof(int32_t)));
    if (v.total)
        SFM_HIP(ctx, hipMemcpyAsync(w.tracks.p, corr_track_ids, (size_t)v.total * sizeof(int32_t), hipMemcpyHostToDevice, st));
    int maxStep = 0; // the refit uses at most all correspondences of a pair
    for (int s = 0; s < v.n_pairs; s++) maxStep = (int)std::max<int64_t>(maxStep, v.corrOffHost[s + 1] - v.corrOffHost[s]);
    ChainArgs a{};
    a.nSteps = v.n_pairs;
    a.nTracks = n_tracks;
    a.offsets = v.corrOff;
    a.tracks = w.tracks.as<int32_t>();
    a.points = v.points;
    a.poses = v.pose16;
    a.used = v.used;
    a.selCount = v.inlCount;
    a.selOff = v.inlOff;
    a.sel = v.inliers;
    a.maxTranslation = max_translation;
    memcpy(a.start, start_pose16, sizeof a.start);
    return chain_run(ctx, w, a, v.total, maxStep, extrinsics_out, points3d_out, counts_out, acc_scale_out);
}
