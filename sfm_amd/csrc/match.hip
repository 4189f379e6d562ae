// Pairwise descriptor matching on gfx950.
//
// What is computed is matchTwoSided of the reference (submodules/match/match.cpp:24-94):
// for every cell (i, j) of the nA x nB grid, sim = metric(descA_i, descB_j) in fp32
// (match.hpp:18-46); the cell passes if sim > threshold and the squared pixel distance
// (fp64) is below maxDistancePxSquared; matches2[j] / matches1[i] are the lowest-index
// arg-max over passing cells with sim > 0; the output is every (i, matches1[i]) whose
// choice is mutual.
//
// How: the N x M similarity matrix is never stored.
//   1. ingest (once per image set, ingest_rows_kernel / ingest_i8_kernel): per descriptor the fp32 squared norm in
//      the reference's summation order and its sqrtf; an fp16 copy and an int8 copy (one scale per image) of the
//      descriptor divided by that norm; per image the statistics the bounds below are made from.  They stay on
//      the device: pair_bounds_kernel turns them into each pair's bounds when a batch is launched.
//   2. match_sim_kernel bounds every cell on the matrix cores: int8 MFMA (exact int32 sums of the quantised
//      descriptors, with a rigorous bound on the quantisation error) or fp16 MFMA (good to 1.0e-3 absolute).  A
//      cell whose bound clears the threshold is a *candidate*; its (i, j) goes to the pair's candidate list.
//      Nothing else leaves the kernel.  Pairs with more candidates than the list holds are redone with the tighter
//      fp16 bound, then with a filter that keeps only cells that can be a row or column best, and as a last
//      resort scored cell by cell.  Small launches cut each row block's columns into parts (one block each).
//   3. match_cand128_kernel / match_cand_kernel: every candidate is re-evaluated exactly as the reference does it
//      (128 sequential fp32 multiply-adds without contraction, IEEE sqrt/divide, fp64 pixel
//      gate) and folded into per-row / per-column 64-bit keys (sim bits : ~index) with
//      atomicMax, which is the reference's arg-max including its lowest-index tie rule.
//   4. match_mutual_kernel: mutual check and ordered compaction into CSR.
// Since every cell that the reference could accept is a candidate, and candidates are
// scored with the reference's own arithmetic, the result is bit-identical to the
// reference's for any finite input, not only for integer-valued SIFT descriptors.
// A batch call runs in chunks of pairs on three streams of the context (first bound pass / everything after it /
// uploads of the next chunk), sizes the candidate lists of later chunks from what its first chunk needed, and
// returns before its work is done.
#include "match_types.h"

#include <initializer_list>
#include <type_traits>
#include <utility>

// Every fp32/fp64 operation of the exact pass must be the one written: no fused multiply-adds.
#pragma clang fp contract(off)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));

#define KDIM 128         // padded descriptor length of the fp16 table
#define ROW_PAD 1024     // every image's fp16 rows are padded to a multiple of this
#define SIM_MARGIN 1.5e-3f

// flags per image
#define IMG_WILD_COS 1 // a descriptor norm outside [1e-12, 1e12] (and not 0/inf/nan)
#define IMG_WILD_DOT 2 // a descriptor norm that is 0, non-finite or outside that range
#define IMG_NOT_U8 4   // an element that is not an integer in [0, 255]
#define IMG_PTS_WILD 8 // a keypoint position that is not finite


// ---------------------------------------------------------------------------------------
// ingest
// ---------------------------------------------------------------------------------------
// One thread per descriptor, the whole row in registers (all its loads in flight at once):
//  - normA += a[i]*a[i] in index order, no contraction (match.hpp:38-43), then sqrtf;
//  - the row divided by that norm, rounded to fp16 (rows whose norm is outside the safe range, and the
//    padding beyond dim, are zero);
//  - per image, over the unit-normalised rows: the largest |element| and the largest L1 norm
//    (stats[2 * image], [2 * image + 1]; bit patterns of non-negative floats, so an integer max orders them),
//    from which the int8 table takes its scale;
//  - normRange[2 * image] = 0x7FFFFFFF - bits of the smallest norm, [2 * image + 1] = bits of the largest
//    (norms inside the safe range only; both start at 0 and grow under atomicMax);
//  - the image's flags.
// A wave combines its 64 rows before it touches the per-image words.
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
// Monotone map float -> unsigned (larger float, larger key); key 0 is below every float.
__device__ __forceinline__ unsigned ordered_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_value(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, d, 64));
    return v;
}

// bbox[4 * image ..]: ordered keys of max x, max -x, max y, max -y over the image's points (as floats, which is
// what the pixel gate's "can it exclude anything at all" test needs); an image with a point that is not finite
// gets IMG_PTS_WILD instead.
// Eight lanes per row, sixteen elements each (a wave takes eight rows): the divisions, conversions and checks of a
// row are spread over its lanes, and only the squared norm runs element after element, handed from lane to
// lane, because its rounding must follow the reference's loop (match.hpp:36-45).
#define INGEST_LANES 8
#define INGEST_ELEMS (KDIM / INGEST_LANES)
#define INGEST_WAVES 8 // waves per block
template <bool VEC>
__global__ __launch_bounds__(64 * INGEST_WAVES) void ingest_rows_kernel(const float *__restrict__ desc32, const ImageRec *__restrict__ images,
                                                         int dim, float *__restrict__ norms, float *__restrict__ norms2,
                                                         int *__restrict__ imageFlags, int *__restrict__ stats,
                                                         int *__restrict__ normRange, _Float16 *__restrict__ desc16,
                                                         const double2 *__restrict__ pts, unsigned *__restrict__ bbox)
{
    static_assert(INGEST_ELEMS == 16, "two half8 stores per lane");
    constexpr int ROWS = 64 * INGEST_WAVES / INGEST_LANES; // rows per block
    const ImageRec im = images[blockIdx.y];
    const int sub = threadIdx.x & (INGEST_LANES - 1);
    const int r = blockIdx.x * ROWS + (threadIdx.x / INGEST_LANES);
    if (blockIdx.x * ROWS >= im.n) return;
    const bool valid = r < im.n;
    const int k0 = sub * INGEST_ELEMS;
    const float *a = desc32 + (size_t)(im.row32 + (valid ? r : 0)) * dim + k0;
    float v[INGEST_ELEMS];
    if constexpr (VEC) { // dim is a multiple of four: rows are 16-byte aligned
#pragma unroll
        for (int k = 0; k < INGEST_ELEMS; k += 4) {
            float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k0 + k < dim) q = *reinterpret_cast<const float4 *>(a + k);
            v[k] = q.x, v[k + 1] = q.y, v[k + 2] = q.z, v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < INGEST_ELEMS; k++) v[k] = (k0 + k < dim) ? a[k] : 0.0f;
    }
    // (the zeros beyond dim change nothing below: x + 0 = x, max(x, 0) = x for x >= 0, and 0 is a byte)
    float sq[INGEST_ELEMS];
    bool bytes = true;
#pragma unroll
    for (int k = 0; k < INGEST_ELEMS; k++) {
        sq[k] = __fmul_rn(v[k], v[k]);
        bytes = bytes && v[k] >= 0.0f && v[k] <= 255.0f && v[k] == floorf(v[k]);
    }
    // The sum in element order: in turn g every lane adds its squares to the sum so far, and the result of the
    // row's lane g - the only one whose squares belong there - is what all of the row's lanes go on with.
    float normA = 0.0f;
#pragma unroll
    for (int g = 0; g < INGEST_LANES; g++) {
        float mine = normA;
#pragma unroll
        for (int k = 0; k < INGEST_ELEMS; k++) mine = __fadd_rn(mine, sq[k]);
        normA = __shfl(mine, (int)(threadIdx.x & ~(INGEST_LANES - 1)) | g, 64);
    }
    // sqrtf, not __fsqrt_rn: the latter compiles to the bare v_sqrt_f32 (1 ulp); sqrtf is correctly rounded
    // under -fhip-fp32-correctly-rounded-divide-sqrt, like std::sqrt on the host (match.hpp:45)
    const float s = sqrtf(normA);
    const bool ok = (s >= 1e-12f) && (s <= 1e12f);
    float mx = 0.0f, l1 = 0.0f;
    _Float16 *out = desc16 + (size_t)(im.row16 + r) * KDIM + k0;
#pragma unroll
    for (int e0 = 0; e0 < INGEST_ELEMS; e0 += 8) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float q = ok ? __fdiv_rn(v[e0 + e], s) : 0.0f;
            h[e] = (_Float16)q;
            mx = fmaxf(mx, fabsf(q));
            l1 += fabsf(q);
        }
        if (valid) *reinterpret_cast<half8 *>(out + e0) = h;
    }
#pragma unroll
    for (int d = 1; d < INGEST_LANES; d <<= 1) l1 += __shfl_xor(l1, d, 64); // the row's sum of magnitudes
    int f = 0;
    unsigned bx[4] = {0, 0, 0, 0};
    if (valid && sub == 0) {
        const double2 pt = pts[im.row32 + r];
        const float px = (float)pt.x, py = (float)pt.y;
        if (fabsf(px) <= 3.0e38f && fabsf(py) <= 3.0e38f) { // (false for NaN too)
            bx[0] = ordered_key(px), bx[1] = ordered_key(-px), bx[2] = ordered_key(py), bx[3] = ordered_key(-py);
        } else {
            f |= IMG_PTS_WILD;
        }
        norms[im.row32 + r] = s;
        norms2[im.row32 + r] = normA; // the squared norm as the reference's loop accumulates it (L2 mode)
    }
    if (valid) {
        f |= bytes ? 0 : IMG_NOT_U8;
        if (!ok) {
            f |= IMG_WILD_DOT;
            if (s != 0.0f && s == s && s != INFINITY) f |= IMG_WILD_COS;
        }
    }
    const bool in = valid && ok;
    const int wMx = wave_max(in ? __float_as_int(mx) : 0);
    const int wL1 = wave_max(in ? __float_as_int(l1 * 1.0001f) : 0); // the sum's own rounding
    const int wLo = wave_max(in ? 0x7FFFFFFF - __float_as_int(s) : 0);
    const int wHi = wave_max(in ? __float_as_int(s) : 0);
    int wF = f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wF |= __shfl_xor(wF, d, 64);
#pragma unroll
    for (int e = 0; e < 4; e++) bx[e] = wave_max_u(bx[e]);
    // One set of atomics per block: the per-image words share a cache line, and atomics on one line are
    // served one after the other, so their number - not the arithmetic above - is what a small launch waits for.
    __shared__ unsigned red[INGEST_WAVES][9];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 4; e++) red[wave][e] = bx[e];
        red[wave][4] = (unsigned)wMx, red[wave][5] = (unsigned)wL1, red[wave][6] = (unsigned)wLo, red[wave][7] = (unsigned)wHi;
        red[wave][8] = (unsigned)wF;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        unsigned v9 = 0; // (every word is non-negative as an int, so unsigned and signed maxima agree)
#pragma unroll
        for (int w = 0; w < INGEST_WAVES; w++) v9 = threadIdx.x == 8 ? (v9 | red[w][threadIdx.x]) : max(v9, red[w][threadIdx.x]);
        const int e = threadIdx.x;
        if (e < 4) {
            if (v9) atomicMax(&bbox[4 * blockIdx.y + e], v9);
        } else if (e < 6) {
            if (v9) atomicMax(&stats[2 * blockIdx.y + (e - 4)], (int)v9);
        } else if (e < 8) {
            if (v9) atomicMax(&normRange[2 * blockIdx.y + (e - 6)], (int)v9);
        } else if (v9) {
            atomicOr(&imageFlags[blockIdx.y], (int)v9);
        }
    }
}

// Several buffers cleared by one launch (block row y clears range y; sizes are multiples of four bytes).
#define CLEAR_RANGES 6
struct ClearRanges {
    void *p[CLEAR_RANGES];
    unsigned long long bytes[CLEAR_RANGES];
};
__global__ __launch_bounds__(256) void clear_ranges_kernel(const ClearRanges r)
{
    unsigned char *base = reinterpret_cast<unsigned char *>(r.p[blockIdx.y]);
    const size_t bytes = r.bytes[blockIdx.y];
    const size_t toAligned = (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
    const size_t head = bytes < toAligned ? bytes : toAligned;
    const size_t n16 = (bytes - head) / 16;
    uint4 *mid = reinterpret_cast<uint4 *>(base + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) mid[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        const size_t tail0 = head + n16 * 16;
        for (size_t i = threadIdx.x * 4; i < head; i += 1024) *reinterpret_cast<unsigned *>(base + i) = 0;
        for (size_t i = tail0 + threadIdx.x * 4; i < bytes; i += 1024) *reinterpret_cast<unsigned *>(base + i) = 0;
    }
}
// Clears up to CLEAR_RANGES (pointer, bytes) ranges on a stream with one launch; ranges of zero bytes are skipped.
static hipError_t clear_ranges(hipStream_t st, std::initializer_list<std::pair<void *, size_t>> ranges)
{
    ClearRanges r;
    int n = 0;
    size_t most = 0;
    for (const auto &x : ranges) {
        if (x.second == 0) continue;
        if (n == CLEAR_RANGES || (x.second & 3)) return hipErrorInvalidValue;
        r.p[n] = x.first;
        r.bytes[n] = x.second;
        most = std::max(most, x.second);
        n++;
    }
    if (n == 0) return hipSuccess;
    const unsigned gx = (unsigned)std::min<size_t>(1024, (most / 16 + 255) / 256 + 1);
    hipLaunchKernelGGL(clear_ranges_kernel, dim3(gx, n), dim3(256), 0, st, r);
    return hipGetLastError();
}

// Byte copy of a table whose elements are all integers in [0, 255] (SIFT descriptors as OpenCV
// delivers them, detectOpenCv.cpp:166-169): the exact pass then reads a quarter of the bytes and
// computes the same fp32 values.  One thread per 16 elements.
__global__ void ingest_u8_kernel(const float *__restrict__ desc32, int64_t rows, unsigned char *__restrict__ desc8u)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * 8) return;
    const float4 *src = reinterpret_cast<const float4 *>(desc32 + t * 16);
    int4v out;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float4 v = src[w];
        out[w] = (int)((unsigned)v.x | ((unsigned)v.y << 8) | ((unsigned)v.z << 16) | ((unsigned)v.w << 24));
    }
    *reinterpret_cast<int4v *>(desc8u + t * 16) = out;
}

// The other direction, for tables handed over as bytes.
__global__ void expand_u8_kernel(const unsigned char *__restrict__ src, int64_t n, float *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = (float)src[t];
}

// Scale of an image's int8 table from the largest |element| of its unit-normalised descriptors (bit pattern).
__host__ __device__ __forceinline__ float int8_scale(int largestBits)
{
    float mx;
    memcpy(&mx, &largestBits, 4);
    return (mx > 0.0f) ? 127.0f / mx : 1.0f;
}

// One thread per 16 elements of the int8 table: round(a / norm * scale), scale from the image's statistics.
__global__ void ingest_i8_kernel(const float *__restrict__ desc32, const ImageRec *__restrict__ images,
                                 const float *__restrict__ norms, int dim, const int *__restrict__ stats,
                                 signed char *__restrict__ desc8)
{
    const ImageRec im = images[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t >> 3, k0 = (t & 7) * 16;
    if (r >= im.n) return;
    const float s = norms[im.row32 + r];
    const bool ok = (s >= 1e-12f) && (s <= 1e12f);
    const float scale = int8_scale(stats[2 * blockIdx.y]);
    const float *a = desc32 + (size_t)(im.row32 + r) * dim;
    int4v out;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        unsigned packed = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = k0 + 4 * w + e;
            float v = 0.0f;
            if (ok && k < dim) v = rintf(__fdiv_rn(a[k], s) * scale);
            const int q = (int)fminf(fmaxf(v, -127.0f), 127.0f);
            packed |= (unsigned)(q & 0xFF) << (8 * e);
        }
        out[w] = (int)packed;
    }
    *reinterpret_cast<int4v *>(desc8 + (size_t)(im.row16 + r) * 128 + k0) = out;
}

// The 6-bit table: e2m3 floats (sign, two exponent bits, three mantissa bits: 0, 1/8 ... 7/8, then 1 ... 7.5 in
// steps of 1/8, 1/4 and 1/2), which the matrix cores multiply at twice the int8 rate.  One scale per image, a
// power of two, puts the image's largest |element| into [3.75, 7.5].
__host__ __device__ __forceinline__ float f6_scale(int largestBits)
{
    float mx;
    memcpy(&mx, &largestBits, 4);
    if (!(mx > 0.0f) || !(mx < 1e30f)) return 1.0f;
    int e;
    const float m = frexpf(mx, &e); // mx = m 2^e, m in [0.5, 1)
    return ldexpf(1.0f, m <= 0.9375f ? e - 3 : e - 2);
}
// x in [0, 7.5] to the nearest e2m3 value (ties to even, as rintf rounds).
__device__ __forceinline__ float f6_round(float x)
{
    const float step = x < 2.0f ? 0.125f : (x < 4.0f ? 0.25f : 0.5f);
    return fminf(rintf(x / step) * step, 7.5f);
}
__device__ __forceinline__ unsigned f6_bits(float v) // v: an e2m3 value (sign included)
{
    const unsigned sign = v < 0.0f ? 32u : 0u;
    const float x = fabsf(v);
    if (x < 1.0f) return sign | (unsigned)(x * 8.0f);
    const int e = x < 2.0f ? 1 : (x < 4.0f ? 2 : 3);
    return sign | ((unsigned)e << 3) | (unsigned)((x / (float)(1 << (e - 1)) - 1.0f) * 8.0f);
}
// One thread per 32 elements (a k-block of the 16x16x128 MFMA: 32 x 6 bits = six dwords per lane).  A row of 128 B
// holds the first four dwords of its four k-blocks in its first 64 bytes and the last two of each, padded to 16
// bytes, in its second: the 16-byte chunk a lane reads for k-block kb is chunk kb, then chunk 4 + kb -- the same
// addresses as the two k-blocks of an int8 row, so the kernel stages and reads both tables alike.
// err6[image] = the largest |a_hat - copy| over the image's rows (float bits, rounded up): what the bound on a
// pair's approximation error is made of.
__global__ __launch_bounds__(256) void ingest_f6_kernel(const float *__restrict__ desc32, const ImageRec *__restrict__ images,
                                                        const float *__restrict__ norms, int dim, const int *__restrict__ stats,
                                                        unsigned char *__restrict__ desc6, int *__restrict__ err6)
{
    const ImageRec im = images[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = t >> 2, kb = t & 3;
    if ((blockIdx.x * blockDim.x) >> 2 >= im.n) return;
    const bool valid = r < im.n;
    float sq = 0.0f;
    if (valid) {
        const float s = norms[im.row32 + r];
        const bool ok = (s >= 1e-12f) && (s <= 1e12f);
        const float scale = f6_scale(stats[2 * blockIdx.y]);
        const float *a = desc32 + (size_t)(im.row32 + r) * dim;
        unsigned w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int k = kb * 32 + j;
            float v = 0.0f;
            if (ok && k < dim) v = __fdiv_rn(a[k], s);
            float q = f6_round(fminf(fabsf(v) / scale, 7.5f));
            if (!(q == q)) q = 0.0f;
            const float back = copysignf(q * scale, v);
            const float d = back - v;
            if (d == d) sq = fmaf(d, d, sq);
            const unsigned bits = f6_bits(copysignf(q, v));
            const int at = 6 * j;
            w[at >> 5] |= bits << (at & 31);
            if ((at & 31) > 26) w[(at >> 5) + 1] |= bits >> (32 - (at & 31));
        }
        unsigned char *row = desc6 + (size_t)(im.row16 + r) * 128;
        *reinterpret_cast<uint4 *>(row + 16 * kb) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4 *>(row + 64 + 16 * kb) = make_uint4(w[4], w[5], 0u, 0u);
    }
    // the row's four threads are neighbours in the wave
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    // (the sum of 128 squares is good to a few 1e-6 of itself, the square root to one in 2^23: one part in a thousand covers both)
    const float e = sqrtf(sq) * 1.001f + 1e-7f;
    const int worst = wave_max(valid ? __float_as_int(e) : 0);
    if ((threadIdx.x & 63) == 0 && worst > 0) atomicMax(&err6[blockIdx.y], worst);
}

// ---------------------------------------------------------------------------------------
// per-pair candidate bounds
// ---------------------------------------------------------------------------------------
// One thread per pair of a chunk: the bound on the approximate cosine (thr_m) and on the int8 sum (thr_i)
// from the two images' statistics, which never leave the device.
struct BoundArgs {
    const int *stats; // [n][2] largest |element| and L1 norm, then [n][2] norm range (bit patterns, see sfm_imageset)
    const int *flags; // [n] IMG_*
    const unsigned *bbox; // [n][4] ordered keys of max x, max -x, max y, max -y
    int nImages;
    int dim;
    int metric;
    float threshold;
    float ratio;
    float maxDistSq;
    int dotStrict; // dot metric: make a cosine bound for the strict passes (see below)
    const int *err6; // [n] largest distance between a unit-normalised descriptor and its e2m3 copy (float bits), or null: no such table
};
__global__ __launch_bounds__(256) void pair_bounds_kernel(PairRec *__restrict__ pairs, int nPairs, const BoundArgs b)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nPairs) return;
    const int ia = pairs[p].ia, ib = pairs[p].ib;
    const float threshold = b.threshold, ratio = b.ratio;
    const int wildMask = (b.metric == SFM_METRIC_COSINE) ? IMG_WILD_COS : IMG_WILD_DOT;
    bool wild = ((b.flags[ia] | b.flags[ib]) & wildMask) != 0;
    float thrp = threshold > 0.0f ? threshold : 0.0f;
    if (b.metric == SFM_METRIC_L2) {
        // d^2 < T  =>  a.b > (|a|^2 + |b|^2 - T) / 2  =>  cos > 1 - T / (2 |a| |b|)   (|a|^2 + |b|^2 >= 2 |a| |b|)
        //          >=  1 - T / (2 minA minB).  The fp32 evaluation of d^2 is within 1e-4 (maxA^2 + maxB^2) of the
        // exact one; the ratio mode collects down to T / ratio^2 (a rival at d2 undoes a best at d1 < T
        // only if d2 <= d1 / ratio^2).
        const int *range = b.stats + 2 * (size_t)b.nImages;
        const double minA = (double)__int_as_float(0x7FFFFFFF - range[2 * ia]), maxA = (double)__int_as_float(range[2 * ia + 1]);
        const double minB = (double)__int_as_float(0x7FFFFFFF - range[2 * ib]), maxB = (double)__int_as_float(range[2 * ib + 1]);
        const double lo = minA * minB; // NaN when an image has no descriptor inside the safe range
        const double hi2 = maxA * maxA + maxB * maxB;
        double T = (double)threshold;
        if (ratio > 0.0f) T = T / ((double)ratio * ratio);
        const double c = 1.0 - (T + 1e-4 * hi2) / (2.0 * lo) - 1e-6;
        const bool open = !(lo > 0.0) || !(c == c) || c < -1.5; // no cosine bound holds for this pair
        thrp = open ? 0.0f : (float)c;
        wild = wild || open;
    }
    const bool dotStrict = b.metric == SFM_METRIC_DOT && b.dotStrict != 0;
    if (dotStrict && thrp > 0.0f) {
        // a.b > T > 0  =>  cos > T / (|a| |b|) >= T / (maxA maxB): the strict passes bound the cosine, the exact
        // pass scores the dot product itself.  (The reference's fp32 dot product is within 8e-6 |a| |b| of the
        // exact one, which the margins of both bounds cover.)  No descriptor inside the safe range: no bound.
        const int *range = b.stats + 2 * (size_t)b.nImages;
        const double maxA = (double)__int_as_float(range[2 * ia + 1]), maxB = (double)__int_as_float(range[2 * ib + 1]);
        const double c = (double)thrp / (maxA * maxB) * (1.0 - 1e-6) - 1e-6;
        const bool open = !(maxA * maxB > 0.0) || !(c == c);
        thrp = open ? 0.0f : (float)(c < 4.0 ? c : 4.0); // (beyond 1 nothing passes, as it must not)
        wild = wild || open;
    }
    // Ratio mode: a row whose best has similarity s1 > threshold is undone by any column at or
    // above 1 - (1 - s1) / ratio^2, which is more than 1 - (1 - threshold) / ratio^2: the bound
    // pass collects down to there, and the exact pass still applies the threshold itself.
    if (b.metric != SFM_METRIC_L2 && ratio > 0.0f && threshold == threshold) {
        const float t = (float)(1.0 - (1.0 - (double)thrp) / ((double)ratio * ratio) - 1e-6);
        thrp = t > -2.0f ? t : -2.0f;
    }
    float thr_m;
    if (wild)
        thr_m = -INFINITY;
    else if (b.metric == SFM_METRIC_DOT && !dotStrict)
        thr_m = (thrp == 0.0f) ? -SIM_MARGIN : 0.0f; // a positive bound on the dot product is tested per cell
    else
        thr_m = (threshold != threshold) ? threshold : thrp - SIM_MARGIN; // (L2: thrp may be negative here)
    // Bound on the int8 dot product S = sum q_k p_k, q = round(a_hat * sA), p = round(b_hat * sB):
    // |a_hat.b_hat * sA * sB - S| <= 0.5001 (sA L1_A + sB L1_B) + 0.7505 dim  (rounding errors
    // e, f with |e|, |f| <= 0.5001 against |q| <= sA |a_hat| + 0.5001), and the reference's cosine
    // is within 1e-4 of a_hat.b_hat.
    const double sA = (double)int8_scale(b.stats[2 * ia]), sB = (double)int8_scale(b.stats[2 * ib]);
    const double l1A = (double)__int_as_float(b.stats[2 * ia + 1]), l1B = (double)__int_as_float(b.stats[2 * ib + 1]);
    const double err = 0.5001 * (sA * l1A + sB * l1B) + 0.7505 * b.dim;
    const double t = ((double)thrp - 1e-4) * sA * sB - err;
    int thr_i;
    if (threshold != threshold)
        thr_i = INT_MAX;
    else if (wild || (b.metric == SFM_METRIC_DOT && !dotStrict) || !(t > -1e9))
        thr_i = -(1 << 30);
    else
        thr_i = (int)floor(t < 2.0e9 ? t : 2.0e9);
    // Bound on the e2m3 dot product S6 = qa.qb with qa = (a_hat + e) / sA, qb = (b_hat + f) / sB, |e| <= EA, |f| <= EB:
    // |a_hat.b_hat - S6 sA sB| = |e.b_hat + (a_hat + e).f| <= EA |b_hat| + (|a_hat| + EA) EB, with both norms below
    // 1 + 1e-6.  The products of the MFMA are exact and its 128-term fp32 sum is good to 128 * 2^-23 of the sum of
    // their magnitudes (at most (1 + EA)(1 + EB) in cosine units): 1e-4 covers that, and another 1e-4 the
    // distance between a_hat.b_hat and the reference's cosine.
    float thr6 = -INFINITY;
    if (b.err6 && !(wild || (b.metric == SFM_METRIC_DOT && !dotStrict))) {
        const double s6A = (double)f6_scale(b.stats[2 * ia]), s6B = (double)f6_scale(b.stats[2 * ib]);
        const double EA = (double)__int_as_float(b.err6[ia]), EB = (double)__int_as_float(b.err6[ib]);
        const double margin = EA * 1.000001 + (1.000001 + EA) * EB;
        const double t6 = ((double)thrp - 2e-4 - margin * (1.0 + 1e-6)) / (s6A * s6B);
        thr6 = (threshold != threshold) ? threshold : (float)(t6 - fabs(t6) * 1e-6); // (the conversion may round up)
    }
    pairs[p].thr_m = thr_m;
    pairs[p].thr_i = thr_i;
    pairs[p].thr6 = thr6;
    // Can the pixel gate (match.cpp:45) exclude any cell of this pair?  Not if the two images' points together
    // span less than the gate (the default, 10000^2, on any image below 7000 x 7000).
    bool gateOff = ((b.flags[ia] | b.flags[ib]) & IMG_PTS_WILD) == 0;
    double ext2 = 0.0;
#pragma unroll
    for (int axis = 0; axis < 2; axis++) {
        const unsigned hiA = b.bbox[4 * ia + 2 * axis], loA = b.bbox[4 * ia + 2 * axis + 1];
        const unsigned hiB = b.bbox[4 * ib + 2 * axis], loB = b.bbox[4 * ib + 2 * axis + 1];
        if (hiA == 0 || hiB == 0 || loA == 0 || loB == 0) gateOff = false; // an image without points
        const double hi = (double)ordered_value(max(hiA, hiB)), lo = -(double)ordered_value(max(loA, loB));
        ext2 += (hi - lo) * (hi - lo);
    }
    gateOff = gateOff && (double)b.maxDistSq > ext2 * 1.0001 + 1e-3; // (the points were rounded to float: 6e-8 relative)
    pairs[p].gateOff = gateOff ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// exact evaluation of one cell, as the reference computes it
// ---------------------------------------------------------------------------------------
// The part of a cell's evaluation after the dot product: metric, threshold, pixel gate, arg-max keys.
__device__ __forceinline__ void fold_dot(const MatchArgs &g, const PairRec &pr, int i, int j, float dot)
{
    if (g.metric == SFM_METRIC_L2) {
        // Extension (no counterpart in the reference): squared Euclidean distance in the expanded form
        // |a|^2 + |b|^2 - 2 a.b, every term an fp32 sum in index order; a cell is admissible below the
        // caller's maximum; rows and columns keep their nearest admissible cell, the lowest index among equals.
        float d2 = __fsub_rn(__fadd_rn(g.norms2[pr.rowA32 + i], g.norms2[pr.rowB32 + j]), __fmul_rn(2.0f, dot));
        if (d2 < 0.0f) d2 = 0.0f;
        if (!(d2 < g.threshold)) return;
        const double2 qa = g.pts[pr.rowA32 + i], qb = g.pts[pr.rowB32 + j];
        const double ex = __dsub_rn(qa.x, qb.x), ey = __dsub_rn(qa.y, qb.y);
        if (!(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)) < (double)g.maxDistSq)) return;
        const u64 key = (u64)(~__float_as_uint(d2)) << 32; // non-negative floats order like their bits: smaller distance, larger key
        atomicMax(&g.best[pr.bestRow + i], key | (u64)(0xFFFFFFFFu - (unsigned)j));
        atomicMax(&g.best[pr.bestCol + j], key | (u64)(0xFFFFFFFFu - (unsigned)i));
        return;
    }
    float sim = dot; // match.hpp:18-27
    if (g.metric == SFM_METRIC_COSINE) // match.hpp:45
        sim = __fdiv_rn(dot, __fmul_rn(g.norms[pr.rowA32 + i], g.norms[pr.rowB32 + j]));
    if (!(sim > g.threshold)) return; // match.cpp:45
    const double2 pa = g.pts[pr.rowA32 + i], pb = g.pts[pr.rowB32 + j];
    const double dx = __dsub_rn(pa.x, pb.x), dy = __dsub_rn(pa.y, pb.y);
    const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    if (!(d2 < (double)g.maxDistSq)) return;
    if (!(sim > 0.0f)) return; // match.cpp:38,49 / :68,72: must beat maxSimilarity = 0
    // max sim first, then lowest index: the strict '>' scans of match.cpp:49 and :72.
    const u64 bits = (u64)__float_as_uint(sim) << 32;
    atomicMax(&g.best[pr.bestRow + i], bits | (u64)(0xFFFFFFFFu - (unsigned)j));
    atomicMax(&g.best[pr.bestCol + j], bits | (u64)(0xFFFFFFFFu - (unsigned)i));
}

__device__ __forceinline__ void fold_cell(const MatchArgs &g, const PairRec &pr, int i, int j)
{
    const float *a = g.desc32 + (size_t)(pr.rowA32 + i) * g.dim;
    const float *b = g.desc32 + (size_t)(pr.rowB32 + j) * g.dim;
    float dot = 0.0f;
    if ((g.dim & 3) == 0) {
        for (int k = 0; k < g.dim; k += 4) {
            const float4 av = *reinterpret_cast<const float4 *>(a + k);
            const float4 bv = *reinterpret_cast<const float4 *>(b + k);
            dot = __fadd_rn(dot, __fmul_rn(av.x, bv.x));
            dot = __fadd_rn(dot, __fmul_rn(av.y, bv.y));
            dot = __fadd_rn(dot, __fmul_rn(av.z, bv.z));
            dot = __fadd_rn(dot, __fmul_rn(av.w, bv.w));
        }
    } else {
        for (int k = 0; k < g.dim; k++) dot = __fadd_rn(dot, __fmul_rn(a[k], b[k]));
    }
    fold_dot(g, pr, i, j, dot);
}

// Ratio mode, second sweep over the candidates (the row bests are final by then).  Row i with best
// column j1 of similarity s1 keeps its match only if every other admissible column (inside the pixel
// gate) stays clear of it on the unit sphere: with d^2 = 2 - 2 sim the test d1 < ratio * d2 reads
// (1 - s1) < ratio^2 (1 - s2).  The condition falls with s2, so testing every column is testing the
// second best.
__device__ __forceinline__ void ratio_fold(const MatchArgs &g, const PairRec &pr, int i, int j, float dot)
{
    const u64 key = g.best[pr.bestRow + i];
    if (key == 0) return;
    if ((int)(0xFFFFFFFFu - (unsigned)key) == j) return;
    if (g.metric == SFM_METRIC_L2) { // d1^2 < ratio^2 d2^2 against every other column inside the pixel gate
        float d2 = __fsub_rn(__fadd_rn(g.norms2[pr.rowA32 + i], g.norms2[pr.rowB32 + j]), __fmul_rn(2.0f, dot));
        if (d2 < 0.0f) d2 = 0.0f;
        if (!(d2 == d2)) return;
        const double2 qa = g.pts[pr.rowA32 + i], qb = g.pts[pr.rowB32 + j];
        const double ex = __dsub_rn(qa.x, qb.x), ey = __dsub_rn(qa.y, qb.y);
        if (!(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)) < (double)g.maxDistSq)) return;
        const double d1 = (double)__uint_as_float(~(unsigned)(key >> 32));
        if (!(d1 < __dmul_rn(g.ratioSq, (double)d2))) g.kill[pr.bestRow + i] = 1;
        return;
    }
    const float sim = __fdiv_rn(dot, __fmul_rn(g.norms[pr.rowA32 + i], g.norms[pr.rowB32 + j]));
    if (!(sim == sim)) return;
    const double2 pa = g.pts[pr.rowA32 + i], pb = g.pts[pr.rowB32 + j];
    const double dx = __dsub_rn(pa.x, pb.x), dy = __dsub_rn(pa.y, pb.y);
    const double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    if (!(d2 < (double)g.maxDistSq)) return;
    const double s1 = (double)__uint_as_float((unsigned)(key >> 32));
    if (!(__dsub_rn(1.0, s1) < __dmul_rn(g.ratioSq, __dsub_rn(1.0, (double)sim)))) g.kill[pr.bestRow + i] = 1;
}

__device__ __forceinline__ float exact_dot(const MatchArgs &g, const PairRec &pr, int i, int j)
{
    const float *a = g.desc32 + (size_t)(pr.rowA32 + i) * g.dim;
    const float *b = g.desc32 + (size_t)(pr.rowB32 + j) * g.dim;
    float dot = 0.0f;
    for (int k = 0; k < g.dim; k++) dot = __fadd_rn(dot, __fmul_rn(a[k], b[k]));
    return dot;
}

// A zero the optimiser cannot see through.  Index arithmetic of the rare candidate path is
// made to depend on it so that none of it is hoisted into the main loop's register budget.
__device__ __forceinline__ int opaque_zero()
{
    int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#endif
    return z;
}

// ---------------------------------------------------------------------------------------
// similarity kernel
// ---------------------------------------------------------------------------------------
// Block = SIM_WAVES waves and owns SIM_BLOCK_ROWS rows of image A: wave w keeps 64 rows x 128 k as
// MFMA A-operand fragments in registers for the whole block; image B streams through LDS,
// SIM_BN descriptors per step, double buffered, one barrier per step.
//
// Two arithmetic forms of the bound:
//   fp16 (v_mfma_f32_16x16x32_f16) on the unit-normalised fp16 table, rows of 256 B;
//   int8 (v_mfma_i32_16x16x64_i8) on the quantised table, rows of 128 B, exact int32 sums.
// Operand map of both: lane l holds A[row l & 15][k-group l >> 4] and B[k-group l >> 4][col l & 15]
// (8 halves or 16 bytes per group); result register q of lane l is C[4 (l >> 4) + q][l & 15].
#define SIM_WAVES 4
#ifndef MATCH_SETS
#define MATCH_SETS 3 // sets of per-chunk buffers a batch call takes in turn
#endif
#define EV_CLEAR MATCH_SETS          // events of the context: [0, MATCH_SETS) a chunk's bound passes are done, then its set is clear again,
#define EV_JOIN (2 * MATCH_SETS)     // fork / join of the streams,
#define EV_UP (2 * MATCH_SETS + 1)   // a chunk's records are on the device
static_assert(3 * MATCH_SETS + 1 <= SFM_CTX_EVENTS, "events per context");
#ifndef SIX_BN
#define SIX_BN 64 // columns per step of the e2m3 instance
#endif
#ifndef SIX_OCC
#define SIX_OCC 5 // blocks per CU the e2m3 instance is compiled for
#endif
#define SIM_WAVE_ROWS 64
#define SIM_BLOCK_ROWS (SIM_WAVES * SIM_WAVE_ROWS)
#define SIM_BN 64

// Candidates leave the kernel through per-wave queues in global memory: every wave of every
// block owns a fixed slice of its pair's candidate list and appends with plain stores at a
// wave-uniform cursor, so the hot kernel has no atomics with return values and never waits on
// memory in the candidate path.  A slice that fills up flags the pair for the next, more
// selective pass.
struct WaveQueue {
    uint2 *base;
    int cap;
    int n; // wave-uniform cursor (may run past cap: then the pair is flagged)
    int abl; // timing-only ablations (wrong results): 4 = entries are not stored, 8 = hits are not queued at all
};

__device__ __forceinline__ void queue_push(WaveQueue &wq, bool take, int i, int j, int lane)
{
    const u64 em = __builtin_amdgcn_ballot_w64(take);
    if (em == 0) return;
    const int slot = wq.n + __builtin_popcountll(em & ((1ull << lane) - 1ull));
    if (take && slot < wq.cap && !(wq.abl & 4)) wq.base[slot] = make_uint2((unsigned)i, (unsigned)j);
    wq.n += __builtin_popcountll(em);
}

// Accumulators of one 32-column step of a wave (64 rows): 2 column tiles x 4 row tiles of 16x16.
// A lane's 32 values are numbered idx = 16 ct + 4 t + q: column 16 ct + (l & 15), row
// 16 t + 4 (l >> 4) + q.  The two groups of 16 (ct = 0, 1) share a column each.
typedef float float4v __attribute__((ext_vector_type(4)));
template <typename V4, typename S> struct SimAcc {
    typedef S VT;
    static constexpr int NV = 32, NGROUP = 2, GROUP = 16;
    V4 a[2][4];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) a[c][t][q] = S(0);
    }
    __device__ __forceinline__ S get(int idx) const { return a[idx >> 4][(idx & 15) >> 2][idx & 3]; }
    __device__ static __forceinline__ int row(int idx, int lane) { return ((idx & 15) >> 2) * 16 + 4 * (lane >> 4) + (idx & 3); }
    __device__ static __forceinline__ int col(int idx, int lane) { return (idx >> 4) * 16 + (lane & 15); }
    __device__ __forceinline__ void keep_alive() const // for the ablation without a candidate test
    {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int t = 0; t < 4; t++) asm volatile("" ::"v"(a[c][t]));
#endif
    }
};

// What the general instance keeps in LDS besides the staged descriptors: the pixel positions of
// the block's rows and of the staged columns, and one running bound per row.
struct GeneralLds {
    const double2 *ptsRow; // this wave's rows
    const double2 *ptsCol; // the 32 columns of the step
    unsigned *bound;       // this wave's rows: largest lower bound so far (ordered key, 0 = none)
};

// Candidate test of one 32-column step and the rare emission path.  colBase = first column of
// the step.
//
// Strict instances (GENERAL = false) queue every cell above the bound; a slice that overflows
// flags the pair, and a more selective pass then redoes the flagged pairs (int8 bound -> fp16
// bound -> general instance; the dot metric always takes the general instance).
//
// The general instance (loose thresholds, where most cells clear the bound) runs twice per pair,
// once with each image in the register-resident role.  A wave streams all columns past its rows
// in order, so "can this cell still be the best of its row" is a running-maximum test that needs
// no communication outside the wave: a cell inside the pixel gate is queued only if its upper
// bound reaches the largest lower bound seen so far in its row.  The true arg-max of the row
// always passes: its upper bound is at least its exact value, which is at least every other
// cell's exact value, which is at least that cell's lower bound; cells outside the gate
// (match.cpp:45) never raise a bound.  Pass 0 thus queues every row best and pass 1 every column
// best.  The bounds live in LDS and are raised with an atomic max whose returned previous value
// decides the test, so the lanes that hold one row in the same step see each other in some
// serial order.  localLo[idx] is a per-lane copy of the row's bound minus the margin, in
// registers, so that most cells are rejected by a register compare; it only ever trails the
// shared bound, which keeps the test permissive.
// SWAP (strict instances that run with either image in the register-resident role): rows are image B's when
// g.orient is set, and the pairs go out as (row of A, row of B) all the same.
// Strict instances: four independent chains of maxima over a lane's 32 accumulators (chain k takes accumulators
// 8 m + 2 k and 8 m + 2 k + 1) keep the dependent-issue latency short; no branch, so the chains can be scheduled
// among MFMAs.  fmaxf ignores NaNs, and a NaN cell can never be accepted.
template <typename AccT> __device__ __forceinline__ void strict_maxima(const AccT &acc, typename AccT::VT (&m)[4])
{
    constexpr bool INTEGER = std::is_same<typename AccT::VT, int>::value;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (INTEGER) m[k] = INT_MIN;
        else m[k] = -INFINITY;
    }
#pragma unroll
    for (int idx = 0; idx < AccT::NV; idx += 2) {
        if constexpr (INTEGER) m[(idx >> 1) & 3] = max(max(m[(idx >> 1) & 3], acc.get(idx)), acc.get(idx + 1));
        else m[(idx >> 1) & 3] = fmaxf(fmaxf(m[(idx >> 1) & 3], acc.get(idx)), acc.get(idx + 1));
    }
}

// m: the strict instances' chain maxima of acc (strict_maxima; the general instance does not look at them).
template <typename AccT, bool DOT, bool GENERAL, bool SWAP = false>
__device__ __forceinline__ void sim_epilogue(const MatchArgs &g, const PairRec &pr, const AccT &acc, int rowBase,
                                             int colBase, int lane, typename AccT::VT thr_m, WaveQueue &wq,
                                             float (&localLo)[GENERAL ? AccT::NV : 1], const GeneralLds &gl,
                                             const typename AccT::VT (&m)[4])
{
    constexpr int NV = AccT::NV;
    constexpr bool INTEGER = std::is_same<typename AccT::VT, int>::value;
    static_assert(!INTEGER || !GENERAL, "the int8 bound has a strict instance only");
    unsigned bits = 0;
    if constexpr (INTEGER) {
        bits = max(max(m[0], m[1]), max(m[2], m[3])) > thr_m;
    } else if constexpr (!GENERAL) {
        bits = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])) > thr_m;
    } else if (DOT) {
        bits = 0xFFFFFFFFu;
    } else {
#pragma unroll
        for (int idx = 0; idx < NV; idx++) {
            const float v = acc.get(idx);
            bits |= (unsigned)(v > thr_m && v >= localLo[idx]) << idx;
        }
    }
    const u64 lanesWithHits = __builtin_amdgcn_ballot_w64(bits != 0);
    if (lanesWithHits == 0) return;

    // rare path
    const int z = opaque_zero();
    const int ln = lane + z;
    if constexpr (!INTEGER) {
        if (!(thr_m > -INFINITY)) { // norms out of range: no bound holds, score the pair cell by cell
            wq.n = wq.cap + 1;
            return;
        }
    }

    bool swapped = false;
    if constexpr (!GENERAL) {
        // Every cell above the bound is queued.  The chain maxima say which quarter of the accumulators holds a
        // hit (typically one: a true match); its eight are compared one by one, the lane masks tested on the
        // scalar side, and only an accumulator with a hit somewhere in the wave pays for the queue.
        // That is the cheap way while hits are sparse (true matches).  Where they are dense (a flooded pair
        // being redone: several lanes with hits in the same step) each lane collects its 32 cells in a mask and
        // the wave walks the masks together.
        const bool sw = SWAP && g.orient != 0;
        if (!INTEGER && __builtin_popcountll(lanesWithHits) > 2) { // (a flooded int8 pass gives up at once: sparse form only)
            bits = 0;
#pragma unroll
            for (int idx = 0; idx < NV; idx++) bits |= (unsigned)(acc.get(idx) > thr_m) << idx;
            while (__builtin_amdgcn_ballot_w64(bits != 0) != 0) {
                const bool has = bits != 0;
                const int bit = has ? __builtin_ctz(bits) : 0;
                bits &= bits - 1;
                const int i = rowBase + AccT::row(bit, ln), jj = colBase + AccT::col(bit, ln);
                queue_push(wq, has && i < pr.nA && jj < pr.nB, sw ? jj : i, sw ? i : jj, lane);
            }
            return;
        }
        const int rowLane = rowBase + 4 * (ln >> 4), colLane = colBase + (ln & 15);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (__builtin_amdgcn_ballot_w64(m[k] > thr_m) == 0) continue;
            if (wq.abl & 8) { wq.n += 1; continue; } // timing only
#pragma unroll
            for (int e = 0; e < NV / 4; e++) {
                const int idx = (e >> 1) * 8 + 2 * k + (e & 1);
                const bool hit = acc.get(idx) > thr_m;
                if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;
                const int i = rowLane + ((idx & 15) >> 2) * 16 + (idx & 3), jj = colLane + (idx >> 4) * 16;
                queue_push(wq, hit && i < pr.nA && jj < pr.nB, sw ? jj : i, sw ? i : jj, lane);
            }
        }
        return;
    } else {
        const float thrp = fmaxf(g.threshold, 0.0f);
        swapped = g.orient != 0;
        unsigned takeBits = 0;
#pragma unroll
        for (int grp = 0; grp < AccT::NGROUP; grp++) {
            constexpr unsigned groupMask = (1u << AccT::GROUP) - 1u;
            const int cj = AccT::col(grp * AccT::GROUP, ln);
            const int jj = colBase + cj;
            const bool colValid = jj < pr.nB;
            const unsigned mine = colValid ? (bits >> (grp * AccT::GROUP)) & groupMask : 0u;
            if (__builtin_amdgcn_ballot_w64(mine != 0) == 0) continue;
            const float sB = (DOT && colValid) ? g.norms[pr.rowB32 + jj] : 1.0f;
            const double2 pb = gl.ptsCol[cj];
            unsigned prev[AccT::GROUP];
            unsigned did = 0;
#pragma unroll
            for (int k = 0; k < AccT::GROUP; k++) {
                const int idx = grp * AccT::GROUP + k;
                const int ri = AccT::row(idx, ln);
                prev[k] = 0;
                if (((mine >> k) & 1u) && rowBase + ri < pr.nA) {
                    float lo = acc.get(idx) - SIM_MARGIN;
                    const double2 pa = gl.ptsRow[ri];
                    const double dx = __dsub_rn(pa.x, pb.x), dy = __dsub_rn(pa.y, pb.y);
                    bool cnd = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)) < (double)g.maxDistSq;
                    if (DOT) { // dot = approx * sA * sB within 1e-3 * sA * sB
                        const float p = g.norms[pr.rowA32 + rowBase + ri] * sB;
                        lo *= p;
                        cnd = cnd && (acc.get(idx) + SIM_MARGIN) * p > thrp;
                    }
                    if (cnd) {
                        did |= 1u << k;
                        prev[k] = atomicMax(&gl.bound[ri], ordered_key(lo));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < AccT::GROUP; k++) {
                const int idx = grp * AccT::GROUP + k;
                if ((did >> k) & 1u) {
                    float lo = acc.get(idx) - SIM_MARGIN, hi = acc.get(idx) + SIM_MARGIN;
                    if (DOT) {
                        const float p = g.norms[pr.rowA32 + rowBase + AccT::row(idx, ln)] * sB;
                        lo *= p;
                        hi *= p;
                    }
                    const float rowLo = prev[k] ? ordered_value(prev[k]) : -INFINITY;
                    if (hi >= rowLo) takeBits |= 1u << idx;
                    // hi >= bound  <=>  v >= bound - margin (cosine; the dot metric has no local copy)
                    if (!DOT) localLo[idx] = fmaxf(localLo[idx], fmaxf(lo, rowLo) - SIM_MARGIN);
                }
            }
        }
        bits = takeBits;
    }
    while (__builtin_amdgcn_ballot_w64(bits != 0) != 0) {
        const bool has = bits != 0;
        const int bit = has ? __builtin_ctz(bits) : 0;
        bits &= bits - 1;
        const int i = rowBase + AccT::row(bit, ln);
        const int jj = colBase + AccT::col(bit, ln);
        queue_push(wq, has && i < pr.nA && jj < pr.nB, swapped ? jj : i, swapped ? i : jj, lane);
    }
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Image B reaches LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, written
// linearly at a wave-uniform base), so the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free is applied to the per-lane source address:
//   fp16: 16-B chunk c of row r sits at chunk c ^ (r & 15) of its 256-B LDS line;
//   int8: a 256-B LDS line holds rows 2L and 2L + 1; chunk c of row 2L + b sits at (8b + c) ^ (L & 15).
// OCC = blocks per CU the register budget is set for.  FORM selects the arithmetic (the int8 forms have strict instances only).
// ABL: timing-only ablations with wrong results: bit 0 = no staging and no barrier, bit 1 = no
// candidate test.
// MODE (fp16 strict instances only): the two-phase form of the filtered passes, for pairs whose pixel gate
// excludes nothing.  A row's best cell is then its best over all columns, so
//   1 = a sweep that only finds the largest approximate cosine of every row and of every column (no
//       candidates) and leaves their bars, max(that maximum - 2 margins, the pair's bound): the rows' from the
//       wave's registers, the columns' through maxima in LDS per block and step and atomic maxima in global
//       memory per pair (match_col_bars_kernel turns those into bars); and
//   2 = one strict sweep whose accumulators start at minus the lower of the cell's two bars (its row's, in
//       registers; its column's, staged beside the column's descriptors): what stays positive is within reach
//       of its row's best or of its column's best and is queued.
// They replace the filtered pass, which has to carry running bounds and pixel positions along, queues the
// harmonic number of cells per row, and runs twice.
template <int FORM, bool DOT, bool GENERAL, int ABL, int MODE = 0>
__device__ __forceinline__ void sim_block(const MatchArgs &g, const int workItem)
{
    static_assert(MODE == 0 || (FORM == 0 && !DOT && !GENERAL), "the two-phase sweeps are fp16 strict instances");
    constexpr bool TWOSIDED = GENERAL || MODE != 0; // runs in both orientations, slices of both in the pair's list
    // FORM 0: fp16 rows of 256 B; 1: int8 rows of 128 B; 2: int8 rows of 256 B (256 signs of a binary descriptor);
    // 3: e2m3 rows of 128 B (one 16x16x128 MFMA per tile, fp32 sums)
    constexpr bool I8 = FORM == 1 || FORM == 2; // int8 MFMA, int32 sums
    constexpr bool F6 = FORM == 3;
    constexpr bool WIDE = FORM == 0 || FORM == 2; // 256-byte rows: the fp16 table's layout in memory and in LDS
    constexpr int ROWB = WIDE ? 256 : 128;                   // bytes per descriptor row in the table and in LDS
    constexpr int BN = F6 ? SIX_BN : SIM_BN;              // columns per step
    constexpr int DMA = BN * ROWB / 1024 / SIM_WAVES;  // LDS-DMA instructions per wave per step
    static_assert(BN * ROWB % (1024 * SIM_WAVES) == 0 && (BN == 64 || (F6 && BN % 32 == 0)), "step must split evenly");
    static_assert(!(I8 || F6) || (!GENERAL && !DOT), "the int8 and e2m3 bounds have a strict cosine instance only");
    static_assert(GENERAL || !DOT, "the dot metric needs the general instance");
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][BN * ROWB];
    // general instance: positions of the block's rows, of the staged columns, and the row bounds
    __shared__ double2 ptsRow[GENERAL ? SIM_BLOCK_ROWS : 1];
    __shared__ double2 ptsCol[GENERAL ? 2 : 1][GENERAL ? BN : 1];
    __shared__ unsigned rowBound[GENERAL ? SIM_BLOCK_ROWS : 1];
    __shared__ float colBar[MODE == 2 ? 2 : 1][MODE == 2 ? BN : 1]; // minus the bars of the staged columns
    __shared__ unsigned colMaxL[MODE == 1 ? 2 : 1][MODE == 1 ? BN : 1]; // largest approximation of each staged column over the block's rows

    const int2 w = g.work[workItem];
    if (w.x < 0) return;
    if (g.onlyFlagged && g.pairOvf[w.x] != g.onlyFlagged) return; // later passes: pairs an earlier pass gave up on
    PairRec pr = g.pairs[w.x];
    if (g.gateMode && (g.gateMode == 1) != (pr.gateOff != 0)) return; // the other form of the filtered passes takes this pair
    // Small launches of the strict passes cut the columns into parts, one block per (row block, part): a pair
    // of 8192 x 8192 is 32 row blocks, far too few for 256 CUs.  The filtered passes need all columns of a row
    // in one wave, in order: they take the items of part 0 and walk every column.
    const int colParts = GENERAL ? 1 : max(g.colSplit, 1); // (the two-phase sweeps take column parts like the strict passes)
    const int colPart = w.y >> 24, rowBlock = w.y & 0xFFFFFF;
    if (GENERAL && colPart != 0) return;
    // slices of the pair's candidate list: one per wave (and column part) of a strict pass, or one per wave
    // of both orientations of the general pass
    const int nQ0 = ((pr.nA + SIM_BLOCK_ROWS - 1) / SIM_BLOCK_ROWS) * SIM_WAVES;
    const int nQ1 = ((pr.nB + SIM_BLOCK_ROWS - 1) / SIM_BLOCK_ROWS) * SIM_WAVES;
    if (TWOSIDED && g.orient) { // image B's rows stay in registers, image A streams
        const int n = pr.nA; pr.nA = pr.nB; pr.nB = n;
        int64_t t = pr.rowA32; pr.rowA32 = pr.rowB32; pr.rowB32 = t;
        t = pr.rowA16; pr.rowA16 = pr.rowB16; pr.rowB16 = t;
        t = pr.bestRow; pr.bestRow = pr.bestCol; pr.bestCol = t;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowBase = rowBlock * SIM_BLOCK_ROWS + wave * SIM_WAVE_ROWS;

    const unsigned char *table = F6 ? g.desc6 : (I8 ? reinterpret_cast<const unsigned char *>(g.desc8) : reinterpret_cast<const unsigned char *>(g.desc16));
    const unsigned char *Arows = table + (size_t)(pr.rowA16 + rowBase) * ROWB;
    const unsigned char *Brows = table + (size_t)pr.rowB16 * ROWB;

    // A fragments: row tile t (16 rows), k-block s; fp16 has four k-blocks of 32, int8 two of 64; an e2m3 lane's 32
    // elements of the tile's one k-block are the first 24 bytes of two such 16-byte pieces.
    constexpr int NKB = WIDE ? 4 : 2;
    typedef typename std::conditional<I8 || F6, int4v, half8>::type Frag; // 16 bytes either way
    Frag a[4][NKB];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < NKB; s++)
            a[t][s] = *reinterpret_cast<const Frag *>(Arows + (size_t)(t * 16 + (lane & 15)) * ROWB + (s * 4 + (lane >> 4)) * 16);

    // this block's steps of BN columns: [chunk0, nChunks)
    const int allChunks = (pr.nB + BN - 1) / BN;
    const int perPart = (allChunks + colParts - 1) / colParts;
    const int chunk0 = colPart * perPart;
    const int nChunks = min(allChunks, chunk0 + perPart);
    // the bound is the same for the whole block: keep it (and the branch on it) scalar
    const float thr_m = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pr.thr_m)));
    const int thr_i = __builtin_amdgcn_readfirstlane(pr.thr_i);
    const float thr6 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pr.thr6)));

    // LDS-DMA: instruction u of this wave fills 1 KiB = four 256-B lines of the step; lane l lands
    // on line 4 (wave DMA + u) + (l >> 4), physical position l & 15, which holds the logical
    // position (l & 15) ^ (line & 15).
    unsigned srcOff[DMA];
#pragma unroll
    for (int u = 0; u < DMA; u++) {
        const int line = 4 * (wave * DMA + u) + (lane >> 4);
        const int logical = (lane & 15) ^ (line & 15);
        srcOff[u] = !WIDE ? (2 * line + (logical >> 3)) * 128 + ((logical & 7) << 4) : line * 256 + (logical << 4);
    }
    auto dma_issue = [&](int c, int buf) __attribute__((always_inline)) {
        const unsigned char *src = Brows + (size_t)c * BN * ROWB;
#pragma unroll
        for (int u = 0; u < DMA; u++)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + srcOff[u]), (lptr_t)&lds[buf][(wave * DMA + u) * 1024], 16, 0, 0);
        if constexpr (GENERAL) { // the step's 64 pixel positions: one instruction (the table has no padding)
            if (wave == 0) {
                const int j = min(c * BN + lane, pr.nB - 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(g.pts + pr.rowB32 + j), (lptr_t)&ptsCol[buf][0], 16, 0, 0);
            }
        }
        if constexpr (MODE == 2) { // the step's 64 column bars (minus the bar, as the first sweeps left them): one instruction
            if (wave == 0) {
                const int j = min(c * BN + lane, pr.nB - 1);
                __builtin_amdgcn_global_load_lds((gptr_t)(g.rowMax + pr.bestCol + j), (lptr_t)&colBar[buf][0], 4, 0, 0);
            }
        }
    };
    if constexpr (GENERAL) {
        for (int rr = tid; rr < SIM_BLOCK_ROWS; rr += SIM_WAVES * 64) {
            ptsRow[rr] = g.pts[pr.rowA32 + min(rowBlock * SIM_BLOCK_ROWS + rr, pr.nA - 1)];
            rowBound[rr] = 0;
        }
    }
    using AccT = typename std::conditional<I8, SimAcc<int4v, int>, SimAcc<float4v, float>>::type;
    // e2m3: the A operands, six dwords each (the block scale of both operands is 2^0: exponent byte 127)
    int8v a6[F6 ? 4 : 1];
    if constexpr (F6) {
#pragma unroll
        for (int t = 0; t < 4; t++) a6[t] = int8v{a[t][0][0], a[t][0][1], a[t][0][2], a[t][0][3], a[t][1][0], a[t][1][1], 0, 0};
    }
    // Two-phase sweeps: per lane its 16 rows (row tile t, register q: row 16 t + 4 (lane >> 4) + q of the wave).
    // MODE 1 keeps their running maxima; MODE 2 starts every accumulator at minus the row's bar.
    float rowVal[MODE != 0 ? 4 : 1][4];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) rowVal[t][q] = -INFINITY;
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = rowBase + 16 * t + 4 * (lane >> 4) + q;
                // Minus the row's bar, as the first sweep left it (see there); a row beyond the image queues nothing.
                rowVal[t][q] = i < pr.nA ? g.rowMax[pr.bestRow + i] : -INFINITY;
            }
    }
    // one 32-column step: column tile ct, k-block s: one B fragment from LDS feeds four MFMAs
    auto mfma_step = [&](AccT &acc, int buf, int sub) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
            // A cell matters when it can be its row's best or its column's best: it starts at minus the lower of
            // the two bars.
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const float nb = colBar[buf][sub * 32 + c * 16 + (lane & 15)];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc.a[c][t][q] = fmaxf(rowVal[t][q], nb);
            }
        } else if constexpr (!F6) {
            acc.zero();
        }
        if constexpr (F6) {
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int row = sub * 32 + ct * 16 + (lane & 15);
                Frag b[2];
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const int chunk = s * 4 + (lane >> 4);
                    b[s] = *reinterpret_cast<const Frag *>(&lds[buf][(row >> 1) * 256 + ((((row & 1) * 8 + chunk) ^ ((row >> 1) & 15)) << 4)]);
                }
                const int8v b6 = {b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], 0, 0};
                const float4v zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc.a[ct][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6[t], b6, zero, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int s = 0; s < NKB; s++) {
                    const int row = sub * 32 + ct * 16 + (lane & 15);
                    const int chunk = s * 4 + (lane >> 4);
                    const int at = !WIDE ? (row >> 1) * 256 + ((((row & 1) * 8 + chunk) ^ ((row >> 1) & 15)) << 4)
                                      : row * 256 + ((chunk ^ (row & 15)) << 4);
                    const Frag b = *reinterpret_cast<const Frag *>(&lds[buf][at]);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if constexpr (I8)
                            acc.a[ct][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][s], b, acc.a[ct][t], 0, 0, 0);
                        else
                            acc.a[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][s], b, acc.a[ct][t], 0, 0, 0);
                    }
                }
        }
    };
    // this wave's slice of the pair's candidate list
    // (MODE 2: as many slices as the filtered passes would have, nQ0 + nQ1, shared out evenly among this sweep's
    // nQ0 * colParts waves, S >= 1 consecutive ones each: pair_queues() counts the same way)
    const int S = MODE == 2 ? max(1, (nQ0 + nQ1) / (nQ0 * colParts)) : 1;
    const int nQueues = MODE == 2 ? nQ0 * colParts * S : (GENERAL ? nQ0 + nQ1 : nQ0 * colParts);
    const int queueId = ((GENERAL && g.orient) ? nQ0 : colPart * nQ0) + rowBlock * SIM_WAVES + wave;
    float localLo[GENERAL ? AccT::NV : 1];
#pragma unroll
    for (int k = 0; k < (GENERAL ? AccT::NV : 1); k++) localLo[k] = -INFINITY;
    WaveQueue wq;
    wq.abl = ABL;
    // (The sweep against the bars of both images runs in one orientation only: its waves share out all slices of
    // the pair, S consecutive ones each, and fill them as one list.)
    const int sliceCap = g.candCap / nQueues;
    wq.cap = sliceCap * S;
    wq.base = g.cand + (size_t)pr.chunkLocal * g.candCap + (size_t)queueId * S * sliceCap;
    wq.n = 0;
    if (MODE == 2 && !(thr_m > -INFINITY)) wq.n = wq.cap + 1; // norms out of range: no bound holds, the pair is scored cell by cell
    auto test = [&](const AccT &acc, int colBase, int buf) __attribute__((always_inline)) {
        if constexpr ((ABL & 2) != 0)
            acc.keep_alive();
        else if constexpr (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) rowVal[t][q] = fmaxf(fmaxf(rowVal[t][q], acc.a[0][t][q]), acc.a[1][t][q]);
            // The columns' maxima, for the bars of the other image: over the lane's 16 rows here, over the block's
            // 256 in LDS, over the pair's rows in global memory (below).  From zero up: a cell at or below zero is
            // nobody's best (match.cpp:38,68), and floats from zero up order like their bit patterns.
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                float cm = 0.0f;
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int q = 0; q < 4; q += 2) cm = fmaxf(fmaxf(cm, acc.a[ct][t][q]), acc.a[ct][t][q + 1]);
                atomicMax(&colMaxL[buf][(colBase & (BN - 1)) + ct * 16 + (lane & 15)], __float_as_uint(cm));
            }
        } else {
            typename AccT::VT m[4] = {};
            if constexpr (!GENERAL) strict_maxima(acc, m);
            sim_epilogue<AccT, DOT, GENERAL, MODE == 2>(g, pr, acc, rowBase, colBase, lane,
                                             typename AccT::VT(I8 ? thr_i : (F6 ? thr6 : (MODE == 2 ? 0.0f : thr_m))), wq, localLo,
                                             GeneralLds{ptsRow + (GENERAL ? wave * SIM_WAVE_ROWS : 0),
                                                        &ptsCol[GENERAL ? buf : 0][GENERAL ? (colBase % BN) : 0],
                                                        rowBound + (GENERAL ? wave * SIM_WAVE_ROWS : 0)}, m);
        }
    };

    AccT acc;
    // Step c + 1 is in flight while step c is consumed; one barrier per step, after it.  The loop
    // is unrolled over the two LDS buffers so that every LDS address is a per-lane base plus an
    // immediate: no address arithmetic competes with the MFMAs for the vector issue port.
    auto step = [&](int c, const int buf) __attribute__((always_inline)) {
        if (!(ABL & 1) && c + 1 < nChunks) dma_issue(c + 1, buf ^ 1);
        if constexpr (GENERAL && !DOT) {
            // Pull the shared row bounds into the per-lane copies: a lane learns of a bound raised by
            // another lane's column only here or when one of its own cells gets past its stale copy.
            // Bounds move often during the first columns and rarely later.
            if (c < 8 || (c & 7) == 0) {
#pragma unroll
                for (int idx = 0; idx < AccT::NV; idx++) {
                    const unsigned key = rowBound[wave * SIM_WAVE_ROWS + AccT::row(idx, lane)];
                    localLo[idx] = fmaxf(localLo[idx], (key ? ordered_value(key) : -INFINITY) - SIM_MARGIN);
                }
            }
        }
        // a strict wave whose slice has overflowed only keeps the staging going: the pair is redone anyway
        if (GENERAL || wq.n <= wq.cap) {
#pragma unroll
            for (int sub = 0; sub < BN / 32; sub++) {
                mfma_step(acc, buf, sub);
                test(acc, c * BN + 32 * sub, buf);
            }
        }
        if (!(ABL & 1)) __syncthreads(); // next step has landed (vmcnt) and nobody still reads this one
        if constexpr (MODE == 1) {
            // One wave hands the step's column maxima on and clears them for the step after the next (which
            // writes them after the next barrier).
            if (wave == (c & 3)) {
                const unsigned v = colMaxL[buf][lane];
                colMaxL[buf][lane] = 0u;
                const int j = c * BN + lane;
                if (v != 0u && j < pr.nB) atomicMax(reinterpret_cast<unsigned *>(g.rowMax) + pr.bestCol + j, v);
            }
        }
    };
    if constexpr (MODE == 1) {
        if (tid < 2 * BN) colMaxL[tid / BN][tid % BN] = 0u;
    }
    if (chunk0 < nChunks) dma_issue(chunk0, 0);
    __syncthreads();
    for (int c = chunk0; c < nChunks; c += 2) {
        step(c, 0);
        if (c + 1 < nChunks) step(c + 1, (ABL & 1) ? 0 : 1);
    }
    if constexpr (MODE == 1) {
        // the 16 lanes that share lane >> 4 hold the same rows for 16 different columns of every tile
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float v = rowVal[t][q];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
                const int i = rowBase + 16 * t + 4 * (lane >> 4) + q;
                // (from zero up, like the columns': the blocks of the row's other column parts add theirs)
                if ((lane & 15) == 0 && i < pr.nA)
                    atomicMax(reinterpret_cast<unsigned *>(g.rowMax) + pr.bestRow + i, __float_as_uint(fmaxf(v, 0.0f)));
            }
        return;
    }
    if constexpr (MODE == 2) {
        // the wave's list, cut back into its S slices
        const int have = min(wq.n, wq.cap);
        for (int k = lane; k < S; k += 64)
            g.candCount[pr.chunkLocal * g.maxQueues + queueId * S + k] = max(0, min(sliceCap, have - k * sliceCap));
    }
    if (lane == 0) {
        if constexpr (MODE != 2) g.candCount[pr.chunkLocal * g.maxQueues + queueId] = min(wq.n, wq.cap);
        if (wq.n > wq.cap) {
            // (the filtered passes rank cells by cosine, which is not the L2 order when norms differ: an L2 pair
            // whose fp16 strict list overflows is scored cell by cell)
            const int flag = TWOSIDED ? 2 : (F6 ? 4 : (I8 ? 3 : (g.metric == SFM_METRIC_L2 ? 2 : 1)));
            g.pairOvf[pr.chunkLocal] = flag;
            atomicAdd(&g.flagCount[flag], 1); // "some pair carries this flag", for the passes that walk the list
        }
    }
}

// Between the two-phase sweeps: the largest approximations of a pair's rows and columns (bit patterns, from zero
// up) become their bars.  A row's (column's) best exact cosine is within one margin of its approximation, which is
// therefore at least the largest approximation minus two margins; 1e-5 more pays for the rounding of the second
// sweep's accumulators (they start at the bar instead of zero).  Stored negated, as the second sweep uses them.
// CLEAR: before the first sweep, the maxima of the pairs that will take it start at zero (only theirs: a chunk
// without such pairs, the usual case, touches nothing).
template <bool CLEAR> __global__ __launch_bounds__(256) void match_col_bars_kernel(const MatchArgs g)
{
    const int p = blockIdx.y;
    if (g.pairOvf[p] != 1) return;
    const PairRec pr = g.pairs[p];
    if (!pr.gateOff) return;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < pr.nA + pr.nB; k += gridDim.x * 256) {
        const int64_t at = k < pr.nA ? pr.bestRow + k : pr.bestCol + (k - pr.nA);
        if (CLEAR) {
            reinterpret_cast<unsigned *>(g.rowMax)[at] = 0u;
        } else {
            const float mx = __uint_as_float(reinterpret_cast<const unsigned *>(g.rowMax)[at]);
            g.rowMax[at] = -fmaxf(mx - 2.0f * SIM_MARGIN - 1e-5f, pr.thr_m);
        }
    }
}

// After the int8 strict pass: a pair whose int8 lists overflowed goes to the fp16 strict pass next, which is wasted
// on it if those lists overflow too (a loose threshold: the reference's default admits most cells).  This looks
// at 64 rows x 256 columns spread over the pair; when the fp16 bound admits cells at a rate that would fill a
// wave's slice four times over, the pair is handed straight to the filtered passes.  A guess either way costs time
// only: every path gives the same lists.  (Cosine only: an L2 pair that overflows is scored cell by cell, far
// too dear for a guess.)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void match_flood_probe_kernel(const MatchArgs g)
{
    __shared__ int total;
    const int p = blockIdx.x;
    if (g.pairOvf[p] != 3) return;
    const PairRec pr = g.pairs[p];
    if (pr.nA < 64 || pr.nB < 256 || !(pr.thr_m > -INFINITY)) return; // (small pairs: the attempt is cheap)
    const int tid = threadIdx.x, r = tid & 63, cg = tid >> 6;
    if (tid == 0) total = 0;
    __syncthreads();
    const int i = (int)(((int64_t)r * pr.nA) >> 6);
    const half2v *a = reinterpret_cast<const half2v *>(g.desc16 + (size_t)(pr.rowA16 + i) * KDIM);
    half2v ar[KDIM / 2];
#pragma unroll
    for (int q = 0; q < KDIM / 2; q++) ar[q] = a[q];
    int hits = 0;
    for (int k = 0; k < 16; k++) { // sixteen groups of 64 threads: 16 of the 256 columns each
        const int j = (int)(((int64_t)(cg * 16 + k) * pr.nB) >> 8);
        const half2v *b = reinterpret_cast<const half2v *>(g.desc16 + (size_t)(pr.rowB16 + j) * KDIM);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int q = 0; q < KDIM / 2; q += 2) {
            s0 = __builtin_amdgcn_fdot2(ar[q], b[q], s0, false);
            s1 = __builtin_amdgcn_fdot2(ar[q + 1], b[q + 1], s1, false);
        }
        hits += (s0 + s1 > pr.thr_m) ? 1 : 0;
    }
    atomicAdd(&total, hits);
    __syncthreads();
    if (tid == 0) {
        const int nQ0 = ((pr.nA + g.blockRows - 1) / g.blockRows) * g.wavesPerBlock, parts = max(g.colSplit, 1);
        const double sliceCap = (double)(g.candCap / (nQ0 * parts));
        const double expect = (double)total / (64.0 * 256.0) * 64.0 * ((double)pr.nB / parts); // of one wave's slice
        if (expect >= 4.0 * sliceCap) {
            g.pairOvf[p] = 1;
            atomicAdd(&g.flagCount[1], 1);
        }
    }
}

// In place of an int8 strict pass that is known to flood (see the chunk loop): every pair with work gets the flag
// such a pass would have left.
__global__ __launch_bounds__(256) void match_flag_flooded_kernel(const MatchArgs g, int np, int flag)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const PairRec pr = g.pairs[p];
    if (pr.nA <= 0 || pr.nB <= 0) return;
    g.pairOvf[p] = flag;
    if (g.flagCount[flag] == 0) g.flagCount[flag] = 1; // (only zero / non-zero matters)
}

// One work item per block: the pass every pair takes.
template <int OCC, int FORM, bool DOT, bool GENERAL, int ABL>
__global__ __launch_bounds__(SIM_WAVES * 64, OCC) void match_sim_kernel(const MatchArgs g)
{
    sim_block<FORM, DOT, GENERAL, ABL>(g, blockIdx.x);
}

// The later passes of the cosine metric redo flagged pairs only, usually none: a grid that just fills the
// device walks the work list instead of launching a block per item to find that out.  Items b and b + 8
// stay on one XCD as long as the grid is a multiple of 8.
template <int OCC, int FORM, bool DOT, bool GENERAL, int ABL, int MODE = 0>
__global__ __launch_bounds__(SIM_WAVES * 64, OCC) void match_sim_walk_kernel(const MatchArgs g)
{
    if (g.onlyFlagged && g.flagCount[g.onlyFlagged] == 0) return; // nothing to redo
    for (int item = blockIdx.x; item < g.nWork; item += gridDim.x) {
        sim_block<FORM, DOT, GENERAL, ABL, MODE>(g, item);
        __syncthreads(); // LDS of the item is free before the next one stages into it
    }
}

// The instances that can be selected with sfm_ctx_set_option("sim_variant").
struct SimVariant {
    void (*strict)(const MatchArgs);   // cosine, every cell above the bound is queued
    void (*strict16)(const MatchArgs); // int8 bound only: the fp16 strict instance that redoes pairs the int8 bound flooded
    const char *name;
};
static const SimVariant kSimVariants[] = {
    {match_sim_kernel<4, false, false, false, 0>, nullptr, "fp16 bound"},                                       // 0
    {match_sim_kernel<5, true, false, false, 0>, match_sim_walk_kernel<3, false, false, false, 0>, "int8 bound"},    // 1: default
    {match_sim_kernel<4, false, false, false, 2>, nullptr, "fp16, no candidate test (timing only)"},            // 2
    {match_sim_kernel<4, false, false, false, 3>, nullptr, "fp16, neither staging nor test (timing only)"},     // 3
    {match_sim_kernel<5, true, false, false, 2>, nullptr, "int8, no candidate test (timing only)"},             // 4
    {match_sim_kernel<5, true, false, false, 3>, nullptr, "int8, neither staging nor test (timing only)"},      // 5
    {match_sim_kernel<5, true, false, false, 4>, nullptr, "int8, candidates not stored (timing only)"},         // 6
    {match_sim_kernel<5, true, false, false, 8>, nullptr, "int8, hits not queued (timing only)"},               // 7
    {match_sim_kernel<5, 3, false, false, 2>, nullptr, "e2m3, no candidate test (timing only)"},                // 8
    {match_sim_kernel<5, 3, false, false, 3>, nullptr, "e2m3, neither staging nor test (timing only)"},         // 9
    {match_sim_kernel<5, 3, false, false, 8>, nullptr, "e2m3, hits not queued (timing only)"},                  // 10
    {match_sim_kernel<5, 3, false, false, 1>, nullptr, "e2m3, no staging (timing only)"},                       // 11
};
// cosine with the row-best filter (loose thresholds), and the dot metric: fp16 bound, two blocks per CU
// Binary descriptors as 256 signs (+1 / -1) per row: the int32 sum is 256 - 2 * Hamming distance, exactly.
void sfm_launch_sim_signs(const MatchArgs &g, int blocks, hipStream_t st)
{
    hipLaunchKernelGGL((match_sim_kernel<4, 2, false, false, 0>), dim3(blocks), dim3(SIM_WAVES * 64), 0, st, g);
}

static void (*const kSimGeneral)(const MatchArgs) = match_sim_walk_kernel<2, false, false, true, 0>;
static void (*const kSimRowMax)(const MatchArgs) = match_sim_walk_kernel<3, false, false, false, 0, 1>;
static void (*const kSimBar)(const MatchArgs) = match_sim_walk_kernel<2, false, false, false, 0, 2>;
static void (*const kSimDot)(const MatchArgs) = match_sim_kernel<2, false, true, true, 0>;
static void (*const kSimSix)(const MatchArgs) = match_sim_kernel<SIX_OCC, 3, false, false, 0>;        // e2m3 bound, every pair
static void (*const kSimEight)(const MatchArgs) = match_sim_walk_kernel<4, 1, false, false, 0>; // int8 bound, the pairs that one flooded
static const int kNumSimVariants = (int)(sizeof(kSimVariants) / sizeof(kSimVariants[0]));

// ---------------------------------------------------------------------------------------
// exact pass over the candidate lists
// ---------------------------------------------------------------------------------------
// Slices of a pair's candidate list as the similarity kernels laid them out: the strict pass
// has one per wave over image A's rows, the general pass adds one per wave over image B's.
__device__ __forceinline__ int pair_queues(const MatchArgs &g, const PairRec &pr, int p)
{
    const int nQ0 = ((pr.nA + g.blockRows - 1) / g.blockRows) * g.wavesPerBlock;
    const int nQ1 = ((pr.nB + g.blockRows - 1) / g.blockRows) * g.wavesPerBlock;
    const int parts = max(g.colSplit, 1);
    if (g.pairOvf[p] == 1 && g.rowMax && pr.gateOff) // the sweep against the bars laid them out (sim_block, MODE 2)
        return nQ0 * parts * max(1, (nQ0 + nQ1) / (nQ0 * parts));
    return ((g.metric == SFM_METRIC_DOT && !g.dotStrict) || g.pairOvf[p] == 1) ? nQ0 + nQ1 : nQ0 * parts;
}

// One wave per candidate slice, four slices per block.  RATIO: the second sweep of the ratio mode;
// pairs that went through the filtered passes (their lists hold possible bests only) are left to
// ratio_rows_kernel.
template <bool RATIO> __global__ __launch_bounds__(256) void match_cand_kernel(const MatchArgs g)
{
    const int p = blockIdx.y;
    if (g.pairOvf[p] == 2) return; // match_exact_kernel scores the whole pair
    if (RATIO && g.pairOvf[p] == 1) return;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g.maxQueues) return;
    const PairRec pr = g.pairs[p];
    const int nQueues = pair_queues(g, pr, p);
    if (q >= nQueues) return;
    const int cap = g.candCap / nQueues;
    const int n = g.candCount[p * g.maxQueues + q];
    if (!RATIO && n > 0 && lane == 0) atomicAdd(&g.stats[0], (u64)n);
    const uint2 *list = g.cand + (size_t)p * g.candCap + (size_t)q * cap;
    for (int k = lane; k < n; k += 64) {
        const uint2 c = list[k];
        if (RATIO) ratio_fold(g, pr, (int)c.x, (int)c.y, exact_dot(g, pr, (int)c.x, (int)c.y));
        else fold_cell(g, pr, (int)c.x, (int)c.y);
    }
}

// The same exact pass for dim == 128 with coalesced loads: a wave takes 64 candidates, brings
// their descriptor rows in through LDS 32 floats at a time (eight lanes fetch one 128-byte piece
// of one row), and every lane then runs its own candidate's multiply-add chain out of LDS in
// index order.  Candidates of a pair are numbered across its slices with a prefix sum so that
// waves are full.
#define CAND_ROWB 272 // LDS bytes per candidate per step: 2 x 128 B of data + 16 B pad (bank spread)
template <bool U8, bool RATIO> __global__ __launch_bounds__(256) void match_cand128_kernel(const MatchArgs g)
{
    __shared__ int prefix[1025];
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][64 * CAND_ROWB];
    __shared__ uint2 ij[4][64];

    const int p = blockIdx.y;
    if (g.pairOvf[p] == 2) return; // match_exact_kernel scores the whole pair
    if (RATIO && g.pairOvf[p] == 1) return;
    const PairRec pr = g.pairs[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nQueues = pair_queues(g, pr, p);
    const int cap = g.candCap / nQueues;

    // exclusive prefix of the slice counts (nQueues <= 1024): thread t owns slices 4t .. 4t + 3, the
    // running sums go through one scan per wave and the four wave totals
    __shared__ int waveSum[4];
    {
        int v[4], own = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = 4 * tid + u;
            v[u] = (q < nQueues) ? g.candCount[p * g.maxQueues + q] : 0;
            own += v[u];
        }
        int incl = own;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) waveSum[wave] = incl;
        __syncthreads();
        int run = incl - own;
        for (int w = 0; w < wave; w++) run += waveSum[w];
        if (tid == 0) prefix[0] = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            run += v[u];
            prefix[4 * tid + u + 1] = run;
        }
        __syncthreads();
    }
    const int total = prefix[nQueues];
    if (!RATIO && blockIdx.x == 0 && tid == 0 && total > 0) atomicAdd(&g.stats[0], (u64)total);

    const float *A32 = g.desc32 + (size_t)pr.rowA32 * 128;
    const float *B32 = g.desc32 + (size_t)pr.rowB32 * 128;
    unsigned char *st = stage[wave];
    for (int tile = blockIdx.x * 4 + wave; tile * 64 < total; tile += gridDim.x * 4) {
        const int c = tile * 64 + lane;
        uint2 me = make_uint2(0, 0);
        if (c < total) {
            int lo = 0, hi = nQueues; // last q with prefix[q] <= c
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (prefix[mid] <= c) lo = mid; else hi = mid;
            }
            me = g.cand[(size_t)p * g.candCap + (size_t)lo * cap + (c - prefix[lo])];
        }
        ij[wave][lane] = me;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        float dot = 0.0f;
        if constexpr (U8) { // whole rows are 128 B: one round, bytes widened on the fly (exact)
            const unsigned char *A8 = g.desc8u + (size_t)pr.rowA32 * 128;
            const unsigned char *B8 = g.desc8u + (size_t)pr.rowB32 * 128;
            uint4 v[16];
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int rc = it * 8 + (lane >> 3);
                const uint2 who = ij[wave][rc >> 1];
                const unsigned char *src = (rc & 1) ? B8 + (size_t)who.y * 128 : A8 + (size_t)who.x * 128;
                v[it] = *reinterpret_cast<const uint4 *>(src + (lane & 7) * 16);
            }
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int rc = it * 8 + (lane >> 3);
                *reinterpret_cast<uint4 *>(st + (rc >> 1) * CAND_ROWB + (rc & 1) * 128 + (lane & 7) * 16) = v[it];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint4 *ma = reinterpret_cast<const uint4 *>(st + lane * CAND_ROWB);
            const uint4 *mb = reinterpret_cast<const uint4 *>(st + lane * CAND_ROWB + 128);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint4 av = ma[e], bv = mb[e];
                const unsigned aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        dot = __fadd_rn(dot, __fmul_rn((float)((aw[w] >> (8 * b)) & 0xFFu), (float)((bw[w] >> (8 * b)) & 0xFFu)));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else
        for (int kc = 0; kc < 4; kc++) {
            float4 v[16];
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int rc = it * 8 + (lane >> 3); // row piece: candidate rc >> 1, image rc & 1
                const uint2 who = ij[wave][rc >> 1];
                const float *src = (rc & 1) ? B32 + (size_t)who.y * 128 : A32 + (size_t)who.x * 128;
                v[it] = *reinterpret_cast<const float4 *>(src + kc * 32 + (lane & 7) * 4);
            }
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int rc = it * 8 + (lane >> 3);
                *reinterpret_cast<float4 *>(st + (rc >> 1) * CAND_ROWB + (rc & 1) * 128 + (lane & 7) * 16) = v[it];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float4 *ma = reinterpret_cast<const float4 *>(st + lane * CAND_ROWB);
            const float4 *mb = reinterpret_cast<const float4 *>(st + lane * CAND_ROWB + 128);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float4 av = ma[e], bv = mb[e];
                dot = __fadd_rn(dot, __fmul_rn(av.x, bv.x));
                dot = __fadd_rn(dot, __fmul_rn(av.y, bv.y));
                dot = __fadd_rn(dot, __fmul_rn(av.z, bv.z));
                dot = __fadd_rn(dot, __fmul_rn(av.w, bv.w));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (c < total) {
            if (RATIO) ratio_fold(g, pr, (int)me.x, (int)me.y, dot);
            else fold_dot(g, pr, (int)me.x, (int)me.y, dot);
        }
    }
}

// Ratio mode for the pairs whose lists do not hold every cell above the bound (filtered passes, or
// scored cell by cell): every row that has a best walks all columns.  Rare and slow like
// match_exact_kernel.  grid (blocks per pair, pairs of the chunk).
__global__ __launch_bounds__(256) void ratio_rows_kernel(const MatchArgs g)
{
    const int p = blockIdx.y;
    const int f = g.pairOvf[p];
    if (f != 1 && f != 2) return;
    const PairRec pr = g.pairs[p];
    for (int i = blockIdx.x; i < pr.nA; i += gridDim.x) {
        if (g.best[pr.bestRow + i] == 0) continue;
        for (int j = threadIdx.x; j < pr.nB; j += 256) ratio_fold(g, pr, i, j, exact_dot(g, pr, i, j));
    }
}

// Rows whose best failed the ratio test lose it (the mutual check then drops them).
__global__ __launch_bounds__(256) void ratio_apply_kernel(const MatchArgs g)
{
    const PairRec pr = g.pairs[blockIdx.x];
    for (int i = threadIdx.x; i < pr.nA; i += 256)
        if (g.kill[pr.bestRow + i]) g.best[pr.bestRow + i] = 0;
}

// Pairs whose candidate list overflowed (one block): ovfList[0] = count, then the pair indices.
__global__ __launch_bounds__(256) void match_overflow_list_kernel(const MatchArgs g, int nPairs, int sixRan)
{
    __shared__ int n;
    if (threadIdx.x == 0) n = 0;
    __syncthreads();
    int general = 0, redone = 0, flagged = 0;
    for (int p = threadIdx.x; p < nPairs; p += blockDim.x) {
        const int f = g.pairOvf[p];
        if (f == 2) g.ovfList[1 + atomicAdd(&n, 1)] = p;
        general += (f == 1 || f == 2);
        redone += (f == 3);
        flagged += (f != 0);
    }
    if (general) atomicAdd(&g.stats[2], (u64)general);
    if (redone) atomicAdd(&g.stats[3], (u64)redone);
    if (flagged && sixRan) atomicAdd(&g.stats[4], (u64)flagged); // (every flag is then the e2m3 pass's or a later one's on a pair it flagged)
    __syncthreads();
    if (threadIdx.x == 0) {
        g.ovfList[0] = n;
        if (n) atomicAdd(&g.stats[1], (u64)n);
        if (g.flagOut) *g.flagOut = nPairs > 0 ? g.pairOvf[0] : 0; // (no later pass changes a flag)
    }
}

// Every cell of every overflowed pair, scored exactly.  Slow (a dependent 128-step chain per
// cell) but only reached when more cells than the list holds can still be a row or column best,
// e.g. many identical descriptors.  Consecutive threads take consecutive columns of one row.
__global__ __launch_bounds__(256) void match_exact_kernel(const MatchArgs g)
{
    const int n = g.ovfList[0];
    for (int k = 0; k < n; k++) {
        const PairRec pr = g.pairs[g.ovfList[1 + k]];
        const long long cells = (long long)pr.nA * pr.nB;
        for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < cells;
             c += (long long)gridDim.x * blockDim.x)
            fold_cell(g, pr, (int)(c / pr.nB), (int)(c % pr.nB));
    }
}

#include "csr_kernels.h"

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
extern "C" void sfm_imageset_destroy(sfm_imageset *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->owns_desc32) s->desc32.release();
    s->desc16.release();
    s->desc8.release();
    s->desc6.release();
    s->desc8u.release();
    s->imageStats.release();
    s->norms.release();
    s->norms2.release();
    if (s->owns_pts) s->pts.release();
    s->imageRecs.release();
    sfm_ctx *owner = s->ctx;
    delete s;
    sfm_ctx_release(owner);
}

extern "C" int64_t sfm_imageset_device_bytes(const sfm_imageset *s)
{
    if (!s) return 0;
    return (int64_t)(s->desc32.cap + s->desc16.cap + s->desc8.cap + s->desc6.cap + s->desc8u.cap + s->norms.cap + s->pts.cap + s->imageRecs.cap + s->imageStats.cap);
}

// Two host tables that together form the set's table (image 0 then image 1): the single-pair entry
// point uploads its two operands without first copying them side by side.
struct TwoParts {
    const float *desc[2];
    const double *pts[2];
    const unsigned char *desc_u8[2] = {nullptr, nullptr}; // byte-valued tables instead of desc (both or neither)
    // image 0 is already on the device (descriptors in the form the host tables have, and points): copied
    // from there instead of uploaded
    const void *dev_desc0 = nullptr;
    const double *dev_pts0 = nullptr;
};

// Image B of the last single-pair matcher call as it was uploaded.  The reference matches frame i against
// frame i + 1 and then frame i + 1 against frame i + 2 (scene.cpp:60), so B of one call is A of the next; a
// caller that promises not to change a table between the two calls saves that upload.
struct PairKeep {
    DevBuf desc, pts;
    const void *host_desc = nullptr;
    const double *host_pts = nullptr;
    int n = 0, dim = 0;
    bool bytes = false;
};

// reuse: slot holding a set built earlier by this function from host tables, whose device buffers are
// kept (they only grow).  The slot is emptied when the set is taken; on failure the set is destroyed like
// a new one, on success the caller puts the result back.
static int imageset_build(sfm_ctx *ctx, int n_images, const int32_t *counts, const float *desc,
                          const double *pts, int dim, bool on_device, sfm_imageset **set_out,
                          const unsigned char *desc_bytes = nullptr, sfm_imageset **reuse = nullptr,
                          const TwoParts *parts = nullptr, bool no_wait = false)
{
    if (!ctx || !set_out) return SFM_ERR_INVALID_ARG;
    *set_out = nullptr;
    if (n_images < 0 || (n_images > 0 && !counts)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "imageset: bad image count");
    if (dim < 1) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "imageset: dim must be >= 1 (was %d)", dim);
    if (dim > KDIM)
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "imageset: descriptor length %d > %d has no kernel", dim, KDIM);
    if (n_images > 65535) SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "imageset: more than 65535 images");
    SFM_HIP(ctx, hipSetDevice(ctx->device));

    const bool fresh = !(reuse && *reuse);
    sfm_imageset *s = fresh ? new (std::nothrow) sfm_imageset() : *reuse;
    if (!s) SFM_FAIL(ctx, SFM_ERR_OOM, "imageset: host allocation failed");
    if (reuse) *reuse = nullptr;
    if (fresh) sfm_ctx_retain(ctx);
    s->ctx = ctx;
    s->n_images = n_images;
    s->dim = dim;
    s->images.resize(n_images);
    int maxN = 0;
    int64_t r32 = 0, r16 = 0;
    for (int i = 0; i < n_images; i++) {
        if (counts[i] < 0) {
            sfm_imageset_destroy(s);
            SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "imageset: counts[%d] < 0", i);
        }
        s->images[i] = ImageRec{counts[i], 0, r32, r16};
        r32 += counts[i];
        r16 += round_up(counts[i], ROW_PAD);
        if (counts[i] > maxN) maxN = counts[i];
    }
    s->rows32 = r32;
    // one extra padded block so that a B tile or an A block may read past the last image
    r16 += ROW_PAD;
    s->rows16 = r16;
    if (r32 > 0 && !parts && ((!desc && !desc_bytes) || !pts)) {
        sfm_imageset_destroy(s);
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "imageset: null descriptor or point table");
    }

#define SET_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            sfm_imageset_destroy(s);                                                       \
            int _code = (_e == hipErrorOutOfMemory) ? SFM_ERR_OOM : SFM_ERR_HIP;           \
            SFM_FAIL(ctx, _code, "%s failed: %s", #expr, hipGetErrorString(_e));           \
        }                                                                                  \
    } while (0)

    const size_t b32 = (size_t)r32 * dim * sizeof(float);
    const size_t bpts = (size_t)r32 * 2 * sizeof(double);
    const bool parts_u8 = parts && (parts->desc_u8[0] || parts->desc_u8[1]);
    if (on_device) {
        s->owns_desc32 = false;
        s->owns_pts = false;
        s->desc32.p = const_cast<float *>(desc);
        s->pts.p = const_cast<double *>(pts);
    } else {
        SET_HIP(s->desc32.reserve(b32 ? b32 : 16));
        SET_HIP(s->pts.reserve(bpts ? bpts : 16));
        if (b32 && desc_bytes) { // byte table: a quarter of the upload, widened on the device
            DevBuf tmp;
            SET_HIP(tmp.reserve((size_t)r32 * dim));
            SET_HIP(hipMemcpyAsync(tmp.p, desc_bytes, (size_t)r32 * dim, hipMemcpyHostToDevice, ctx->stream));
            const int64_t n = r32 * dim;
            hipLaunchKernelGGL(expand_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const unsigned char *)tmp.p, n, s->desc32.as<float>());
            SET_HIP(hipStreamSynchronize(ctx->stream));
            tmp.release();
        } else if (parts) {
            int64_t row = 0;
            if (parts_u8) SET_HIP(s->desc8u.reserve((size_t)(r32 ? r32 : 1) * dim));
            // The second image goes up on a stream of its own: copies queued on one stream start one after the
            // other with a pause in between, on two streams they share the link without one.  (The device is
            // idle here: every single-pair call ends with a wait.)
            const bool twoStreams = ctx->stream3 && ctx->match_overlap && n_images == 2 && counts[0] > 0 && counts[1] > 0;
            for (int k = 0; k < 2 && k < n_images; k++) {
                const hipStream_t upk = (k == 1 && twoStreams) ? ctx->stream3 : ctx->stream;
                const size_t nk = (size_t)counts[k];
                if (nk && k == 0 && parts->dev_desc0) {
                    void *dst = parts_u8 ? s->desc8u.p : s->desc32.p;
                    SET_HIP(hipMemcpyAsync(dst, parts->dev_desc0, nk * dim * (parts_u8 ? 1 : sizeof(float)), hipMemcpyDeviceToDevice,
                                           ctx->stream));
                    SET_HIP(hipMemcpyAsync(s->pts.p, parts->dev_pts0, nk * 2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                } else if (nk) {
                    if (parts_u8) // a quarter of the bytes; widened below
                        SET_HIP(hipMemcpyAsync(s->desc8u.as<unsigned char>() + (size_t)row * dim, parts->desc_u8[k], nk * dim,
                                               hipMemcpyHostToDevice, upk));
                    else
                    SET_HIP(hipMemcpyAsync(s->desc32.as<float>() + (size_t)row * dim, parts->desc[k], nk * dim * sizeof(float),
                                           hipMemcpyHostToDevice, upk));
                    SET_HIP(hipMemcpyAsync(s->pts.as<double>() + (size_t)row * 2, parts->pts[k], nk * 2 * sizeof(double),
                                           hipMemcpyHostToDevice, upk));
                }
                row += counts[k];
            }
            if (twoStreams) {
                SET_HIP(hipEventRecord(ctx->ev[EV_JOIN], ctx->stream3));
                SET_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev[EV_JOIN], 0));
            }
            if (parts_u8 && r32 > 0) {
                const int64_t n = r32 * dim;
                hipLaunchKernelGGL(expand_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const unsigned char *)s->desc8u.p, n, s->desc32.as<float>());
            }
        } else if (b32)
            SET_HIP(hipMemcpyAsync(s->desc32.p, desc, b32, hipMemcpyHostToDevice, ctx->stream));
        if (bpts && !parts) SET_HIP(hipMemcpyAsync(s->pts.p, pts, bpts, hipMemcpyHostToDevice, ctx->stream));
    }
    SET_HIP(s->desc16.reserve((size_t)r16 * KDIM * sizeof(_Float16)));
    SET_HIP(s->desc8.reserve((size_t)r16 * 128));
    const size_t nImg = (size_t)(n_images ? n_images : 1);
    // The 6-bit table serves the first bound pass of batch calls; a set made for one single-pair call goes without.
    const bool six = ctx->match_six_bit && !no_wait;
    if (six) SET_HIP(s->desc6.reserve((size_t)r16 * 128));
    SET_HIP(s->imageStats.reserve(nImg * 10 * sizeof(int))); // [n][2] int8 statistics, [n][2] norm range, [n] flags, [n][4] extent of the points, [n] e2m3 error
    SET_HIP(s->norms2.reserve((size_t)(r32 ? r32 : 1) * sizeof(float)));
    SET_HIP(s->norms.reserve((size_t)(r32 ? r32 : 1) * sizeof(float)));
    SET_HIP(s->imageRecs.reserve(nImg * sizeof(ImageRec)));
    // the padding rows of both tables must read as zero, and the per-image words start at zero
    SET_HIP(clear_ranges(ctx->stream, {{s->desc8.p, (size_t)r16 * 128},
                                       {s->desc16.p, (size_t)r16 * KDIM * sizeof(_Float16)},
                                       {s->imageStats.p, nImg * 10 * sizeof(int)},
                                       {six ? s->desc6.p : nullptr, six ? (size_t)r16 * 128 : 0}}));
    int *const imageFlags = s->imageStats.as<int>() + 4 * nImg;
    s->flags.assign(n_images, 0);
    if (n_images > 0 && maxN > 0) {
        SET_HIP(hipMemcpyAsync(s->imageRecs.p, s->images.data(), (size_t)n_images * sizeof(ImageRec),
                               hipMemcpyHostToDevice, ctx->stream));
        const int rowsPerBlock = 64 * INGEST_WAVES / INGEST_LANES;
        dim3 g1((maxN + rowsPerBlock - 1) / rowsPerBlock, n_images);
        hipLaunchKernelGGL(((dim & 3) == 0 ? ingest_rows_kernel<true> : ingest_rows_kernel<false>), g1, dim3(64 * INGEST_WAVES), 0, ctx->stream,
                           s->desc32.as<float>(), s->imageRecs.as<ImageRec>(), dim, s->norms.as<float>(), s->norms2.as<float>(),
                           imageFlags, s->imageStats.as<int>(), s->imageStats.as<int>() + 2 * nImg, s->desc16.as<_Float16>(),
                           s->pts.as<double2>(), s->imageStats.as<unsigned>() + 5 * nImg);
        // int8 table: per-image scale from the largest unit-normalised element (statistics stay on the device)
        dim3 g3(((int64_t)maxN * 8 + 255) / 256, n_images);
        hipLaunchKernelGGL(ingest_i8_kernel, g3, dim3(256), 0, ctx->stream, s->desc32.as<float>(),
                           s->imageRecs.as<ImageRec>(), s->norms.as<float>(), dim, s->imageStats.as<int>(),
                           s->desc8.as<signed char>());
        if (six) {
            dim3 g6(((int64_t)maxN * 4 + 255) / 256, n_images);
            hipLaunchKernelGGL(ingest_f6_kernel, g6, dim3(256), 0, ctx->stream, s->desc32.as<float>(),
                               s->imageRecs.as<ImageRec>(), s->norms.as<float>(), dim, s->imageStats.as<int>(),
                               s->desc6.as<unsigned char>(), s->imageStats.as<int>() + 9 * nImg);
        }
        SET_HIP(hipGetLastError());
        if (!no_wait) { // the flags decide below whether the exact pass gets a byte copy
            SET_HIP(hipMemcpyAsync(s->flags.data(), imageFlags, (size_t)n_images * sizeof(int),
                                   hipMemcpyDeviceToHost, ctx->stream));
            SET_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    s->flags_on_host = !no_wait;
    // byte copy for the exact pass when the whole table is byte-valued
    // (tables that came as bytes are their own byte copy: with 128 elements per row the upload has its layout)
    const bool bytes_in_place = parts_u8 && dim == 128 && r32 > 0;
    s->u8_exact = dim == 128 && r32 > 0 && !no_wait;
    for (int i = 0; i < n_images; i++) s->u8_exact = s->u8_exact && !(s->flags[i] & IMG_NOT_U8);
    if (bytes_in_place) {
        s->u8_exact = true;
    } else if (s->u8_exact) {
        SET_HIP(s->desc8u.reserve((size_t)r32 * 128));
        hipLaunchKernelGGL(ingest_u8_kernel, dim3((unsigned)((r32 * 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                           s->desc32.as<float>(), r32, s->desc8u.as<unsigned char>());
        SET_HIP(hipGetLastError());
    }
    // The caller of a build that does not wait goes on queueing work on the same stream and waits once, at its end.
    if (!no_wait) SET_HIP(hipStreamSynchronize(ctx->stream));
#undef SET_HIP
    *set_out = s;
    return SFM_OK;
}

extern "C" int sfm_imageset_create(sfm_ctx *ctx, int n_images, const int32_t *counts, const float *desc,
                                   const double *pts, int dim, sfm_imageset **set_out)
{
    return imageset_build(ctx, n_images, counts, desc, pts, dim, false, set_out);
}

extern "C" int sfm_imageset_create_u8(sfm_ctx *ctx, int n_images, const int32_t *counts, const uint8_t *desc,
                                      const double *pts, int dim, sfm_imageset **set_out)
{
    return imageset_build(ctx, n_images, counts, nullptr, pts, dim, false, set_out, desc);
}

extern "C" int sfm_imageset_create_device(sfm_ctx *ctx, int n_images, const int32_t *counts,
                                          const float *desc_device, const double *pts_device, int dim,
                                          sfm_imageset **set_out)
{
    return imageset_build(ctx, n_images, counts, desc_device, pts_device, dim, true, set_out);
}

extern "C" void sfm_match_result_destroy(sfm_match_result *r)
{
    if (!r) return;
    (void)hipSetDevice(r->ctx->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    (void)hipStreamSynchronize(r->ctx->stream2); // a call that failed half-way may have left work of this result there
    (void)hipStreamSynchronize(r->ctx->stream3);
    if (r->done) (void)hipEventDestroy(r->done);
    r->readFence.destroy();
    if (r->upDone) (void)hipEventDestroy(r->upDone);
    if (r->statsEv) (void)hipEventDestroy(r->statsEv);
    r->hostStats.release();
    r->hostTotals.release();
    for (hipEvent_t &e : r->listEv)
        if (e) (void)hipEventDestroy(e);
    for (void *q : r->retired) (void)hipFree(q);
    DevBuf *bufs[] = {&r->offsets, &r->pairs, &r->pairRecs, &r->work, &r->cand, &r->candCount, &r->pairOvf,
                      &r->best, &r->ovfList, &r->counts, &r->stats, &r->m1, &r->kill, &r->candProbe, &r->rowMax};
    for (DevBuf *b : bufs) b->release();
    r->hostRecs.release();
    r->hostWork.release();
    r->hostOut.release();
    sfm_ctx *owner = r->ctx;
    delete r;
    sfm_ctx_release(owner);
}

int sfm_num_sim_variants() { return kNumSimVariants; }

static int match_batch_impl(sfm_ctx *ctx, const sfm_imageset *set, const int32_t *pair_list, int n_pairs,
                            int metric, float threshold, float maxDistSq, sfm_match_result **result_inout,
                            bool want_m1m2, float ratio = 0.0f, bool result_idle = false)
{
    if (!ctx || !set || !result_inout) return SFM_ERR_INVALID_ARG;
    if (ratio != 0.0f && !(ratio > 0.0f && ratio <= 1.0f))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: the ratio of the second-best test must lie in (0, 1], was %g", (double)ratio);
    if (ratio != 0.0f && metric == SFM_METRIC_DOT)
        SFM_FAIL(ctx, SFM_ERR_UNSUPPORTED, "match: the second-best ratio test is not defined for the dot metric");
    if (set->ctx != ctx) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: image set belongs to another context");
    if (n_pairs < 0 || (n_pairs > 0 && !pair_list)) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: bad pair list");
    if (metric != SFM_METRIC_COSINE && metric != SFM_METRIC_DOT && metric != SFM_METRIC_L2)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: unknown similarity metric %d", metric);
    SFM_HIP(ctx, hipSetDevice(ctx->device));

    sfm_match_result *res = *result_inout;
    if (!res) {
        res = new (std::nothrow) sfm_match_result();
        if (!res) SFM_FAIL(ctx, SFM_ERR_OOM, "match: host allocation failed");
        sfm_ctx_retain(ctx);
        res->ctx = ctx;
        *result_inout = res;
    } else if (res->ctx != ctx) {
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: result belongs to another context");
    } else if (!result_idle) {
        // The pinned staging buffers of a reused result may still be the source of copies in flight: wait for the
        // uploads of the call before (not for the call itself: its kernels may still be running while this call is
        // prepared and queued behind it; everything that touches the result's device buffers is ordered on the
        // context's streams).  A call that failed half-way left no such marker: wait for the streams.
        if (res->upDone && res->upValid) {
            SFM_HIP(ctx, hipEventSynchronize(res->upDone));
        } else {
            SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            SFM_HIP(ctx, hipStreamSynchronize(ctx->stream2));
            SFM_HIP(ctx, hipStreamSynchronize(ctx->stream3));
        }
    }
    res->upValid = false;
    for (void *q : res->retired) (void)hipFree(q); // list buffers the call before outgrew (hipFree waits for the device)
    res->retired.clear();
    // a gather of the call before may still read this result's lists on the communicator's stream
    SFM_HIP(ctx, res->readFence.wait_on(ctx->stream));
    res->n_pairs = n_pairs;
    res->total = -1;
    sfm_span_reset(ctx, 0);

    const SimVariant &variant = kSimVariants[ctx->sim_variant];
    const int BLOCK_ROWS = SIM_BLOCK_ROWS; // <= ROW_PAD
    const int chunkPairs = (int)std::min<int64_t>(ctx->pairs_per_launch, n_pairs > 0 ? n_pairs : 1);
    const int candCapMax = (int)ctx->cand_capacity;
    // Candidate slots per pair.  "cand_capacity" is what a pair may need at most (loose thresholds: hundreds of
    // thousands); at a selective threshold a pair fills a fraction of a per cent of that, and three sets of lists
    // for chunks of 2048 pairs at the maximum are 26 GB.  A call of several chunks therefore runs a short first
    // chunk (at most 128 pairs) with the maximum, looks at how far its slices filled, and sizes the lists of all
    // later chunks from that; the size is remembered in the result object for later calls with the same options.
    // Pairs that then overflow take the more selective passes as they always did (same lists, more time), and the
    // next call on the result object starts from four times the size.
    const MatchKey key = {set, metric, threshold, maxDistSq, ratio, candCapMax, ctx->sim_variant};
    bool probing = false;
    // how the last call fared whose counters have arrived (they are fetched at the end of every call, and nobody
    // waits for them)
    const bool sameKey = memcmp(&res->observedFor, &key, sizeof key) == 0;
    bool haveStats = false;
    u64 last[STATS_WORDS] = {};
    if (res->statsEv && res->statsPending && hipEventQuery(res->statsEv) == hipSuccess) {
        memcpy(last, res->hostStats.p, sizeof last);
        res->statsPending = false;
        haveStats = sameKey;
    }
    // A call whose e2m3 pass flooded half its pairs or more (descriptors whose unrelated cosines lie within that
    // bound's margin of the threshold): the next fifteen calls with the same options start with the int8 pass, the
    // sixteenth looks again.  (A call of several chunks learns the same from its first chunk.)
    if (!sameKey) res->sixSkip = 0;
    else if (haveStats && res->sixPairs > 0 && 2 * last[4] >= (u64)res->sixPairs) res->sixSkip = 15;
    else if (res->sixSkip > 0) res->sixSkip--;
    int64_t sixPairs = 0; // pairs of this call that took the e2m3 pass
    if (ctx->cand_observe && candCapMax > 0 && n_pairs > chunkPairs) {
        if (res->candObserved > 0 && sameKey) {
            if (haveStats && last[1] + last[2] + last[3] > 0)
                res->candObserved = (int)std::min<int64_t>(candCapMax, (int64_t)res->candObserved * 4);
        } else {
            res->candObserved = 0;
            res->floodObserved = false;
            res->sixFlooded = false;
            probing = true;
        }
    } else {
        res->candObserved = 0;
        res->floodObserved = false;
        res->sixFlooded = false;
    }
    memset(&res->observedFor, 0, sizeof res->observedFor);
    res->observedFor = key;
    std::vector<int> cstart; // first pair of every chunk, then n_pairs
    for (int c0 = 0; c0 < n_pairs;) {
        cstart.push_back(c0);
        c0 += (probing && c0 == 0) ? std::min(chunkPairs, 128) : chunkPairs;
    }
    cstart.push_back(n_pairs);
    const int nChunks = (int)cstart.size() - 1;

    // --- sizes ------------------------------------------------------------------------------
    // The records and work lists themselves are written chunk by chunk further down, each chunk
    // right before its launches, so that the device is busy while the host prepares the next one.
    SFM_HIP(ctx, res->hostRecs.reserve((size_t)(n_pairs ? n_pairs : 1) * sizeof(PairRec)));
    PairRec *recs = res->hostRecs.as<PairRec>();
    int64_t outTotalCap = 0, maxBest = 0;
    // work lists per chunk: [0] row blocks of image A (both passes), [1] row blocks of image B
    // (the general pass in its second orientation)
    std::vector<int64_t> chunkWorkBase[2];
    std::vector<int> chunkWorkCount[2];
    int64_t workTotal = 0;
    int maxNA = 0, maxNB = 0;
    const int wavesPerBlock = SIM_WAVES;
    auto row_blocks = [&](int nA, int nB, int orient) {
        if (nA <= 0 || nB <= 0) return 0;
        return ((orient ? nB : nA) + BLOCK_ROWS - 1) / BLOCK_ROWS;
    };
    int64_t mostBlocks = 0; // row blocks of image A in the chunk that has most
    std::vector<int64_t> chunkWorst((size_t)nChunks, 0); // most matches a chunk can have: min(nA, nB) per pair
    for (int cj = 0; cj < nChunks; cj++) {
        const int c0 = cstart[cj], c1 = cstart[cj + 1];
        int64_t bestOff = 0, blocks = 0;
        for (int p = c0; p < c1; p++) {
            const int ia = pair_list[2 * p], ib = pair_list[2 * p + 1];
            if (ia < 0 || ia >= set->n_images || ib < 0 || ib >= set->n_images)
                SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: pair %d names image %d/%d outside the set", p, ia, ib);
            const int nA = set->images[ia].n, nB = set->images[ib].n;
            bestOff += (int64_t)nA + nB;
            outTotalCap += std::min(nA, nB);
            chunkWorst[cj] += std::min(nA, nB);
            maxNA = std::max(maxNA, nA);
            maxNB = std::max(maxNB, nB);
            blocks += row_blocks(nA, nB, 0);
        }
        maxBest = std::max(maxBest, bestOff);
        mostBlocks = std::max(mostBlocks, blocks);
    }
    // Column parts of the strict passes (see sim_block): a launch that would leave most of the device idle, or
    // fill it unevenly with a few long blocks, is cut into more and shorter ones.  Every part keeps at least
    // eight steps of 64 columns, and the slices of a pair's candidate list stay within what the exact pass
    // numbers in one block (1024).
    // The dot metric (match.hpp:18-27) takes the strict passes first as well: a.b > T > 0 implies a cosine above
    // T / (maxA maxB) (pair_bounds_kernel), and the exact pass scores the dot product; pairs whose lists overflow go
    // on to the pass that tests the dot product cell by cell against the norms.  (Not with the ratio test, whose
    // bound is made for the cosine.)
    const bool dotStrict = metric == SFM_METRIC_DOT && ctx->match_dot_strict && ratio == 0.0f && candCapMax > 0;
    int colSplit = 1;
    if ((metric != SFM_METRIC_DOT || dotStrict) && mostBlocks > 0) {
        const int64_t want = 16 * (int64_t)ctx->n_cus; // three rounds of five blocks per CU: short blocks level the CUs out
        const int stepsB = (maxNB + SIM_BN - 1) / SIM_BN;
        const int queuesA = ((maxNA + BLOCK_ROWS - 1) / BLOCK_ROWS) * wavesPerBlock;
        while (colSplit < 16 && mostBlocks * colSplit < want && stepsB / (2 * colSplit) >= 8 &&
               (set->dim != 128 || queuesA * 2 * colSplit <= 1024))
            colSplit *= 2;
        if (ctx->match_col_split > 0) colSplit = ctx->match_col_split;
    }
    // Work lists: block b of a chunk's launch takes entry b.  Blocks b and b + 8 run on the same XCD (round-robin
    // dispatch), so the row blocks that stream the same columns (one pair, one column part) are dealt to one
    // residue class mod 8 and share that XCD's L2 copy of them.
    auto residue = [&](int pairInChunk, int part) { return (pairInChunk + part) & 7; };
    for (int cj = 0; cj < nChunks; cj++) {
        const int c0 = cstart[cj], c1 = cstart[cj + 1];
        int64_t queueLen[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
        for (int p = c0; p < c1; p++) {
            const int nA = set->images[pair_list[2 * p]].n, nB = set->images[pair_list[2 * p + 1]].n;
            for (int part = 0; part < colSplit; part++) queueLen[0][residue(p - c0, part)] += row_blocks(nA, nB, 0);
            queueLen[1][(p - c0) & 7] += row_blocks(nA, nB, 1);
        }
        for (int o = 0; o < 2; o++) {
            int64_t qmax = 0;
            for (int x = 0; x < 8; x++) qmax = std::max(qmax, queueLen[o][x]);
            chunkWorkBase[o].push_back(workTotal);
            chunkWorkCount[o].push_back((int)(qmax * 8));
            workTotal += qmax * 8;
        }
    }
    SFM_HIP(ctx, res->hostWork.reserve((size_t)(workTotal ? workTotal : 1) * sizeof(int2)));
    int2 *work = res->hostWork.as<int2>();

    // Records of one chunk, and its work lists.
    auto prepare_chunk = [&](int c0, int c1, size_t ci) {
        int64_t bestOff = 0;
        for (int p = c0; p < c1; p++) {
            const int ia = pair_list[2 * p], ib = pair_list[2 * p + 1];
            const ImageRec &A = set->images[ia], &B = set->images[ib];
            PairRec &r = recs[p];
            r.nA = A.n;
            r.nB = B.n;
            r.rowA32 = A.row32;
            r.rowB32 = B.row32;
            r.rowA16 = A.row16;
            r.rowB16 = B.row16;
            r.bestRow = bestOff;
            r.bestCol = bestOff + A.n;
            bestOff += (int64_t)A.n + B.n;
            r.chunkLocal = p - c0;
            r.ia = ia;
            r.ib = ib;
            r.thr_m = 0.0f; // pair_bounds_kernel fills the two bounds in on the device
            r.thr_i = 0;
            r.gateOff = 0;
        }
        for (int o = 0; o < 2; o++) {
            int2 *wl = work + chunkWorkBase[o][ci];
            const int cnt = chunkWorkCount[o][ci];
            for (int k = 0; k < cnt; k++) wl[k] = make_int2(-1, 0);
            int64_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int p = c0; p < c1; p++) {
                const int nb = row_blocks(recs[p].nA, recs[p].nB, o);
                for (int part = 0; part < (o == 0 ? colSplit : 1); part++) {
                    const int x = o == 0 ? residue(p - c0, part) : (p - c0) & 7;
                    for (int rb = 0; rb < nb; rb++) wl[(pos[x]++) * 8 + x] = make_int2(p - c0, rb | (part << 24));
                }
            }
        }
    };

    // --- device buffers -----------------------------------------------------------------
    SFM_HIP(ctx, res->pairRecs.reserve((size_t)(n_pairs ? n_pairs : 1) * sizeof(PairRec)));
    SFM_HIP(ctx, res->work.reserve((size_t)(workTotal ? workTotal : 1) * sizeof(int2)));
    SFM_HIP(ctx, res->offsets.reserve((size_t)(n_pairs + 1) * sizeof(int64_t)));
    // Match lists.  The worst case, min(nA, nB) matches per pair, is 64 KiB per pair of 8192 x 8192 (33 GB for half a
    // million pairs) and several times what selective thresholds leave.  A call of more than 4 LIST_RING chunks
    // therefore starts with room for the worst case of its first LIST_RING chunks (or what the result object
    // already holds), lets the device report the running total after every chunk (a copy into page-locked memory,
    // an event per chunk), and keeps the buffer at least as large as the last total it has seen plus the worst case
    // of every chunk enqueued since.  The host runs at most LIST_RING chunks ahead of the lists (the device stays
    // busy: a chunk takes ~10 ms, its enqueue well under 1), so the seen total is recent; when the buffer must grow
    // it grows once to 1.15 times the total extrapolated from the pairs seen, the lists so far move over on the
    // stream that writes them, and the old buffer is freed when the next call starts.  Shorter calls keep the
    // worst case: it is small there, and the host running far ahead is worth 1-2 % of the time (measured on the
    // benchmark's 10 chunks: 201 k against 197 k pairs/s).
    constexpr int LIST_RING = 8;
    const bool listObserve = ctx->match_list_observe != 0 && (nChunks > 4 * LIST_RING || ctx->match_list_observe > 1) &&
                             nChunks > LIST_RING && !want_m1m2;
    int64_t listCap = outTotalCap ? outTotalCap : 1; // entries
    if (listObserve) {
        int64_t first = 0;
        for (int cj = 0; cj < LIST_RING; cj++) first += chunkWorst[cj];
        listCap = std::min(listCap, std::max<int64_t>(first, 1));
        SFM_HIP(ctx, res->hostTotals.reserve((size_t)nChunks * sizeof(int64_t)));
        for (int cj = 0; cj < nChunks; cj++) res->hostTotals.as<int64_t>()[cj] = -1;
        for (hipEvent_t &e : res->listEv)
            if (!e) SFM_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    SFM_HIP(ctx, res->pairs.reserve((size_t)listCap * sizeof(int2)));
    listCap = (int64_t)(res->pairs.cap / sizeof(int2)); // (what the buffer holds: a reused result may hold more)
    int listKnown = -1;          // last chunk whose running total the host has seen
    int64_t listKnownTotal = 0;  // that total
    // Chunks in flight: while chunk c + 1 runs its bound passes on the context's stream, the exact pass, the
    // mutual check and the list output of chunk c run on a second stream, in the gaps the bound pass leaves
    // (its blocks fill the register files: kernels with few registers run beside it, the exact pass mostly
    // while one bound pass drains and the next has not started).  The per-chunk buffers exist up to three times
    // (parts of one allocation each), taken in turn; the second stream clears a set for its next use as
    // soon as the chunk's lists are written, so the first stream neither clears nor, as long as the second
    // keeps up, waits.  Events: ev[0..2] a chunk's bound passes are done, ev[3..5] its set is clear again
    // (by set), ev[6] fork / join of the two streams.
    const bool overlap = ctx->match_overlap && nChunks > 1 && !want_m1m2;
    const int sets = overlap ? std::min(MATCH_SETS, nChunks) : 1;
    const int maxQueues = std::max(1, std::max((maxNA + BLOCK_ROWS - 1) / BLOCK_ROWS + (maxNB + BLOCK_ROWS - 1) / BLOCK_ROWS,
                                               (maxNA + BLOCK_ROWS - 1) / BLOCK_ROWS * colSplit)) * wavesPerBlock;
    // lists: the first chunk of a call that observes has a region of its own at the maximum; the sets of the
    // other chunks are sized once that chunk's bound passes have run (see below)
    int candCap = res->candObserved > 0 ? std::min(res->candObserved, candCapMax) : candCapMax;
    size_t candSet = 0;
    auto reserve_lists = [&]() -> hipError_t {
        candSet = (size_t)round_up((int64_t)((size_t)chunkPairs * (candCap ? candCap : 1) * sizeof(uint2)), 256);
        return res->cand.reserve(candSet * sets);
    };
    if (probing)
        SFM_HIP(ctx, res->candProbe.reserve((size_t)(cstart[1] - cstart[0]) * candCapMax * sizeof(uint2)));
    else
        SFM_HIP(ctx, reserve_lists());
    const size_t candCountSet = (size_t)round_up((int64_t)((size_t)chunkPairs * maxQueues * sizeof(int)), 256);
    SFM_HIP(ctx, res->candCount.reserve(candCountSet * sets));
    const size_t pairOvfSet = (size_t)round_up((int64_t)((size_t)(chunkPairs + FLAG_VALUES) * sizeof(int)), 256); // flags, then how often each value was set
    SFM_HIP(ctx, res->pairOvf.reserve(pairOvfSet * sets));
    const size_t bestSet = (size_t)round_up((int64_t)((size_t)(maxBest ? maxBest : 1) * sizeof(u64)), 256);
    SFM_HIP(ctx, res->best.reserve(bestSet * sets));
    const size_t ovfListSet = (size_t)round_up((int64_t)((size_t)(chunkPairs + 1) * sizeof(int)), 256);
    SFM_HIP(ctx, res->ovfList.reserve(ovfListSet * sets));
    const size_t countsSet = (size_t)round_up((int64_t)((size_t)chunkPairs * sizeof(int)), 256);
    SFM_HIP(ctx, res->counts.reserve(countsSet * sets));
    SFM_HIP(ctx, res->stats.reserve(STATS_WORDS * sizeof(u64)));
    const size_t killSet = (size_t)round_up((int64_t)((size_t)(maxBest ? maxBest : 1) * sizeof(int)), 256);
    if (ratio > 0.0f) SFM_HIP(ctx, res->kill.reserve(killSet * sets));
    // two-phase filtered passes (cosine): one float per row of a chunk
    const bool twoPhase = ctx->match_two_phase && metric == SFM_METRIC_COSINE && candCapMax > 0;
    const size_t rowMaxSet = (size_t)round_up((int64_t)((size_t)(maxBest ? maxBest : 1) * sizeof(float)), 256);
    if (twoPhase) SFM_HIP(ctx, res->rowMax.reserve(rowMaxSet * sets));
    // The single-pair entry point fetches everything with one copy: its count, the per-row and per-column bests
    // and the list lie side by side, [count: 16 B][m1: nA][m2: nB][(i, j): min(nA, nB)].
    unsigned char *packed = nullptr;
    int32_t *packedM1 = nullptr, *packedM2 = nullptr;
    int2 *packedPairs = nullptr;
    if (want_m1m2) {
        if (n_pairs != 1) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: per-row bests are kept for single-pair calls only");
        const size_t bests = (size_t)round_up(((int64_t)maxNA + maxNB) * (int64_t)sizeof(int32_t), 16);
        SFM_HIP(ctx, res->m1.reserve(16 + bests + (size_t)std::min(maxNA, maxNB) * sizeof(int2)));
        packed = res->m1.as<unsigned char>();
        packedM1 = reinterpret_cast<int32_t *>(packed + 16);
        packedM2 = packedM1 + maxNA;
        packedPairs = reinterpret_cast<int2 *>(packed + 16 + bests);
        res->packedBytes = 16 + bests; // up to the list
    }
    int pairGuessFrom = 0; // single-pair calls: the flag of the call before with the same options (0: none, or time to look again)
    if (want_m1m2) {
        if (memcmp(&res->pairKey, &key, sizeof key) == 0 && (++res->pairCalls & 15u) != 0) pairGuessFrom = res->pairFlag;
        else res->pairCalls = memcmp(&res->pairKey, &key, sizeof key) == 0 ? res->pairCalls : 0;
        memset(&res->pairKey, 0, sizeof res->pairKey);
        res->pairKey = key;
        res->pairFlag = 0; // (until this call's flag has been read)
    }
    hipStream_t st = ctx->stream;
    if (n_pairs == 0) SFM_HIP(ctx, clear_ranges(st, {{res->offsets.p, sizeof(int64_t)}, {res->stats.p, STATS_WORDS * sizeof(u64)}}));
    if (overlap) { // the other two streams start where the first one stands
        SFM_HIP(ctx, hipEventRecord(ctx->ev[EV_JOIN], ctx->stream));
        SFM_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev[EV_JOIN], 0));
        SFM_HIP(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev[EV_JOIN], 0));
    }

    MatchArgs g;
    g.flagOut = packed ? reinterpret_cast<int *>(packed + 8) : nullptr;
    g.desc16 = set->desc16.as<_Float16>();
    g.desc8 = set->desc8.as<signed char>();
    g.desc6 = set->desc6.p ? set->desc6.as<unsigned char>() : nullptr;
    g.desc8u = set->u8_exact ? set->desc8u.as<unsigned char>() : nullptr;
    g.desc32 = set->desc32.as<float>();
    g.norms = set->norms.as<float>();
    g.norms2 = set->norms2.as<float>();
    g.pts = set->pts.as<double2>();
    g.maxQueues = maxQueues;
    g.colSplit = colSplit;
    g.dotStrict = dotStrict ? 1 : 0;
    g.blockRows = BLOCK_ROWS;
    g.wavesPerBlock = wavesPerBlock;
    g.stats = res->stats.as<u64>();
    g.dim = set->dim;
    g.metric = metric;
    g.threshold = threshold;
    g.maxDistSq = maxDistSq;
    g.ratioSq = (double)ratio * (double)ratio;

    // After the bound passes of the first chunk of a call that observes: how far did its slices fill?  (The one
    // wait of such a call.)  Sets candCap for the chunks that follow and makes their lists.
    auto observe = [&](hipStream_t stream, int np) -> hipError_t {
        const size_t nCount = (size_t)np * maxQueues;
        hipError_t e = res->hostOut.reserve((nCount + np) * sizeof(int));
        if (e != hipSuccess) return e;
        int *seen = res->hostOut.as<int>();
        if ((e = hipMemcpyAsync(seen, g.candCount, nCount * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(seen + nCount, g.pairOvf, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        int fullest = 0;
        bool gaveUp = false; // a pair went to the cell-by-cell pass: the maximum was not enough for it
        for (size_t k = 0; k < nCount; k++) fullest = std::max(fullest, seen[k]);
        int flooded = 0; // pairs whose int8 lists overflowed and that the fp16 strict pass did not settle either
        int flagged = 0; // pairs that any strict pass gave up on (the first of them is the e2m3 pass when there is one)
        for (int k = 0; k < np; k++) {
            gaveUp = gaveUp || seen[nCount + k] == 2;
            flooded += (seen[nCount + k] == 1 || seen[nCount + k] == 2) ? 1 : 0;
            flagged += seen[nCount + k] != 0 ? 1 : 0;
        }
        res->floodObserved = np >= 16 && 10 * flooded >= 9 * np;
        res->sixFlooded = 2 * flagged >= np;
        int64_t slice = 64;
        while (slice < 2 * (int64_t)fullest + 64) slice *= 2;
        candCap = gaveUp ? candCapMax : (int)std::min<int64_t>(candCapMax, slice * maxQueues);
        res->candObserved = candCap;
        return reserve_lists();
    };

    // A chunk's records and work lists go up, and its bounds are made, one chunk ahead on a stream of their own,
    // so that nothing but the launch itself lies between two bound passes on the context's stream.
    const hipStream_t up = overlap ? ctx->stream3 : ctx->stream;
    auto upload = [&](size_t ci) -> int {
        const int c0 = cstart[ci], c1 = cstart[ci + 1], np = c1 - c0;
        prepare_chunk(c0, c1, ci);
        SFM_HIP(ctx, hipMemcpyAsync(res->pairRecs.as<PairRec>() + c0, recs + c0, (size_t)np * sizeof(PairRec),
                                    hipMemcpyHostToDevice, up));
        const int64_t w0 = chunkWorkBase[0][ci], wn = (int64_t)chunkWorkCount[0][ci] + chunkWorkCount[1][ci];
        if (wn > 0)
            SFM_HIP(ctx, hipMemcpyAsync(res->work.as<int2>() + w0, work + w0, (size_t)wn * sizeof(int2), hipMemcpyHostToDevice, up));
        const int nImg = set->n_images ? set->n_images : 1;
        const BoundArgs b = {set->imageStats.as<int>(), set->imageStats.as<int>() + 4 * (size_t)nImg, set->imageStats.as<unsigned>() + 5 * (size_t)nImg,
                             nImg, set->dim, metric, threshold, ratio, maxDistSq, dotStrict ? 1 : 0,
                             set->desc6.p ? set->imageStats.as<int>() + 9 * (size_t)nImg : nullptr};
        hipLaunchKernelGGL(pair_bounds_kernel, dim3((np + 255) / 256), dim3(256), 0, up, res->pairRecs.as<PairRec>() + c0, np, b);
        if (overlap) SFM_HIP(ctx, hipEventRecord(ctx->ev[EV_UP + (int)(ci % (size_t)sets)], up));
        return SFM_OK;
    };
    if (nChunks > 0) {
        const int rc = upload(0);
        if (rc != SFM_OK) return rc;
    }

    for (size_t ci = 0; ci < (size_t)nChunks; ci++) {
        const int c0 = cstart[ci], c1 = cstart[ci + 1];
        const int np = c1 - c0;
        const bool probe = probing && ci == 0;
        // this chunk's set of buffers; st = the stream of its bound passes
        const int par = (int)(ci % (size_t)sets);
        st = ctx->stream;
        if (overlap) SFM_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[EV_UP + par], 0)); // its records are on the device
        char *const bestBuf = res->best.as<char>() + par * bestSet;
        char *const candCountBuf = res->candCount.as<char>() + par * candCountSet;
        char *const pairOvfBuf = res->pairOvf.as<char>() + par * pairOvfSet;
        char *const killBuf = ratio > 0.0f ? res->kill.as<char>() + par * killSet : nullptr;
        int *const countsBuf = reinterpret_cast<int *>(res->counts.as<char>() + par * countsSet);
        g.cand = probe ? res->candProbe.as<uint2>() : reinterpret_cast<uint2 *>(res->cand.as<char>() + par * candSet);
        g.candCap = probe ? candCapMax : candCap;
        g.candCount = reinterpret_cast<int *>(candCountBuf);
        g.pairOvf = reinterpret_cast<int *>(pairOvfBuf);
        g.flagCount = g.pairOvf + chunkPairs;
        g.best = reinterpret_cast<u64 *>(bestBuf);
        g.ovfList = reinterpret_cast<int *>(res->ovfList.as<char>() + par * ovfListSet);
        g.kill = reinterpret_cast<int *>(killBuf);
        g.rowMax = twoPhase ? reinterpret_cast<float *>(res->rowMax.as<char>() + par * rowMaxSet) : nullptr;
        g.gateMode = 0;
        g.pairs = res->pairRecs.as<PairRec>() + c0;
        const int2 *workList[2] = {res->work.as<int2>() + chunkWorkBase[0][ci], res->work.as<int2>() + chunkWorkBase[1][ci]};
        g.work = workList[0];
        g.orient = 0;
        int64_t bestLen = recs[c1 - 1].bestCol + recs[c1 - 1].nB;
        if (overlap && ci >= (size_t)sets) { // the chunk that had this set before has written its lists and cleared it
            SFM_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[EV_CLEAR + par], 0));
        } else { // (the first chunk also clears the first offset and the call's counters)
            SFM_HIP(ctx, clear_ranges(st, {{bestBuf, (size_t)bestLen * sizeof(u64)},
                                           {candCountBuf, (size_t)np * maxQueues * sizeof(int)},
                                           {pairOvfBuf, (size_t)(chunkPairs + FLAG_VALUES) * sizeof(int)},
                                           {res->offsets.p, ci == 0 ? sizeof(int64_t) : 0},
                                           {res->stats.p, ci == 0 ? STATS_WORDS * sizeof(u64) : 0}}));
        }
        const int nblk = chunkWorkCount[0][ci];
        // Grid of the passes that walk the work list: as many blocks as the device holds of that kernel at once (its
        // blocks per CU), so that all of them start together and share the items evenly -- with more, the late
        // blocks run their whole share on a nearly empty device.  A multiple of 8 keeps a pair's items on one XCD.
        auto walk_grid = [&](int blocksPerCu) { return std::max(8, blocksPerCu * ctx->n_cus / 8 * 8); };
        if (nblk > 0) {
            const bool cosine = metric != SFM_METRIC_DOT; // the L2 mode runs the cosine bound passes with its own per-pair bound
            const bool strictFirst = cosine || dotStrict;  // (the dot metric too, under a cosine bound from the images' largest norms)
            bool forked = false;
            // A loose threshold (the reference's default, match.hpp:59) floods the int8 lists of every pair, and the
            // attempt is wasted: when nine pairs in ten of the observed first chunk ended in the filtered passes, the
            // later chunks (and later calls with the same options) start where a flooded int8 pass would leave them
            // -- every pair flagged for the probe below, which still decides pair by pair between the fp16 strict
            // pass and the filtered ones.  A wrong guess costs time only: all paths give the same lists.
            // A single-pair call (the reference's pattern: frame after frame with the same options) remembers the flag
            // its pair ended with: after a call whose int8 lists flooded the next one starts where that one ended up --
            // at the probe (flag 3), or, when the fp16 lists flooded too, at the filtered passes (flag 1: no probe
            // either) -- except every sixteenth call, which takes the whole way again to see whether it still holds.
            const bool guessable = variant.strict16 && ctx->match_flood_probe && metric == SFM_METRIC_COSINE && ratio == 0.0f;
            const int pairGuess = (want_m1m2 && guessable && pairGuessFrom && twoPhase) ? (pairGuessFrom == 3 ? 3 : 1) : 0;
            const bool skipInt8 = pairGuess != 0 || (res->floodObserved && !probe && guessable);
            // The e2m3 bound comes first where there is a table for it: twice the int8 rate, a wider margin.  The
            // pairs it floods are redone under the int8 bound; when half the observed first chunk was, the later
            // chunks start with int8 as if there were no such table.  (Single-pair calls have none.)
            const bool sixFirst = strictFirst && !skipInt8 && g.desc6 && ctx->match_six_bit && ctx->sim_variant == 1 && res->sixSkip == 0 && (probe || !res->sixFlooded);
            if (strictFirst) {
                SFM_HIP(ctx, sfm_span_begin(ctx, 0, st));
                g.onlyFlagged = 0;
                if (skipInt8)
                    hipLaunchKernelGGL(match_flag_flooded_kernel, dim3((np + 255) / 256), dim3(256), 0, st, g, np, pairGuess ? pairGuess : 3);
                else if (sixFirst) hipLaunchKernelGGL(kSimSix, dim3(nblk), dim3(SIM_WAVES * 64), 0, st, g);
                else hipLaunchKernelGGL(variant.strict, dim3(nblk), dim3(SIM_WAVES * 64), 0, st, g);
                SFM_HIP(ctx, sfm_span_end(ctx, 0, st));
                // The first bound passes of consecutive chunks follow each other on this stream (each fills the
                // device on its own, and their timing spans stay clean); everything else of the chunk, the passes
                // that redo flagged pairs included, moves to the second stream.  (The chunk that observes how far
                // its lists fill stays on this stream until that is known.)
                if (overlap && !probe) {
                    SFM_HIP(ctx, hipEventRecord(ctx->ev[par], st));
                    st = ctx->stream2;
                    SFM_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[par], 0));
                    forked = true;
                }
            }
            if (ci + 1 < (size_t)nChunks) { // while the device runs this chunk, the host prepares the next
                const int rc = upload(ci + 1);
                if (rc != SFM_OK) return rc;
            }
            if (sixFirst) { // pairs flooded under the e2m3 margin: int8 bound
                g.onlyFlagged = 4;
                g.nWork = nblk;
                hipLaunchKernelGGL(kSimEight, dim3(std::min(nblk, walk_grid(4))), dim3(SIM_WAVES * 64), 0, st, g);
            }
            if (strictFirst && variant.strict16) { // pairs flooded under the wide int8 margin: fp16 bound
                if (ctx->match_flood_probe && metric == SFM_METRIC_COSINE && pairGuess != 1)
                    hipLaunchKernelGGL(match_flood_probe_kernel, dim3(np), dim3(1024), 0, st, g);
                g.onlyFlagged = 3;
                g.nWork = nblk;
                hipLaunchKernelGGL(variant.strict16, dim3(std::min(nblk, walk_grid(3))), dim3(SIM_WAVES * 64), 0, st, g);
            }
            // Cosine: pairs whose slices overflowed, again, with the row-best filter in both
            // orientations.  Dot: every pair takes this path.
            g.onlyFlagged = strictFirst ? 1 : 0;
            if (twoPhase) {
                // Pairs whose pixel gate excludes nothing: every row's largest approximation, then a strict sweep
                // against that bar, each with both images in the register-resident role.  The other flagged pairs
                // take the filtered pass below.
                g.gateMode = 1;
                g.orient = 0;
                g.work = workList[0];
                g.nWork = chunkWorkCount[0][ci];
                hipLaunchKernelGGL(match_col_bars_kernel<true>, dim3(8, np), dim3(256), 0, st, g);
                hipLaunchKernelGGL(kSimRowMax, dim3(std::min(g.nWork, walk_grid(3))), dim3(SIM_WAVES * 64), 0, st, g);
                hipLaunchKernelGGL(match_col_bars_kernel<false>, dim3(8, np), dim3(256), 0, st, g);
                hipLaunchKernelGGL(kSimBar, dim3(std::min(g.nWork, walk_grid(4))), // (two blocks per CU: two full rounds, which evens out the items' unequal cost)
                                   dim3(SIM_WAVES * 64), 0, st, g);
                g.gateMode = 2;
            }
            for (int o = 0; o < 2; o++) {
                g.orient = o;
                g.work = workList[o];
                if (!strictFirst && o == 0) SFM_HIP(ctx, sfm_span_begin(ctx, 0, st));
                g.nWork = chunkWorkCount[o][ci];
                hipLaunchKernelGGL(cosine ? kSimGeneral : kSimDot, dim3(cosine ? std::min(g.nWork, walk_grid(2)) : g.nWork),
                                   dim3(SIM_WAVES * 64), 0, st, g);
                if (!strictFirst && o == 0) SFM_HIP(ctx, sfm_span_end(ctx, 0, st));
            }
            g.gateMode = 0;
            if (probe) SFM_HIP(ctx, observe(st, np));
            if (overlap && !forked) {
                SFM_HIP(ctx, hipEventRecord(ctx->ev[par], st));
                st = ctx->stream2;
                SFM_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[par], 0));
            }
            g.onlyFlagged = 0;
            g.orient = 0;
            g.work = workList[0];
            const bool fast128 = set->dim == 128 && maxQueues <= 1024;
            // blocks per pair of the exact pass: every block first sums the pair's slice counts, so few of them when
            // the launch has many pairs (measured on 2048-pair chunks: 4.9 ms per step outside the similarity kernel
            // with one or two, 5.6 with four, 6.5 with eight)
            const int candBlocks = std::max(2, std::min(32, 2 * ctx->n_cus / np));
            if (candCap > 0) {
                if (fast128)
                    hipLaunchKernelGGL((g.desc8u ? match_cand128_kernel<true, false> : match_cand128_kernel<false, false>),
                                       dim3(candBlocks, np), dim3(256), 0, st, g);
                else
                    hipLaunchKernelGGL(match_cand_kernel<false>, dim3((maxQueues + 3) / 4, np), dim3(256), 0, st, g);
            }
            hipLaunchKernelGGL(match_overflow_list_kernel, dim3(1), dim3(256), 0, st, g, np, sixFirst ? 1 : 0);
            if (sixFirst) sixPairs += np;
            hipLaunchKernelGGL(match_exact_kernel, dim3(4 * ctx->n_cus), dim3(256), 0, st, g);
            if (ratio > 0.0f) { // second sweep: rows whose second best comes too close lose their best
                SFM_HIP(ctx, hipMemsetAsync(killBuf, 0, (size_t)bestLen * sizeof(int), st));
                if (candCap > 0) {
                    if (fast128)
                        hipLaunchKernelGGL((g.desc8u ? match_cand128_kernel<true, true> : match_cand128_kernel<false, true>),
                                           dim3(candBlocks, np), dim3(256), 0, st, g);
                    else
                        hipLaunchKernelGGL(match_cand_kernel<true>, dim3((maxQueues + 3) / 4, np), dim3(256), 0, st, g);
                }
                hipLaunchKernelGGL(ratio_rows_kernel, dim3(64, np), dim3(256), 0, st, g);
                hipLaunchKernelGGL(ratio_apply_kernel, dim3(np), dim3(256), 0, st, g);
            }
        } else { // a chunk without work: its memsets still precede what follows
            if (ci + 1 < (size_t)nChunks) {
                const int rc = upload(ci + 1);
                if (rc != SFM_OK) return rc;
            }
            if (probe) SFM_HIP(ctx, observe(st, np));
            if (overlap) {
                SFM_HIP(ctx, hipEventRecord(ctx->ev[par], st));
                st = ctx->stream2;
                SFM_HIP(ctx, hipStreamWaitEvent(st, ctx->ev[par], 0));
            }
        }
        if (listObserve) {
            int64_t *const totals = res->hostTotals.as<int64_t>();
            auto learn = [&]() -> hipError_t { // the total after the next chunk not yet seen (waits for that chunk's lists)
                const hipError_t e = hipEventSynchronize(res->listEv[(listKnown + 1) % LIST_RING]);
                if (e != hipSuccess) return e;
                listKnown++;
                listKnownTotal = totals[listKnown];
                if (listKnownTotal < 0) // (not what a completed event allows; the offsets themselves are complete by now)
                    return hipMemcpy(&listKnownTotal, res->offsets.as<int64_t>() + cstart[listKnown + 1], sizeof(int64_t), hipMemcpyDeviceToHost);
                return hipSuccess;
            };
            auto need = [&]() {
                int64_t s = listKnownTotal;
                for (int cj = listKnown + 1; cj <= (int)ci; cj++) s += chunkWorst[cj];
                return s;
            };
            while (listKnown < (int)ci - LIST_RING) SFM_HIP(ctx, learn()); // (this chunk takes the event of chunk ci - LIST_RING)
            while (need() > listCap && listKnown < (int)ci - 1) SFM_HIP(ctx, learn());
            if (need() > listCap) {
                int64_t est = outTotalCap;
                if (listKnown >= 0 && cstart[listKnown + 1] > 0) {
                    est = (int64_t)((double)listKnownTotal / cstart[listKnown + 1] * n_pairs * 1.15);
                    for (int cj = (int)ci; cj < std::min<int>(nChunks, (int)ci + LIST_RING); cj++) est += chunkWorst[cj];
                }
                const int64_t newCap = std::min<int64_t>(outTotalCap, std::max(need(), est));
                void *grown = nullptr;
                SFM_HIP(ctx, hipMalloc(&grown, (size_t)newCap * sizeof(int2) + 256));
                const int64_t written = need() - chunkWorst[ci]; // no list so far reaches beyond this
                if (written > 0)
                    SFM_HIP(ctx, hipMemcpyAsync(grown, res->pairs.p, (size_t)written * sizeof(int2), hipMemcpyDeviceToDevice, st));
                res->retired.push_back(res->pairs.p);
                res->pairs.p = grown;
                res->pairs.cap = (size_t)newCap * sizeof(int2) + 256;
                listCap = newCap;
            }
        }
        // (the scan continues the offsets of the chunk before: same stream, in order)
        if (np == 1) { // one pair: its list starts at the offset before it, no counting pass and no scan
            hipLaunchKernelGGL((match_mutual_kernel<true, 16>), dim3(1), dim3(1024), 0, st, g.pairs, g.best, (int *)nullptr,
                               res->offsets.as<int64_t>() + c0, packed ? packedPairs : res->pairs.as<int2>(),
                               packed ? packedM1 : nullptr, packed ? packedM2 : nullptr, res->offsets.as<int64_t>() + c0,
                               packed ? reinterpret_cast<int64_t *>(packed) : nullptr);
        } else {
            hipLaunchKernelGGL((match_mutual_kernel<false>), dim3(np), dim3(256), 0, st, g.pairs, g.best, countsBuf,
                               (const int64_t *)nullptr, (int2 *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                               (int64_t *)nullptr, (int64_t *)nullptr);
            hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, countsBuf, np, res->offsets.as<int64_t>(),
                               (int64_t)c0);
            hipLaunchKernelGGL((match_mutual_kernel<true>), dim3(np), dim3(256), 0, st, g.pairs, g.best,
                               (int *)nullptr, res->offsets.as<int64_t>() + c0, res->pairs.as<int2>(),
                               (int32_t *)nullptr, (int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr);
        }
        if (listObserve) {
            SFM_HIP(ctx, hipMemcpyAsync(res->hostTotals.as<int64_t>() + ci, res->offsets.as<int64_t>() + c1, sizeof(int64_t),
                                        hipMemcpyDeviceToHost, st));
            SFM_HIP(ctx, hipEventRecord(res->listEv[ci % LIST_RING], st));
        }
        if (overlap && ci + sets < (size_t)nChunks) { // clear the set for the chunk that takes it next (sizes: upper bounds)
            SFM_HIP(ctx, clear_ranges(st, {{bestBuf, (size_t)maxBest * sizeof(u64)},
                                           {candCountBuf, (size_t)chunkPairs * maxQueues * sizeof(int)},
                                           {pairOvfBuf, (size_t)(chunkPairs + FLAG_VALUES) * sizeof(int)}}));
            SFM_HIP(ctx, hipEventRecord(ctx->ev[EV_CLEAR + par], st));
        }
        SFM_HIP(ctx, hipGetLastError());
    }
    if (overlap) { // the caller's stream continues once both are done
        SFM_HIP(ctx, hipEventRecord(ctx->ev[EV_JOIN], ctx->stream2));
        SFM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev[EV_JOIN], 0));
    }
    // for the next call on this result object: when its staging buffers are free again, and how this call fared
    if (!res->upDone) SFM_HIP(ctx, hipEventCreateWithFlags(&res->upDone, hipEventDisableTiming));
    SFM_HIP(ctx, hipEventRecord(res->upDone, up));
    res->upValid = true;
    if (n_pairs > 0 && !want_m1m2) {
        if (!res->statsEv) SFM_HIP(ctx, hipEventCreateWithFlags(&res->statsEv, hipEventDisableTiming));
        SFM_HIP(ctx, res->hostStats.reserve(STATS_WORDS * sizeof(u64)));
        SFM_HIP(ctx, hipMemcpyAsync(res->hostStats.p, res->stats.p, STATS_WORDS * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
        res->sixPairs = sixPairs;
        SFM_HIP(ctx, hipEventRecord(res->statsEv, ctx->stream));
        res->statsPending = true;
    }
    return SFM_OK;
}

// The point on the context's stream where a batch call's lists are complete (the multi-GPU gather, which
// runs on a stream of its own, waits for it).
int sfm_match_result_mark_done(sfm_ctx *ctx, sfm_match_result *res)
{
    if (!res->done) SFM_HIP(ctx, hipEventCreateWithFlags(&res->done, hipEventDisableTiming));
    SFM_HIP(ctx, hipEventRecord(res->done, ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_match_batch(sfm_ctx *ctx, const sfm_imageset *set, const int32_t *pair_list, int n_pairs,
                               int metric, float threshold, float max_distance_px_squared,
                               sfm_match_result **result_inout)
{
    const int rc = match_batch_impl(ctx, set, pair_list, n_pairs, metric, threshold, max_distance_px_squared,
                                    result_inout, false);
    return rc == SFM_OK ? sfm_match_result_mark_done(ctx, *result_inout) : rc;
}

extern "C" int sfm_match_batch_ratio(sfm_ctx *ctx, const sfm_imageset *set, const int32_t *pair_list, int n_pairs,
                                     float threshold, float max_distance_px_squared, float ratio,
                                     sfm_match_result **result_inout)
{
    if (ctx && !(ratio > 0.0f && ratio <= 1.0f))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: the ratio of the second-best test must lie in (0, 1], was %g", (double)ratio);
    const int rc = match_batch_impl(ctx, set, pair_list, n_pairs, SFM_METRIC_COSINE, threshold, max_distance_px_squared,
                                    result_inout, false, ratio);
    return rc == SFM_OK ? sfm_match_result_mark_done(ctx, *result_inout) : rc;
}

extern "C" int sfm_match_batch_l2(sfm_ctx *ctx, const sfm_imageset *set, const int32_t *pair_list, int n_pairs,
                                  float max_distance_squared, float max_distance_px_squared, float ratio,
                                  sfm_match_result **result_inout)
{
    if (ctx && ratio != 0.0f && !(ratio > 0.0f && ratio <= 1.0f))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: the ratio of the second-best test must lie in (0, 1] (0 = no test), was %g", (double)ratio);
    const int rc = match_batch_impl(ctx, set, pair_list, n_pairs, SFM_METRIC_L2, max_distance_squared, max_distance_px_squared,
                                    result_inout, false, ratio);
    return rc == SFM_OK ? sfm_match_result_mark_done(ctx, *result_inout) : rc;
}

extern "C" int sfm_match_result_total(sfm_match_result *res, int64_t *total_out)
{
    if (!res || !total_out) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = res->ctx;
    if (res->total < 0) {
        int64_t t = 0;
        SFM_HIP(ctx, hipMemcpyAsync(&t, res->offsets.as<int64_t>() + res->n_pairs, sizeof t, hipMemcpyDeviceToHost,
                                    ctx->stream));
        SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        res->total = t;
    }
    *total_out = res->total;
    return SFM_OK;
}

extern "C" int sfm_match_result_offsets(sfm_match_result *res, int64_t *offsets_out)
{
    if (!res || !offsets_out) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = res->ctx;
    SFM_HIP(ctx, hipMemcpyAsync(offsets_out, res->offsets.p, (size_t)(res->n_pairs + 1) * sizeof(int64_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_match_result_pairs(sfm_match_result *res, int32_t *pairs_out)
{
    if (!res) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = res->ctx;
    int64_t total = 0;
    int rc = sfm_match_result_total(res, &total);
    if (rc != SFM_OK) return rc;
    if (total == 0) return SFM_OK;
    if (!pairs_out) return SFM_ERR_INVALID_ARG;
    SFM_HIP(ctx, hipMemcpyAsync(pairs_out, res->pairs.p, (size_t)total * sizeof(int2), hipMemcpyDeviceToHost,
                                ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SFM_OK;
}

extern "C" int sfm_match_result_device(sfm_match_result *res, const int64_t **offsets_device,
                                       const int32_t **pairs_device)
{
    if (!res) return SFM_ERR_INVALID_ARG;
    if (offsets_device) *offsets_device = res->offsets.as<int64_t>();
    if (pairs_device) *pairs_device = res->pairs.as<int32_t>();
    return SFM_OK;
}

extern "C" int sfm_match_result_stats(sfm_match_result *res, int64_t *candidates_out, int64_t *overflow_out)
{
    if (!res) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = res->ctx;
    u64 s[2] = {0, 0};
    SFM_HIP(ctx, hipMemcpyAsync(s, res->stats.p, sizeof s, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (candidates_out) *candidates_out = (int64_t)s[0];
    if (overflow_out) *overflow_out = (int64_t)s[1];
    return SFM_OK;
}

static int match_pair_impl(sfm_ctx *ctx, int algorithm, const void *descA, const double *ptsA, int nA, const void *descB,
                           const double *ptsB, int nB, bool bytes, int dim, int metric, float threshold,
                           float max_distance_px_squared, int32_t *pairs_out, int capacity, int *count_out,
                           int32_t *matches1_out, int32_t *matches2_out)
{
    if (!ctx || !count_out) return SFM_ERR_INVALID_ARG;
    *count_out = 0;
    if (algorithm != SFM_MATCH_TWO_SIDED)
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "Match: Unknown matching algorithm. Was %d", algorithm);
    if (nA < 0 || nB < 0 || capacity < 0) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: negative size");
    if ((nA > 0 && (!descA || !ptsA)) || (nB > 0 && (!descB || !ptsB)))
        SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: null input");

    // two-image set: A then B.  The set and the result live in the context from call to call, so a
    // sequence of calls (the reference matches frame after frame) allocates once.
    const int32_t counts[2] = {nA, nB};
    TwoParts parts = {{nullptr, nullptr}, {ptsA, ptsB}};
    if (bytes) {
        static const unsigned char none = 0; // (an empty image still marks the pair as byte-valued)
        parts.desc_u8[0] = descA ? static_cast<const unsigned char *>(descA) : &none;
        parts.desc_u8[1] = descB ? static_cast<const unsigned char *>(descB) : &none;
    } else {
        parts.desc[0] = static_cast<const float *>(descA);
        parts.desc[1] = static_cast<const float *>(descB);
    }
    PairKeep *keep = ctx->pair_keep;
    if (ctx->match_pair_reuse && keep && nA > 0 && keep->host_desc == descA && keep->host_pts == ptsA && keep->n == nA &&
        keep->dim == dim && keep->bytes == bytes) {
        parts.dev_desc0 = keep->desc.p;
        parts.dev_pts0 = keep->pts.as<double>();
    }
    if (keep) keep->host_desc = nullptr; // whatever happens below, the kept table is not offered twice
    sfm_imageset *set = nullptr;
    // Nothing waits for the device between the upload and the copy of the results: the set is built without
    // its host-side flags, and the result kept in the context is idle because every call ends with a wait.
    int rc = imageset_build(ctx, 2, counts, nullptr, nullptr, dim, false, &set, nullptr, &ctx->pair_set, &parts, true);
    if (rc != SFM_OK) return rc;
    ctx->pair_set = set;
    if (ctx->match_pair_reuse && nB > 0) {
        if (!keep) keep = ctx->pair_keep = new (std::nothrow) PairKeep();
        if (!keep) SFM_FAIL(ctx, SFM_ERR_OOM, "match: host allocation failed");
        const size_t bDesc = (size_t)nB * dim * (bytes ? 1 : sizeof(float)), bPts = (size_t)nB * 2 * sizeof(double);
        SFM_HIP(ctx, keep->desc.reserve(bDesc));
        SFM_HIP(ctx, keep->pts.reserve(bPts));
        const void *src = bytes ? (const void *)(set->desc8u.as<unsigned char>() + (size_t)nA * dim)
                                : (const void *)(set->desc32.as<float>() + (size_t)nA * dim);
        SFM_HIP(ctx, hipMemcpyAsync(keep->desc.p, src, bDesc, hipMemcpyDeviceToDevice, ctx->stream));
        SFM_HIP(ctx, hipMemcpyAsync(keep->pts.p, set->pts.as<double>() + (size_t)nA * 2, bPts, hipMemcpyDeviceToDevice, ctx->stream));
        keep->host_desc = descB;
        keep->host_pts = ptsB;
        keep->n = nB;
        keep->dim = dim;
        keep->bytes = bytes;
    }
    const int32_t pl[2] = {0, 1};
    rc = match_batch_impl(ctx, set, pl, 1, metric, threshold, max_distance_px_squared, &ctx->pair_result, true, 0.0f, true);
    if (rc != SFM_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    sfm_match_result *res = ctx->pair_result;
    // The count, the per-row / per-column bests and min(nA, nB, capacity) entries of the list (the count is not
    // known yet) come in one copy into a pinned buffer: one wait for the whole call.
    const size_t nList = (size_t)std::min(std::min(nA, nB), capacity);
    const size_t bytesOut = res->packedBytes + nList * sizeof(int2);
    SFM_HIP(ctx, res->hostOut.reserve(bytesOut));
    unsigned char *stage = res->hostOut.as<unsigned char>();
    hipStream_t st = ctx->stream;
    SFM_HIP(ctx, hipMemcpyAsync(stage, res->m1.p, bytesOut, hipMemcpyDeviceToHost, st));
    SFM_HIP(ctx, hipStreamSynchronize(st));
    const size_t b1 = matches1_out ? (size_t)nA * sizeof(int32_t) : 0, b2 = matches2_out ? (size_t)nB * sizeof(int32_t) : 0;
    int64_t total = 0;
    memcpy(&total, stage, sizeof total);
    memcpy(&res->pairFlag, stage + 8, sizeof(int)); // the flag the pair ended with: where the next call starts
    res->total = total;
    if (total > capacity) SFM_FAIL(ctx, SFM_ERR_CAPACITY, "match: output capacity too small");
    if (total > 0) {
        if (!pairs_out) SFM_FAIL(ctx, SFM_ERR_INVALID_ARG, "match: null output list");
        memcpy(pairs_out, stage + res->packedBytes, (size_t)total * sizeof(int2));
    }
    if (b1) memcpy(matches1_out, stage + 16, b1);
    if (b2) memcpy(matches2_out, stage + 16 + (size_t)nA * sizeof(int32_t), b2);
    *count_out = (int)total;
    return SFM_OK;
}

extern "C" int sfm_match_pair(sfm_ctx *ctx, int algorithm, const float *descA, const double *ptsA, int nA,
                              const float *descB, const double *ptsB, int nB, int dim, int metric,
                              float threshold, float max_distance_px_squared, int32_t *pairs_out, int capacity,
                              int *count_out, int32_t *matches1_out, int32_t *matches2_out)
{
    return match_pair_impl(ctx, algorithm, descA, ptsA, nA, descB, ptsB, nB, false, dim, metric, threshold, max_distance_px_squared,
                           pairs_out, capacity, count_out, matches1_out, matches2_out);
}

extern "C" int sfm_match_pair_u8(sfm_ctx *ctx, int algorithm, const uint8_t *descA, const double *ptsA, int nA,
                                 const uint8_t *descB, const double *ptsB, int nB, int dim, int metric,
                                 float threshold, float max_distance_px_squared, int32_t *pairs_out, int capacity,
                                 int *count_out, int32_t *matches1_out, int32_t *matches2_out)
{
    return match_pair_impl(ctx, algorithm, descA, ptsA, nA, descB, ptsB, nB, true, dim, metric, threshold, max_distance_px_squared,
                           pairs_out, capacity, count_out, matches1_out, matches2_out);
}

void sfm_pair_match_free(sfm_ctx *ctx)
{
    if (ctx->pair_result) sfm_match_result_destroy(ctx->pair_result);
    if (ctx->pair_set) sfm_imageset_destroy(ctx->pair_set);
    if (ctx->pair_keep) {
        ctx->pair_keep->desc.release();
        ctx->pair_keep->pts.release();
        delete ctx->pair_keep;
    }
    ctx->pair_result = nullptr;
    ctx->pair_set = nullptr;
    ctx->pair_keep = nullptr;
}

extern "C" int64_t sfm_match_result_device_bytes(const sfm_match_result *r)
{
    if (!r) return 0;
    const DevBuf *bufs[] = {&r->offsets, &r->pairs, &r->pairRecs, &r->work, &r->cand, &r->candCount, &r->pairOvf,
                            &r->best, &r->ovfList, &r->counts, &r->stats, &r->m1, &r->kill, &r->candProbe, &r->rowMax};
    int64_t total = 0;
    for (const DevBuf *b : bufs) total += (int64_t)b->cap;
    return total;
}

extern "C" int sfm_match_result_stats_ex(sfm_match_result *res, int64_t *stats4_out)
{
    if (!res || !stats4_out) return SFM_ERR_INVALID_ARG;
    sfm_ctx *ctx = res->ctx;
    u64 s[4] = {0, 0, 0, 0};
    SFM_HIP(ctx, hipMemcpyAsync(s, res->stats.p, sizeof s, hipMemcpyDeviceToHost, ctx->stream));
    SFM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; i++) stats4_out[i] = (int64_t)s[i];
    return SFM_OK;
}
