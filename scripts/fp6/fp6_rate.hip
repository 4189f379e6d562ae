// Rate of the bound loop's inner part on one kernel shape: B fragments from LDS, T row tiles per wave, the maximum
// test of the strict passes; e2m3 (one 16x16x128 MFMA per tile) against int8 (two 16x16x64).  LDS holds noise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, bool FP6, bool TEST, int PAD>
__global__ __launch_bounds__(256) void rate(const unsigned *src, int iters, int *out)
{
    // 64 columns x 128 B (int8, or e2m3 padded to 32 B per k-block) or x 96 B (e2m3 dense, PAD = 0)
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 128];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 128 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = src[i];
    __syncthreads();
    int8v a6[T];
    int4v a8[T][2];
    for (int t = 0; t < T; t++) {
        for (int w = 0; w < 8; w++) a6[t][w] = (int)src[(t * 64 + lane) * 8 + w];
        for (int s = 0; s < 2; s++) for (int w = 0; w < 4; w++) a8[t][s][w] = (int)src[(t * 64 + lane) * 8 + s * 4 + w];
    }
    int hits = 0;
    const float thr = 1e30f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            const int col = ct * 16 + (lane & 15), kb = lane >> 4;
            if constexpr (FP6) {
                float4v acc[T];
                int8v b = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (PAD) {
                    const int4v lo = *reinterpret_cast<const int4v *>(&lds[col * 128 + ((kb * 2) ^ (col & 7)) * 16]);
                    const int4v hi = *reinterpret_cast<const int4v *>(&lds[col * 128 + ((kb * 2 + 1) ^ (col & 7)) * 16]);
                    b[0] = lo[0], b[1] = lo[1], b[2] = lo[2], b[3] = lo[3], b[4] = hi[0], b[5] = hi[1];
                } else {
                    const int2 *p = reinterpret_cast<const int2 *>(&lds[col * 96 + kb * 24]);
                    const int2 x = p[0], y = p[1], z = p[2];
                    b[0] = x.x, b[1] = x.y, b[2] = y.x, b[3] = y.y, b[4] = z.x, b[5] = z.y;
                }
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const float4v z = {0, 0, 0, 0};
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6[t], b, z, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
                if constexpr (TEST) {
                    float m0 = -1e30f, m1 = -1e30f;
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        m0 = fmaxf(fmaxf(m0, acc[t][0]), acc[t][1]);
                        m1 = fmaxf(fmaxf(m1, acc[t][2]), acc[t][3]);
                    }
                    if (__builtin_amdgcn_ballot_w64(fmaxf(m0, m1) > thr) != 0) hits++;
                } else {
#pragma unroll
                    for (int t = 0; t < T; t++) asm volatile("" ::"v"(acc[t]));
                }
            } else {
                int4v acc[T];
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = int4v{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const int4v b = *reinterpret_cast<const int4v *>(&lds[col * 128 + ((s * 4 + kb) ^ (col & 7)) * 16]);
#pragma unroll
                    for (int t = 0; t < T; t++) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a8[t][s], b, acc[t], 0, 0, 0);
                }
                if constexpr (TEST) {
                    int m0 = -(1 << 30), m1 = -(1 << 30);
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        m0 = max(max(m0, acc[t][0]), acc[t][1]);
                        m1 = max(max(m1, acc[t][2]), acc[t][3]);
                    }
                    if (__builtin_amdgcn_ballot_w64(max(m0, m1) > (1 << 30)) != 0) hits++;
                } else {
#pragma unroll
                    for (int t = 0; t < T; t++) asm volatile("" ::"v"(acc[t]));
                }
            }
        }
    }
    if (hits == 12345) out[0] = hits;
}

template <int T, bool FP6, bool TEST, int PAD> static void run(const char *name, const unsigned *src, int *out, int blocksPerCu)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * blocksPerCu;
    hipLaunchKernelGGL((rate<T, FP6, TEST, PAD>), dim3(grid), dim3(256), 0, 0, src, 10, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<T, FP6, TEST, PAD>), dim3(grid), dim3(256), 0, 0, src, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tiles = (double)grid * 4 * iters * 4 * T; // 16x16x128 tiles
    printf("%-34s %d blocks/CU: %7.3f ms, %6.2f Ttile/s = %6.2f PFLOP/s\n", name, blocksPerCu, ms, tiles / (ms * 1e-3) / 1e12, tiles * 65536 / (ms * 1e-3) / 1e15);
}
int main()
{
    std::vector<unsigned> h(64 * 1024);
    srand(3);
    for (auto &x : h) x = (unsigned)rand() * 2654435761u;
    unsigned *src; int *out;
    CHECK(hipMalloc(&src, h.size() * 4)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int occ = 1; occ <= 2; occ++) {
        run<4, false, false, 1>("int8 T=4 no test", src, out, occ);
        run<4, false, true, 1>("int8 T=4 test", src, out, occ);
        run<4, true, false, 1>("e2m3 T=4 padded no test", src, out, occ);
        run<4, true, true, 1>("e2m3 T=4 padded test", src, out, occ);
        run<8, true, false, 1>("e2m3 T=8 padded no test", src, out, occ);
        run<8, true, true, 1>("e2m3 T=8 padded test", src, out, occ);
        run<8, true, true, 0>("e2m3 T=8 dense test", src, out, occ);
        run<4, true, true, 0>("e2m3 T=4 dense test", src, out, occ);
    }
    return 0;
}
