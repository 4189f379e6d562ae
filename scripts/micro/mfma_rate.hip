// Raw MFMA issue rates from registers on random data, every CU busy: int8 16x16x64 against block-scaled
// fp8 16x16x128 (unit scales), with and without 18 independent v_max3-class fillers per 16 (8) MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int MODE, int FILL> __global__ __launch_bounds__(256) void k(const int *src, int *dst, int iters)
{
    const int lane = threadIdx.x;
    int4v a4[4], b4[4];
    int8v a8[4], b8[2];
    for (int t = 0; t < 4; t++) {
        for (int q = 0; q < 4; q++) a4[t][q] = src[(lane * 37 + t * 4 + q) & 4095];
        for (int q = 0; q < 8; q++) a8[t][q] = src[(lane * 53 + t * 8 + q) & 4095] & 0x3F3F3F3F;
    }
    for (int f = 0; f < 4; f++)
        for (int q = 0; q < 4; q++) b4[f][q] = src[(lane * 11 + f * 131 + q) & 4095];
    for (int f = 0; f < 2; f++)
        for (int q = 0; q < 8; q++) b8[f][q] = src[(lane * 17 + f * 257 + q) & 4095] & 0x3F3F3F3F;
    int4v ci[8];
    float4v cf[8];
    int m = 0;
    float mf = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) ci[u] = int4v{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int s = 0; s < 2; s++) ci[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4[u & 3], b4[2 * (u >> 2) + s], ci[u], 0, 0, 0);
            if (FILL) {
#pragma unroll
                for (int u = 0; u < 8; u++) m = max(max(m, ci[u][0]), max(ci[u][1], max(ci[u][2], ci[u][3])));
            } else {
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("" ::"v"(ci[u]));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) cf[u] = float4v{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 8; u++)
                cf[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[u & 3], b8[u >> 2], cf[u], 0, 0, 0, 127, 0, 127);
            if (FILL) {
#pragma unroll
                for (int u = 0; u < 8; u++) mf = fmaxf(fmaxf(mf, cf[u][0]), fmaxf(cf[u][1], fmaxf(cf[u][2], cf[u][3])));
            } else {
#pragma unroll
                for (int u = 0; u < 8; u++) asm volatile("" ::"v"(cf[u]));
            }
        }
        b4[0][0] ^= it;
        b4[2][1] ^= it;
        b8[0][0] ^= (it & 0x0F0F0F0F);
        b8[1][3] ^= (it & 0x0F0F0F0F);
    }
    dst[blockIdx.x * 256 + lane] = m + (int)mf;
}

template <int MODE, int FILL> static void run(const char *name, const int *src, int *dst)
{
    const int iters = 20000, blocks = 256 * 5;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, src, dst, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, FILL>), dim3(blocks), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)blocks * 4 * iters * 16 * (16.0 * 16 * 64); // both modes: 16 x (16x16x64) per iteration
    printf("%-28s %8.3f ms  %7.1f Top/s\n", name, ms, 2 * macs / (ms * 1e-3) / 1e12);
}

int main()
{
    std::vector<int> h(4096);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    int *src, *dst;
    hipMalloc(&src, 4096 * 4); hipMalloc(&dst, 256 * 5 * 256 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 0>("int8 16x16x64", src, dst);
    run<1, 0>("fp8 scaled 16x16x128", src, dst);
    run<0, 1>("int8 16x16x64 + maxima", src, dst);
    run<1, 1>("fp8 scaled 16x16x128 + maxima", src, dst);
    run<0, 0>("int8 16x16x64", src, dst);
    run<1, 0>("fp8 scaled 16x16x128", src, dst);
    return 0;
}
