// The bound loop's inner part as the matcher has it -- B fragments from LDS, four row tiles per wave, chain maxima, a
// branch to a rare path -- with the number of 32-column groups per branch (G) and the blocks per CU as parameters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int G, int OCC, bool BARRIER, bool BRANCH = true, bool TEST = true>
__global__ __launch_bounds__(256, OCC) void loop(const unsigned *src, int iters, int *out, float thr)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 128];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 128 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = src[i];
    __syncthreads();
    int8v a6[4];
    for (int t = 0; t < 4; t++) {
        for (int w = 0; w < 6; w++) a6[t][w] = (int)src[(t * 64 + lane) * 8 + w];
        a6[t][6] = a6[t][7] = 0;
    }
    int n = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g0 = 0; g0 < 2; g0 += G) {
            float4v acc[G][2][4];
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    const int row = ((g0 + g + it) & 1) * 32 + ct * 16 + (lane & 15);
                    int4v b[2];
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int chunk = s * 4 + (lane >> 4);
                        b[s] = *reinterpret_cast<const int4v *>(&lds[(row >> 1) * 256 + ((((row & 1) * 8 + chunk) ^ ((row >> 1) & 15)) << 4)]);
                    }
                    const int8v b6 = {b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], 0, 0};
                    const float4v zero = {0, 0, 0, 0};
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        acc[g][ct][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6[t], b6, zero, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            if (!TEST) {
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int t = 0; t < 4; t++) asm volatile("" ::"v"(acc[g][ct][t]));
                continue;
            }
            float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        m[(2 * t) & 3] = fmaxf(fmaxf(m[(2 * t) & 3], acc[g][ct][t][0]), acc[g][ct][t][1]);
                        m[(2 * t + 1) & 3] = fmaxf(fmaxf(m[(2 * t + 1) & 3], acc[g][ct][t][2]), acc[g][ct][t][3]);
                    }
            const bool hit = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])) > thr;
            if (!BRANCH) n += __builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
            else if (__builtin_amdgcn_ballot_w64(hit) != 0) { // rare: every accumulator is looked at
#pragma unroll
                for (int g = 0; g < G; g++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                if (acc[g][ct][t][q] > thr) out[64 + ((n++) & 1023)] = lane;
            }
        }
        if (BARRIER) __syncthreads();
    }
    if (n == 12345) out[0] = n;
}

template <int G, int OCC, bool BARRIER, bool BRANCH = true, bool TEST = true> static void run(const char *name, const unsigned *src, int *out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * OCC;
    hipLaunchKernelGGL((loop<G, OCC, BARRIER, BRANCH, TEST>), dim3(grid), dim3(256), 0, 0, src, 10, out, 1e30f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop<G, OCC, BARRIER, BRANCH, TEST>), dim3(grid), dim3(256), 0, 0, src, iters, out, 1e30f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tiles = (double)grid * 4 * iters * 4 * 4; // 16x16x128 tiles
    printf("%-40s %7.3f ms  %6.2f PFLOP/s\n", name, ms, tiles * 65536 / (ms * 1e-3) / 1e15);
}
int main()
{
    std::vector<unsigned> h(64 * 1024);
    srand(3);
    for (auto &x : h) x = (unsigned)rand() * 2654435761u;
    unsigned *src; int *out;
    CHECK(hipMalloc(&src, h.size() * 4)); CHECK(hipMalloc(&out, 8192));
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<1, 2, false, true, false>("no test, 2 blocks/CU", src, out);
    run<1, 4, false, true, false>("no test, 4 blocks/CU", src, out);
    run<1, 5, false, true, false>("no test, 5 blocks/CU", src, out);
    run<1, 5, true, true, false>("no test, barrier, 5 blocks/CU", src, out);
    run<1, 2, false, false>("no branch, test per 32 columns, 2 blocks/CU", src, out);
    run<1, 4, false, false>("no branch, test per 32 columns, 4 blocks/CU", src, out);
    run<1, 5, false, false>("no branch, test per 32 columns, 5 blocks/CU", src, out);
    run<2, 4, false, false>("no branch, test per 64 columns, 4 blocks/CU", src, out);
    run<1, 5, true, false>("no branch, barrier, 5 blocks/CU", src, out);
    run<1, 5, false>("branch per 32 columns, 5 blocks/CU", src, out);
    run<1, 5, true>("same, barrier per 64 columns", src, out);
    run<2, 4, false>("branch per 64 columns, 4 blocks/CU", src, out);
    run<2, 4, true>("same, barrier per 64 columns", src, out);
    run<1, 4, false>("branch per 32 columns, 4 blocks/CU", src, out);
    run<1, 2, false>("branch per 32 columns, 2 blocks/CU", src, out);
    run<2, 2, false>("branch per 64 columns, 2 blocks/CU", src, out);
    run<2, 3, false>("branch per 64 columns, 3 blocks/CU", src, out);
    return 0;
}
