// Register blocking of the e2m3 bound loop: T row tiles per wave share one B fragment from LDS (24 bytes per lane);
// the test runs per 16 columns on the T * 4 accumulators of a lane.  LDS rows vary with the iteration so that nothing
// is loop-invariant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int T, int OCC, bool TEST>
__global__ __launch_bounds__(256, OCC) void loop(const unsigned *src, int iters, int *out, float thr)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 128];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 128 / 4; i += 256) reinterpret_cast<unsigned *>(lds)[i] = src[i];
    __syncthreads();
    int8v a6[T];
    for (int t = 0; t < T; t++) {
        for (int w = 0; w < 6; w++) a6[t][w] = (int)src[(t * 64 + lane) * 8 + w];
        a6[t][6] = a6[t][7] = 0;
    }
    int n = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) {
            float4v acc[T];
            const int row = ((ct + it) & 3) * 16 + (lane & 15);
            int4v b[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int chunk = s * 4 + (lane >> 4);
                b[s] = *reinterpret_cast<const int4v *>(&lds[(row >> 1) * 256 + ((((row & 1) * 8 + chunk) ^ ((row >> 1) & 15)) << 4)]);
            }
            const int8v b6 = {b[0][0], b[0][1], b[0][2], b[0][3], b[1][0], b[1][1], 0, 0};
            const float4v zero = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < T; t++)
                acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6[t], b6, zero, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (!TEST) {
#pragma unroll
                for (int t = 0; t < T; t++) asm volatile("" ::"v"(acc[t]));
                continue;
            }
            float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int t = 0; t < T; t++) {
                m[(2 * t) & 3] = fmaxf(fmaxf(m[(2 * t) & 3], acc[t][0]), acc[t][1]);
                m[(2 * t + 1) & 3] = fmaxf(fmaxf(m[(2 * t + 1) & 3], acc[t][2]), acc[t][3]);
            }
            const bool hit = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])) > thr;
            if (__builtin_amdgcn_ballot_w64(hit) != 0) {
#pragma unroll
                for (int t = 0; t < T; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (acc[t][q] > thr) out[64 + ((n++) & 1023)] = lane;
            }
        }
    }
    if (n == 12345) out[0] = n;
}

template <int T, int OCC, bool TEST> static void run(const char *name, const unsigned *src, int *out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * OCC;
    hipLaunchKernelGGL((loop<T, OCC, TEST>), dim3(grid), dim3(256), 0, 0, src, 10, out, 1e30f);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop<T, OCC, TEST>), dim3(grid), dim3(256), 0, 0, src, iters, out, 1e30f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tiles = (double)grid * 4 * iters * 4 * T;
    printf("%-44s %7.3f ms  %6.2f PFLOP/s\n", name, ms, tiles * 65536 / (ms * 1e-3) / 1e15);
}
int main()
{
    std::vector<unsigned> h(64 * 1024);
    srand(3);
    for (auto &x : h) x = (unsigned)rand() * 2654435761u;
    unsigned *src; int *out;
    CHECK(hipMalloc(&src, h.size() * 4)); CHECK(hipMalloc(&out, 8192));
    CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<4, 5, false>("4 tiles per fragment, no test, 5 blocks/CU", src, out);
    run<4, 5, true>("4 tiles per fragment, test, 5 blocks/CU", src, out);
    run<8, 4, false>("8 tiles per fragment, no test, 4 blocks/CU", src, out);
    run<8, 4, true>("8 tiles per fragment, test, 4 blocks/CU", src, out);
    run<8, 3, true>("8 tiles per fragment, test, 3 blocks/CU", src, out);
    run<8, 2, true>("8 tiles per fragment, test, 2 blocks/CU", src, out);
    run<16, 2, false>("16 tiles per fragment, no test, 2 blocks/CU", src, out);
    run<16, 2, true>("16 tiles per fragment, test, 2 blocks/CU", src, out);
    return 0;
}
