// The inner step of the RANSAC matrix-core pre-test in isolation: per iteration a wave makes 8 products
// (v_mfma_f32_16x16x32_f16, C = 0, four tiles x {num, q}) and tests them (per tile two packed FMAs, two
// minima, a compare and a branch that is never taken).  Four waves per SIMD, every CU busy.  Reports the
// SIMD time per wave-iteration, to be held against 8 x 16 = 128 cycles of matrix-pipe time.
//   MODE 0: products only (results kept alive without an instruction)
//   MODE 1: products + tests
//   MODE 2: products accumulate into their own results (C = previous) + tests
//   MODE 3: as 1, the two B fragments of every iteration loaded from global memory one iteration ahead
//   MODE 4: tests only (on register values)
//   MODE 5: as 1 with all 8 products first, then all tests (no interleaving by the source order)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k(const half8 *src, int *dst, int iters)
{
    const int lane = threadIdx.x & 63;
    half8 An[4], Aq[4];
    for (int t = 0; t < 4; t++) {
        An[t] = src[(lane * 7 + t) & 1023];
        Aq[t] = src[(lane * 13 + t + 64) & 1023];
    }
    half8 bn = src[lane], bq = src[lane + 64];
    const floatx4 zero = {0, 0, 0, 0};
    floatx4 nu[4], qq[4];
    for (int t = 0; t < 4; t++) nu[t] = qq[t] = zero;
    int hits = 0;
    const half8 *ops = src + lane;
    for (int it = 0; it < iters; it++) {
        half8 bn2 = bn, bq2 = bq;
        if (MODE == 3) {
            const half8 *nx = ops + ((it + 1) & 7) * 128;
            bn2 = nx[0];
            bq2 = nx[64];
        }
        if (MODE != 4) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                nu[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(An[t], bn, MODE == 2 ? nu[t] : zero, 0, 0, 0);
                qq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aq[t], bq, MODE == 2 ? qq[t] : zero, 0, 0, 0);
            }
        }
        if (MODE == 5) {
#pragma unroll
            for (int t = 0; t < 4; t++) asm volatile("" : "+v"(nu[t]), "+v"(qq[t]));
        }
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 4; t++) asm volatile("" ::"v"(nu[t]), "v"(qq[t]));
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const f32x2 dA = __builtin_elementwise_fma(f32x2{nu[t][0], nu[t][1]}, f32x2{nu[t][0], nu[t][1]}, f32x2{qq[t][0], qq[t][1]});
                const f32x2 dB = __builtin_elementwise_fma(f32x2{nu[t][2], nu[t][3]}, f32x2{nu[t][2], nu[t][3]}, f32x2{qq[t][2], qq[t][3]});
                const float least = fminf(fminf(dA.x, dA.y), fminf(dB.x, dB.y));
                if (__builtin_amdgcn_fcmpf(least, -1.0e30f, 4) != 0) hits += 1 + t; // never
            }
            if (MODE == 4) { // keep the values moving so that nothing is hoisted
                nu[it & 3][0] += 1.0f;
            }
        }
        if (MODE == 3) {
            bn = bn2;
            bq = bq2;
        } else {
            bn[0] = (_Float16)(float)(it & 7);
        }
    }
    dst[blockIdx.x * 256 + threadIdx.x] = hits + (int)nu[0][0] + (int)qq[3][1];
}

template <int MODE> static void run(const char *name, const half8 *src, int *dst)
{
    const int iters = 40000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, 400);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 4); // SIMD time per wave-iteration (four waves per SIMD)
    const double tf = (double)blocks * 4 * iters * 8 * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %6.1f ns per wave-iteration (= %5.0f cycles at 2.1 GHz)  %7.1f TF/s\n", name, ms, ns, ns * 2.1, MODE == 4 ? 0.0 : tf);
}

int main()
{
    std::vector<_Float16> h(1024 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
    half8 *src;
    int *dst;
    hipMalloc(&src, 1024 * 16); hipMalloc(&dst, 256 * 4 * 256 * 4);
    hipMemcpy(src, h.data(), 1024 * 16, hipMemcpyHostToDevice);
    run<0>("products only", src, dst);
    run<1>("products + tests", src, dst);
    run<2>("accumulating products + tests", src, dst);
    run<3>("products + tests, operands from global", src, dst);
    run<4>("tests only", src, dst);
    run<5>("all products, then all tests", src, dst);
    run<0>("products only (again)", src, dst);
    return 0;
}
