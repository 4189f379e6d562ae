// Layout check of v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands against a host sum, on exact data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned *A, const unsigned *B, float *C)
{
    const int lane = threadIdx.x;
    int8v a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    // lane l: row l & 15, k-block l >> 4 (32 elements, 24 bytes = 6 dwords)
    for (int w = 0; w < 6; w++) {
        a[w] = (int)A[((lane & 15) * 4 + (lane >> 4)) * 6 + w];
        b[w] = (int)B[((lane & 15) * 4 + (lane >> 4)) * 6 + w];
    }
    float4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int q = 0; q < 4; q++) C[(4 * (lane >> 4) + q) * 16 + (lane & 15)] = c[q];
}

static float dec(unsigned v)
{
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float x = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
    return s ? -x : x;
}
int main()
{
    std::vector<unsigned char> ea(16 * 128), eb(16 * 128);
    srand(7);
    for (auto &x : ea) x = rand() & 63;
    for (auto &x : eb) x = rand() & 63;
    auto pack = [](const std::vector<unsigned char> &e) {
        std::vector<unsigned> out(16 * 4 * 6, 0);
        for (int r = 0; r < 16; r++)
            for (int kb = 0; kb < 4; kb++)
                for (int j = 0; j < 32; j++) {
                    const unsigned v = e[r * 128 + kb * 32 + j];
                    const int bit = 6 * j;
                    unsigned *w = &out[(r * 4 + kb) * 6];
                    w[bit >> 5] |= v << (bit & 31);
                    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= v >> (32 - (bit & 31));
                }
        return out;
    };
    auto pa = pack(ea), pb = pack(eb);
    unsigned *dA, *dB; float *dC;
    hipMalloc(&dA, pa.size() * 4); hipMalloc(&dB, pb.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, pa.data(), pa.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, pb.data(), pb.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    float C[256];
    hipMemcpy(C, dC, sizeof C, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int kk = 0; kk < 128; kk++) s += (double)dec(ea[i * 128 + kk]) * dec(eb[j * 128 + kk]);
            if (fabs(s - C[i * 16 + j]) > 1e-3) { if (bad < 5) printf("C[%d][%d] = %f, host %f\n", i, j, C[i * 16 + j], s); bad++; }
        }
    printf("%d cells differ\n", bad);
    return bad != 0;
}
