// Mutual check, ordered compaction and the CSR offsets of a chunk of pairs: shared by the float and
// the binary matcher.  The kernels sit in an unnamed namespace so that every translation unit that
// includes this header gets its own copy (device code is not linked across files).
#pragma once
#include "match_types.h"

typedef unsigned long long u64;

namespace {

// ---------------------------------------------------------------------------------------
// mutual check + ordered compaction (match.cpp:84-91)
// ---------------------------------------------------------------------------------------
// One block per pair.  WRITE = false counts, WRITE = true writes (i, j) at offsets[p].
template <bool WRITE>
__global__ __launch_bounds__(256) void match_mutual_kernel(const PairRec *__restrict__ pairs,
                                                           const u64 *__restrict__ best,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ offsets,
                                                           int2 *__restrict__ out,
                                                           int32_t *__restrict__ m1out,
                                                           int32_t *__restrict__ m2out)
{
    __shared__ int waveCnt[4];
    const int p = blockIdx.x;
    const PairRec pr = pairs[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 *rowKey = best + pr.bestRow;
    const u64 *colKey = best + pr.bestCol;
    int2 *dst = WRITE ? out + offsets[p] : nullptr;
    int running = 0;
    for (int base = 0; base < pr.nA; base += 256) {
        const int i = base + tid;
        bool ok = false;
        int j = -1;
        if (i < pr.nA) {
            const u64 kr = rowKey[i];
            if (kr != 0) {
                j = (int)(0xFFFFFFFFu - (unsigned)kr);
                const u64 kc = colKey[j];
                ok = kc != 0 && (int)(0xFFFFFFFFu - (unsigned)kc) == i;
            }
            if (WRITE && m1out) m1out[i] = j;
        }
        const u64 mask = __builtin_amdgcn_ballot_w64(ok);
        const int prefix = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) waveCnt[wave] = __builtin_popcountll(mask);
        __syncthreads();
        int off = running, tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k < wave) off += waveCnt[k];
            tot += waveCnt[k];
        }
        if (WRITE && ok) dst[off + prefix] = make_int2(i, j);
        running += tot;
        __syncthreads();
    }
    if (!WRITE && tid == 0) counts[p] = running;
    if (WRITE && m2out) {
        for (int j = tid; j < pr.nB; j += 256) {
            const u64 kc = colKey[j];
            m2out[j] = kc ? (int)(0xFFFFFFFFu - (unsigned)kc) : -1;
        }
    }
}

// offsets[first + k + 1] = offsets[first] + sum(counts[0..k]); one block.
__global__ __launch_bounds__(1024) void match_scan_kernel(const int *__restrict__ counts, int n,
                                                          int64_t *__restrict__ offsets, int64_t first)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = tid * per, hi = min(n, lo + per);
    int64_t s = 0;
    for (int k = lo; k < hi; k++) s += counts[k];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = offsets[first] + (tid ? part[tid - 1] : 0);
    for (int k = lo; k < hi; k++) {
        run += counts[k];
        offsets[first + k + 1] = run;
    }
}


} // namespace
