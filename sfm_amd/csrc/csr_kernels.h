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
// One block per pair.  WRITE = false counts, WRITE = true writes (i, j) at offsets[p].  A thread has four rows
// in flight at a time (the row key leads to the column key: two dependent loads per row).
// offsetsOut / totalOut (WRITE only, either may be null): a launch over a single pair needs no counting pass and
// no scan; it writes offsetsOut[p + 1] = offsets[p] + count and totalOut[0] = count itself.
// WAVES: waves per block (a launch over one pair has one block only and takes sixteen).
template <bool WRITE, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES) void match_mutual_kernel(const PairRec *__restrict__ pairs,
                                                           const u64 *__restrict__ best,
                                                           int *__restrict__ counts,
                                                           const int64_t *__restrict__ offsets,
                                                           int2 *__restrict__ out,
                                                           int32_t *__restrict__ m1out,
                                                           int32_t *__restrict__ m2out,
                                                           int64_t *__restrict__ offsetsOut,
                                                           int64_t *__restrict__ totalOut)
{
    constexpr int U = 4;
    constexpr int T = 64 * WAVES;
    __shared__ int waveCnt[U][WAVES];
    const int p = blockIdx.x;
    const PairRec pr = pairs[p];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 *rowKey = best + pr.bestRow;
    const u64 *colKey = best + pr.bestCol;
    const int64_t first = WRITE ? offsets[p] : 0;
    int2 *dst = WRITE ? out + first : nullptr;
    int running = 0;
    for (int base = 0; base < pr.nA; base += T * U) {
        u64 kr[U], kc[U];
        int j[U], prefix[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + u * T + tid;
            kr[u] = (i < pr.nA) ? rowKey[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            j[u] = kr[u] ? (int)(0xFFFFFFFFu - (unsigned)kr[u]) : -1;
            kc[u] = kr[u] ? colKey[j[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = base + u * T + tid;
            ok[u] = kc[u] != 0 && (int)(0xFFFFFFFFu - (unsigned)kc[u]) == i;
            if (WRITE && m1out && i < pr.nA) m1out[i] = j[u];
            const u64 mask = __builtin_amdgcn_ballot_w64(ok[u]);
            prefix[u] = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (lane == 0) waveCnt[u][wave] = __builtin_popcountll(mask);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            int off = running, tot = 0;
#pragma unroll
            for (int k = 0; k < WAVES; k++) {
                if (k < wave) off += waveCnt[u][k];
                tot += waveCnt[u][k];
            }
            if (WRITE && ok[u]) dst[off + prefix[u]] = make_int2(base + u * T + tid, j[u]);
            running += tot;
        }
        __syncthreads();
    }
    if (!WRITE && tid == 0) counts[p] = running;
    if (WRITE && tid == 0) {
        if (offsetsOut) offsetsOut[p + 1] = first + running;
        if (totalOut) totalOut[0] = running;
    }
    if (WRITE && m2out) {
        for (int j2 = tid; j2 < pr.nB; j2 += T) {
            const u64 k2 = colKey[j2];
            m2out[j2] = k2 ? (int)(0xFFFFFFFFu - (unsigned)k2) : -1;
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
