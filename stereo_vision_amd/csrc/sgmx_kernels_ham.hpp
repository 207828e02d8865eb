// sgmx_kernels_ham.hpp -- thresholded Hamming matching of binary signatures, the rest of the
// core/hamming library next to compute_hamming (core/hamming.h:73-104):
//   match_hamming_count / match_hamming_thres_prealloc   core/hamming.c:300-322, 725-765
//   crossmatch_hamming_count / crossmatch_hamming_prealloc  core/hamming.c:385-407, 810-839
// The reference emits the matches of its double loop (query i outer, base j inner; cross matching
// only j > i) in loop order.  Here one wave owns one query: a count pass, an exclusive prefix sum
// over the queries, and an emit pass in which the wave walks the base signatures 64 at a time and
// places its matches by ballot rank -- the same order, no atomics.
#pragma once
#include "sgmx_kernels.hpp"

namespace sgmx {

// popcount(a xor b) over ncodes bytes; wa = query words already in registers when ncodes <= 16
__device__ __forceinline__ u32 ham_dist(const uint8_t *__restrict__ pa, const uint8_t *__restrict__ pb, int ncodes)
{
    u32 s = 0;
    int k = 0;
    if ((ncodes & 3) == 0)
        for (; k < ncodes; k += 4)
            s += __builtin_popcount(*reinterpret_cast<const u32 *>(pa + k) ^ *reinterpret_cast<const u32 *>(pb + k));
    for (; k < ncodes; k++)
        s += __builtin_popcount((u32)(pa[k] ^ pb[k]));
    return s;
}

// counts[i] = number of j (j > i when cross) with dist(a_i, b_j) <= ht.  One wave per query.
__global__ __launch_bounds__(256) void ham_count_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int n1,
                                                        int n2, int ncodes, int ht, int cross,
                                                        unsigned long long *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n1)
        return;
    const uint8_t *pa = a + (size_t)i * ncodes;
    unsigned long long n = 0;
    for (int j0 = cross ? i + 1 : 0; j0 < n2; j0 += 64) {
        const int j = j0 + lane;
        const bool hit = j < n2 && (int)ham_dist(pa, b + (size_t)j * ncodes, ncodes) <= ht;
        n += (unsigned long long)__builtin_popcountll(__ballot(hit));
    }
    if (lane == 0)
        counts[i] = n;
}

// In-place exclusive prefix sum of n 64-bit counts by one workgroup; total[0] = the sum.
__global__ __launch_bounds__(1024) void ham_scan_kernel(unsigned long long *__restrict__ v, int n,
                                                        unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = lo + per < n ? lo + per : n;
    unsigned long long s = 0;
    for (int k = lo; k < hi; k++)
        s += v[k];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; k++) {
            const unsigned long long x = part[k];
            part[k] = run;
            run += x;
        }
        total[0] = run;
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int k = lo; k < hi; k++) {
        const unsigned long long x = v[k];
        v[k] = run;
        run += x;
    }
}

// Matches of query i go to positions offs[i] ... in loop order: idx[2p] = i, idx[2p+1] = j, hams[p] = distance.
// Positions >= capacity are dropped (the host reports the overflow).
__global__ __launch_bounds__(256) void ham_emit_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int n1,
                                                       int n2, int ncodes, int ht, int cross,
                                                       const unsigned long long *__restrict__ offs,
                                                       unsigned long long capacity, int *__restrict__ idx,
                                                       uint16_t *__restrict__ hams)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n1)
        return;
    const uint8_t *pa = a + (size_t)i * ncodes;
    unsigned long long pos = offs[i];
    for (int j0 = cross ? i + 1 : 0; j0 < n2; j0 += 64) {
        const int j = j0 + lane;
        u32 h = 0;
        bool hit = false;
        if (j < n2) {
            h = ham_dist(pa, b + (size_t)j * ncodes, ncodes);
            hit = (int)h <= ht;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const unsigned long long p = pos + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (p < capacity) {
                idx[2 * p] = i;
                idx[2 * p + 1] = j;
                hams[p] = (uint16_t)h;
            }
        }
        pos += (unsigned long long)__builtin_popcountll(m);
    }
}

} // namespace sgmx
