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

// ---------------------------------------------------------------------------
// match_hamming_thres_nt (core/hamming.h:111-112, core/hamming.c:863-919): the OpenMP form cuts both sets into
// blocks of HAMBLOCK = 128 signatures, counts every block pair, and lets every block pair write its matches at
// its prefix offset -- so the output is ordered by block pair first, and the indices it stores are the ones
// match_hamming_thres_prealloc sees: LOCAL to the block pair (0..127), not global.  Block pair bl covers queries
// (bl / nblock1) * 128 ... and base signatures (bl % nblock1) * 128 ... (:876-880; the reference divides and takes
// the remainder by nblock1 for both, which only indexes inside the two sets when nblock1 == nblock2 -- the host
// side refuses the other shapes).  Here one wave owns one (block pair, local query) item; items in the reference's
// order, then count / prefix sum / emit exactly as above.
// ---------------------------------------------------------------------------
constexpr int HAMBLOCK = 128;

struct HamBlocked {
    int nblock1; // blocks of the first set = divisor of the block-pair index
    int n1, n2;
};

__device__ __forceinline__ bool ham_blocked_item(const HamBlocked &g, long item, int &i, int &j0, int &j1)
{
    const int bl = (int)(item / HAMBLOCK), il = (int)(item % HAMBLOCK);
    const int s1 = (bl / g.nblock1) * HAMBLOCK, s2 = (bl % g.nblock1) * HAMBLOCK;
    i = s1 + il;
    j0 = s2;
    j1 = s2 + HAMBLOCK < g.n2 ? s2 + HAMBLOCK : g.n2;
    return i < g.n1;
}

__global__ __launch_bounds__(256) void ham_count_blocked_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                                HamBlocked g, long nitems, int ncodes, int ht,
                                                                unsigned long long *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems)
        return;
    int i, j0, j1;
    unsigned long long n = 0;
    if (ham_blocked_item(g, item, i, j0, j1)) {
        const uint8_t *pa = a + (size_t)i * ncodes;
        for (int jb = j0; jb < j1; jb += 64) {
            const int j = jb + lane;
            const bool hit = j < j1 && (int)ham_dist(pa, b + (size_t)j * ncodes, ncodes) <= ht;
            n += (unsigned long long)__builtin_popcountll(__ballot(hit));
        }
    }
    if (lane == 0)
        counts[item] = n;
}

__global__ __launch_bounds__(256) void ham_emit_blocked_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                               HamBlocked g, long nitems, int ncodes, int ht,
                                                               const unsigned long long *__restrict__ offs,
                                                               int *__restrict__ idx, uint16_t *__restrict__ hams)
{
    const int lane = threadIdx.x & 63;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nitems)
        return;
    int i, j0, j1;
    if (!ham_blocked_item(g, item, i, j0, j1))
        return;
    const uint8_t *pa = a + (size_t)i * ncodes;
    unsigned long long pos = offs[item];
    for (int jb = j0; jb < j1; jb += 64) {
        const int j = jb + lane;
        u32 h = 0;
        bool hit = false;
        if (j < j1) {
            h = ham_dist(pa, b + (size_t)j * ncodes, ncodes);
            hit = (int)h <= ht;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const unsigned long long p = pos + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            idx[2 * p] = (int)(item % HAMBLOCK); // block-local, as the reference stores them
            idx[2 * p + 1] = j - j0;
            hams[p] = (uint16_t)h;
        }
        pos += (unsigned long long)__builtin_popcountll(m);
    }
}

// (idx, hams) -> hammatch_t records { int qid; int bid; uint16 score; } (core/hamming.h:52-56; 12 bytes each, the
// two padding bytes zeroed)
__global__ __launch_bounds__(256) void ham_pack_kernel(const int *__restrict__ idx, const uint16_t *__restrict__ hams,
                                                       unsigned long long n, int *__restrict__ out)
{
    const unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    out[3 * p] = idx[2 * p];
    out[3 * p + 1] = idx[2 * p + 1];
    out[3 * p + 2] = (int)hams[p];
}

} // namespace sgmx
