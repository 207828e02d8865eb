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

// The same prefix sum for many counts, in three launches: every workgroup scans its own 1024 counts in place
// (coalesced loads, wave scans by DPP-free shuffles through LDS) and leaves their sum in bsum; ham_scan_kernel scans
// the sums; ham_scan_add_kernel adds every workgroup's offset.  (One workgroup walking 100 000 counts in per-thread
// chunks took 140 us, as long as the counting kernel itself.)
__global__ __launch_bounds__(1024) void ham_scan_blocks_kernel(unsigned long long *__restrict__ v, int n,
                                                               unsigned long long *__restrict__ bsum)
{
    __shared__ unsigned long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int k = blockIdx.x * 1024 + t;
    const unsigned long long x = k < n ? v[k] : 0ull;
    unsigned long long inc = x; // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(inc, d, 64);
        if (lane >= d)
            inc += y;
    }
    if (lane == 63)
        wsum[wv] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int w2 = 0; w2 < wv; w2++)
        base += wsum[w2];
    if (k < n)
        v[k] = base + inc - x;
    if (t == 1023)
        bsum[blockIdx.x] = base + inc;
}

__global__ __launch_bounds__(1024) void ham_scan_add_kernel(unsigned long long *__restrict__ v, int n,
                                                            const unsigned long long *__restrict__ boff)
{
    const int k = blockIdx.x * 1024 + threadIdx.x;
    if (k < n)
        v[k] += boff[blockIdx.x];
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
// compute_hamming (core/hamming.c:167-236: dis[j * na + i] = popcount(a_i xor b_j), fast paths for 4 / 8 / 16-byte
// codes) for signatures of NW dwords.  The matrix is write-bound (2 bytes out per distance, 8 NW VALU lane
// operations): a thread keeps HAM_II = 8 neighbouring signatures a_i in registers, walks HAM_JT rows b_j (one
// address per wave: scalar loads) and stores the 8 distances of a row as one 16-byte piece -- a wave writes 1 KB
// of a matrix row per store.
// ---------------------------------------------------------------------------
constexpr int HAM_II = 8, HAM_JT = 32;

// 8 uint16 (packed in 4 dwords) to a 2-byte aligned address: one 16-byte store when it is 4-byte aligned, else a
// leading and a trailing halfword around three dwords
__device__ __forceinline__ void ham_store8(uint16_t *p, const u32 (&v)[4])
{
    if ((reinterpret_cast<uintptr_t>(p) & 2) == 0) {
        __builtin_memcpy(__builtin_assume_aligned(p, 4), v, 16);
    } else {
        p[0] = (uint16_t)v[0];
        u32 m[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
            m[k] = __builtin_amdgcn_alignbit(v[k + 1], v[k], 16);
        __builtin_memcpy(__builtin_assume_aligned(p + 1, 4), m, 12);
        p[7] = (uint16_t)(v[3] >> 16);
    }
}

template <int NW>
__global__ __launch_bounds__(256) void hamming_matrix_tiled_kernel(uint16_t *__restrict__ dis, const u32 *__restrict__ a,
                                                                   const u32 *__restrict__ b, int na, int nb)
{
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * HAM_II;
    if (i0 >= na)
        return;
    u32 wa[HAM_II][NW];
#pragma unroll
    for (int k = 0; k < HAM_II; k++) {
        const int i = i0 + k < na ? i0 + k : na - 1;
        __builtin_memcpy(wa[k], a + (size_t)i * NW, 4 * NW);
    }
    const int j0 = blockIdx.y * HAM_JT, j1 = j0 + HAM_JT < nb ? j0 + HAM_JT : nb;
    const bool full = i0 + HAM_II <= na;
    for (int j = j0; j < j1; j++) {
        u32 wb[NW];
        __builtin_memcpy(wb, b + (size_t)j * NW, 4 * NW); // one address per workgroup: scalar loads
        u32 d[HAM_II];
#pragma unroll
        for (int k = 0; k < HAM_II; k++) {
            u32 s = 0;
#pragma unroll
            for (int n = 0; n < NW; n++)
                s += (u32)__builtin_popcount(wa[k][n] ^ wb[n]);
            d[k] = s;
        }
        uint16_t *row = dis + (size_t)j * na + i0;
        if (full) {
            u32 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                v[k] = d[2 * k] | (d[2 * k + 1] << 16);
            ham_store8(row, v);
        } else {
#pragma unroll
            for (int k = 0; k < HAM_II; k++)
                if (i0 + k < na)
                    row[k] = (uint16_t)d[k];
        }
    }
}

// ---------------------------------------------------------------------------
// The matchers for signatures of NW dwords: a wave owns HAM_QW = 8 consecutive queries, whose words sit in
// registers (one address per wave), and walks the base signatures 64 at a time -- every lane loads its b_j once
// for the 8 queries (the one-query-per-wave kernels above read the whole base set once per query).  EMIT = false
// counts, EMIT = true places the matches of query i from offs[i] on by ballot rank: the reference's loop order
// (query outer, base inner; cross: j > i only).
// ---------------------------------------------------------------------------
constexpr int HAM_QW = 8;

template <int NW, bool EMIT>
__global__ __launch_bounds__(256) void ham_match_tiled_kernel(const u32 *__restrict__ a, const u32 *__restrict__ b, int n1, int n2,
                                                              int ht, int cross, unsigned long long *__restrict__ counts,
                                                              unsigned long long capacity, int *__restrict__ idx,
                                                              uint16_t *__restrict__ hams)
{
    const int lane = threadIdx.x & 63;
    const int q0 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))) * HAM_QW;
    if (q0 >= n1)
        return;
    u32 wa[HAM_QW][NW];
    unsigned long long pos[HAM_QW];
#pragma unroll
    for (int k = 0; k < HAM_QW; k++) {
        const int i = q0 + k < n1 ? q0 + k : n1 - 1;
        __builtin_memcpy(wa[k], a + (size_t)i * NW, 4 * NW);
        pos[k] = EMIT ? counts[i] : 0ull; // (EMIT: counts holds the exclusive prefix sums)
    }
    for (int jb = cross ? q0 + 1 : 0; jb < n2; jb += 64) {
        const int j = jb + lane;
        u32 wb[NW] = {};
        if (j < n2)
            __builtin_memcpy(wb, b + (size_t)j * NW, 4 * NW);
#pragma unroll
        for (int k = 0; k < HAM_QW; k++) {
            u32 h = 0;
#pragma unroll
            for (int n = 0; n < NW; n++)
                h += (u32)__builtin_popcount(wa[k][n] ^ wb[n]);
            const bool hit = j < n2 && (int)h <= ht && q0 + k < n1 && (!cross || j > q0 + k);
            const unsigned long long m = __ballot(hit);
            if (EMIT && hit) {
                const unsigned long long p = pos[k] + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (p < capacity) {
                    idx[2 * p] = q0 + k;
                    idx[2 * p + 1] = j;
                    hams[p] = (uint16_t)h;
                }
            }
            pos[k] += (unsigned long long)__builtin_popcountll(m);
        }
    }
    if (!EMIT && lane == 0) {
#pragma unroll
        for (int k = 0; k < HAM_QW; k++)
            if (q0 + k < n1)
                counts[q0 + k] = pos[k];
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
