// sgmx_kernels_v2.hpp -- the census/Hamming fast path (P2 <= 255): no data-cost
// volume and no summed-cost volume in HBM.
//
//   * The Hamming data cost C(p,d) = popcount(censL[p] ^ censR[p-d]) (stereo_sgm.cc:75-101,
//     stereo_costs.h:120-132) is recomputed inside the sweeps from the two census images, which
//     stay cache resident (2 x 1.8 MB per pair), instead of being streamed from a 59 MB volume.
//   * aggregate_path (stereo_sgm.h:119-147) gives L = C + E with
//         E = min(prior[d], prior[d+-1] + P1, m + P2) - m,   0 <= E <= P2,
//     so with P2 <= 255 every direction stores only E -- as G = P2 - E, what the sweeps compute -- as ONE
//     byte per voxel into its own volume (write-only) instead of read-modify-writing a uint16 sum
//     (sum_costs, stereo_sgm.h:105-117).
//     On start pixels L = L_{k-1} + C (stereo_sgm.cc:201,218,236,254) can be larger: they are
//     image-border pixels, their E is 0 and L_{k-1} goes into the small border accumulator SB
//     (uint16 ring).  The final kernel rebuilds
//         S(p,d) = N C(p,d) + sum_k E_k(p,d) (+ SB on border pixels) = N (C + P2) - sum_k G_k (+ SB)
//     exactly (integers), and does WTA / right-view argmin / LR check / subpixel on it.
//   HBM bytes per voxel: N (G writes) + N (G reads) = 16 at N = 8, against 41 for the materialised
//   C + S formulation that the algorithmic-byte model counts.  With that the sweeps run at what a loop of
//   their VALU work and their stores can write, so the mapping minimises instructions per pixel:
//
// Lane mapping ("16 x RP x 2, mirrored"): a pixel's Dp = 32 RP disparities live in ONE 16-lane DPP
// row; a wave carries 4 pixels -- 4 different paths of the same direction, walked in lock step.
// Lane q of a row holds RP packed uint16 pairs, register j = ( d = RP q + j , d = Dp-1 - (RP q + j) ):
// the low halves run upwards from 0, the high halves downwards from Dp-1.
//   - The recurrence only needs the better of the neighbours d-1 and d+1; with the mirrored halves those are
//     registers j-1 and j+1 for BOTH halves (d-1 of the low half and d+1 of the high half sit in register
//     j-1, and vice versa), no shuffles inside a lane.
//   - "Register -1" is lane q-1's top register (one row_shr DPP; lane 0 has no neighbour on either
//     half).  "Register RP" is lane q+1's register 0 (one row_shl DPP); in lane 15 the two halves of the
//     top register (d = Dp/2-1, Dp/2) are each other's neighbours, so it keeps its own top register with
//     the halves swapped.
//   - The extremum over d is a few packed 3-input operations, one fold of the two halves, and 4 row-rotate
//     DPP steps that leave it in every lane of the row, no scalar round trip.
//   Since round 2 the sweeps carry Q = K - L instead of L (bigger is better, maxima instead of minima, the
//   byte volumes hold G = P2 - E): see aggregate3_kernel.
//
// Memory layouts private to this path (d_lo = RP q + j, d_hi = Dp-1 - d_lo):
//   census left  uint32 [h][w];  census right uint32 [h][wp], wp = Dp + w, pixel x in column Dp + x
//     (so the RP words censR[x-d] of a lane window are one contiguous in-row load for every x >= 0;
//      costs with d > x are forced to 255, stereo_sgm.cc:43)
//   E_k   uint8  [h][w][Dp]     byte 2 (RP q + j) = G(d_lo), byte 2 (RP q + j) + 1 = G(d_hi),  G = P2 - E
//   BL/SB uint16 [2w+2h][Dp]    uint32 (RP q + j) = ( value(d_lo) , value(d_hi) << 16 )
#pragma once
#include "sgmx_kernels.hpp"

namespace sgmx {

// E bytes are written once and read once, a whole launch later: non-temporal stores (aux bit 1 = nt)
// keep them from evicting the census images out of L2 (measured: 0.84 -> 0.60 ms per sweep).
#ifndef SGMX_E_AUX
#define SGMX_E_AUX 2
#endif
constexpr int E_STORE_AUX = SGMX_E_AUX; // cache policy bits of a buffer store: 2 = nt
#ifndef SGMX_FUSE
#define SGMX_FUSE 1 // small launch groups: directions 0 / 1 and 5 / 6 side by side (0: A/B builds only)
#endif
#ifndef SGMX_WRAP
#define SGMX_WRAP 1 // diagonal sweeps walk wrapped diagonals (0: one row per diagonal, A/B builds only)
#endif
typedef short ss2 __attribute__((ext_vector_type(2)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

struct Agg2Args {
    int w, h, D, wp;   // cropped size, real disparity count, census-right row pitch
    int dx, dy;        // predecessor of (y,x) is (y-dy, x-dx)
    int P1, P2;
    int npaths, npairs;
    // Two directions of one launch group side by side (small groups only, where a sweep leaves most of the chip idle):
    // the horizontal direction k+1 starts on the image column where the diagonal direction k starts too, so its start
    // values L_k = L_{k-1} + C are known without running direction k: L_{k+1} = L_{k-1} + 2 C there.
    //   fuse = 1 (direction k): reads the ring, writes nothing into it (direction k+1 writes every border pixel
    //            after it in the reference's order), and leaves the carry of its column starts to direction k+1;
    //   fuse = 2 (direction k+1): start values BL + 2 C, carry 2 BL + C (its own and direction k's), ring reads from
    //            BL, ring writes into BLout (direction k still reads BL).
    int fuse;
    uint16_t *BLout; // fuse = 2: the ring this launch writes; otherwise unused (writes go to BL)
};

template <bool B> struct BoolC { static constexpr bool value = B; };

// Minimum over the 16 lanes of a row, left in every lane.  0xFFFFFFFF as `old` is the identity of
// umin, which lets the compiler fold each step into one v_min_u32_dpp.
__device__ __forceinline__ u32 row_min_u32(u32 x)
{
    x = min(x, dpp<0x121>(0xFFFFFFFFu, x));
    x = min(x, dpp<0x122>(0xFFFFFFFFu, x));
    x = min(x, dpp<0x124>(0xFFFFFFFFu, x));
    x = min(x, dpp<0x128>(0xFFFFFFFFu, x));
    return x;
}

// Maximum over the 16 lanes of a row, left in every lane (0 is the identity of umax: v_max_u32_dpp).
__device__ __forceinline__ u32 row_max_u32(u32 x)
{
    x = max(x, dpp<0x121>(0u, x));
    x = max(x, dpp<0x122>(0u, x));
    x = max(x, dpp<0x124>(0u, x));
    x = max(x, dpp<0x128>(0u, x));
    return x;
}

// max(a - b, 0) on both halves (v_pk_sub_u16 ... clamp)
__device__ __forceinline__ u32 pk_sub_sat(u32 a, u32 b)
{
    return __builtin_bit_cast(u32, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
// Three-input packed uint16 maximum.  gfx950 has no integer form; v_pk_maximum3_f16 is exact on halves below
// 0x7C00 (positive finite f16 patterns order like the integers, denormals included: exp/pkmax3_check.hip,
// profiles/r02_valu_issue.txt).  Every caller's values are path-cost terms far below that.
__device__ __forceinline__ u32 pk_max3(u32 a, u32 b, u32 c)
{
    u32 r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (max(lo, hi), max(lo, hi)): src1 enters with its halves swapped, one instruction
__device__ __forceinline__ u32 pk_max_halves(u32 a)
{
    u32 r;
    asm("v_pk_max_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}

// Reads n16 16-byte pieces once and keeps nothing: run on the census images between the census kernel and the
// first sweep, it leaves them in the memory-side cache, where the sweeps' reads find them next to the G write
// stream (a first sweep that has to fetch them from HBM in between its stores takes 20-40 % longer).
__global__ __launch_bounds__(256) void cache_warm_kernel(const u32x4 *__restrict__ p, size_t n16, u32 *__restrict__ sink)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    u32 acc = 0;
    for (size_t i = tid; i < n16; i += nth) {
        const u32x4 v = p[i];
        acc |= v.x & v.y & v.z & v.w;
    }
    if (acc == 0xFFFFFFFFu && n16 == 0) // never: n16 > 0 in every launch
        sink[tid] = acc;
}

// RP consecutive census words through a buffer descriptor (4-byte aligned)
template <int RP> __device__ __forceinline__ void buf_load_win(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u32 (&o)[RP]);
// (RP = 1, 2, 4 below; the generic form after them serves 8, 12, 16)
template <> __device__ __forceinline__ void buf_load_win<1>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u32 (&o)[1])
{
    o[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
}
template <> __device__ __forceinline__ void buf_load_win<2>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u32 (&o)[2])
{
    const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    o[0] = t.x; o[1] = t.y;
}
template <> __device__ __forceinline__ void buf_load_win<4>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u32 (&o)[4])
{
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <int RP> __device__ __forceinline__ void buf_load_win(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, u32 (&o)[RP])
{
    static_assert(RP % 4 == 0, "window loads come in 16-byte pieces");
#pragma unroll
    for (int m = 0; m < RP / 4; m++) {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * m, soff, 0);
        o[4 * m] = t.x; o[4 * m + 1] = t.y; o[4 * m + 2] = t.z; o[4 * m + 3] = t.w;
    }
}

// RP packed pairs (each half <= 255) -> 2 RP bytes [lo0, hi0, lo1, hi1, ...] in NW = ceil(RP/2) dwords
template <int RP> __device__ __forceinline__ void pack_e(const u32 (&e)[RP], u32 (&v)[(RP + 1) / 2])
{
    if constexpr (RP == 1) {
        v[0] = __builtin_amdgcn_perm(0u, e[0], 0x0c0c0200u);
    } else {
#pragma unroll
        for (int m = 0; m < RP / 2; m++)
            v[m] = __builtin_amdgcn_perm(e[2 * m + 1], e[2 * m], 0x06040200u);
    }
}
template <int RP>
__device__ __forceinline__ void buf_store_ev(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const u32 (&v)[(RP + 1) / 2])
{
    if constexpr (RP == 1) {
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v[0], rs, voff, soff, E_STORE_AUX);
    } else if constexpr (RP == 2) {
        __builtin_amdgcn_raw_buffer_store_b32(v[0], rs, voff, soff, E_STORE_AUX);
    } else if constexpr (RP == 4) {
        u32x2 t;
        t.x = v[0]; t.y = v[1];
        __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff, soff, E_STORE_AUX);
    } else {
        constexpr int NW = RP / 2;
#pragma unroll
        for (int m = 0; m + 4 <= NW; m += 4) {
            u32x4 t;
            t.x = v[m]; t.y = v[m + 1]; t.z = v[m + 2]; t.w = v[m + 3];
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, voff + 4 * m, soff, E_STORE_AUX);
        }
        if constexpr (NW % 4 == 2) {
            u32x2 t;
            t.x = v[NW - 2]; t.y = v[NW - 1];
            __builtin_amdgcn_raw_buffer_store_b64(t, rs, voff + 4 * (NW - 2), soff, E_STORE_AUX);
        }
    }
}
// Hamming costs of a lane: wl[i] = censR[x - RP q - (RP-1) + i] (low half of register RP-1-i),
// wh[j] = censR[x - (Dp-1) + RP q + j] (high half of register j).  With MASK the halves whose
// disparity exceeds x become 255 (cost <= 32, so OR-ing 0xFF works): low half of register j when
// j > tl = x - RP q, high half when j < th = Dp-1 - RP q - x.
template <int RP, bool MASK>
__device__ __forceinline__ void cost_from_census(const u32 (&wl)[RP], const u32 (&wh)[RP], u32 cl, int tl, int th, u32 (&c)[RP])
{
#pragma unroll
    for (int j = 0; j < RP; j++) {
        const u32 lo = (u32)__builtin_popcount(wl[RP - 1 - j] ^ cl);
        const u32 hi = (u32)__builtin_popcount(wh[j] ^ cl);
        c[j] = lo | (hi << 16);
        if (MASK)
            c[j] |= (j > tl ? 0xFFu : 0u) | (j < th ? 0xFF0000u : 0u);
    }
}

// Per-lane path geometry (each 16-lane row walks its own path); see path_geometry.
struct LaneGeom {
    int x0, y0, len;
    bool edge;
};
__device__ __forceinline__ LaneGeom lane_geometry(int p, int w, int h, int dx, int dy)
{
    const int xe = dx > 0 ? 0 : w - 1;
    const int ye = dy > 0 ? 0 : h - 1;
    const bool on_col = (dy == 0) || (dx != 0 && p < h);
    const int q = p - h;
    LaneGeom g;
    g.x0 = on_col ? xe : (dx == 0 ? p : (dx > 0 ? q + 1 : q));
    g.y0 = on_col ? p : ye;
    const int sx = dx == 0 ? 0x7fffffff : (dx > 0 ? w - g.x0 : g.x0 + 1);
    const int sy = dy == 0 ? 0x7fffffff : (dy > 0 ? h - g.y0 : g.y0 + 1);
    g.len = sx < sy ? sx : sy;
    g.edge = (dy == 0 && (g.y0 == 0 || g.y0 == h - 1)) || (dx == 0 && (g.x0 == 0 || g.x0 == w - 1));
    return g;
}

// ---------------------------------------------------------------------------
// K3' one direction sweep (stereo_sgm.cc:191-268 + stereo_sgm.h:119-147), 4 paths per wave.
// cl/cr: census images of the launch group, E: this direction's byte volume, BL: border ring of
// L_{k-1} (in) / L_k (out), SB: border accumulator of the start-pixel carry.
//
// Addressing costs no VALU: all 4 paths of a wave share the per-step byte stride, so each stream
// is  base' + voffset(lane: start of its path + its window) + soffset(step),  with soffset walked by
// one scalar add.  For a negative stride the walk runs downwards from T|s| (T = longest path - 1)
// and base' = base - T|s|, so soffset never goes negative; the descriptors are unbounded.
// ---------------------------------------------------------------------------
// HZ != 0 (horizontal sweeps, dx = HZ = +-1): a path's Dp census words censR[x-Dp+1 .. x] sit in its 16 lanes as
// one long shift register -- the low windows run down the lanes, the high windows up, and the two meet in lane
// 15.  Stepping x by one shifts every window by one word: the word a lane gains is the word its neighbour drops
// (one row_shr and one row_shl DPP move), lane 15 hands its own word from one half to the other, and only lane 0
// takes a word from memory (a 4-byte broadcast load per path and step instead of 2 x 4 RP bytes per lane).  The
// shifts inside a lane are register renaming once the loop is unrolled RP steps (PF is a multiple of RP here).
//
// WRAP (diagonal sweeps, dx != 0 and dy != 0): a 16-lane row does not walk one diagonal but a "wrapped" one, the pixels
// (x, y) = ((c + t dx) mod w, y0 + t dy), t = 0 .. h-1, for c = 0 .. w-1 -- diagonal c and then, from the opposite
// image column on, the diagonal that starts one row further.  That gives w composite paths of the same length h
// instead of w + h - 1 paths of lengths 1 .. min(w, h): as many equally long waves as a vertical sweep has.  Where a
// row wraps it does what the reference does at those pixels: the pixel before the wrap ends a path (its L goes into
// the ring), the pixel after it is a start pixel (L_{k-1} + C from the ring, stereo_sgm.cc:201,236).  Rows of a wave
// wrap at consecutive steps; those few steps ("zone" zs .. ze below, every w steps) run as the per-lane slow step,
// everything else as the steady-state loops.  Address streams keep their common per-step stride; a row that wraps
// moves its voffset by w pixels (the load streams PF steps earlier than the store stream).
template <int RP, bool FIRST, bool HAS_PAD, int PF, int HZ = 0, bool WRAP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RP <= 4 ? 4 : (RP <= 8 ? 3 : 2)))) void aggregate3_kernel(const u32 *__restrict__ cl, const u32 *__restrict__ cr,
                                                         uint8_t *__restrict__ E, uint16_t *__restrict__ BL,
                                                         uint16_t *__restrict__ SB, const Agg2Args a)
{
    constexpr int Dp = 32 * RP, NW = (RP + 1) / 2;
    const int lane = threadIdx.x & 63;
    const int q = lane & 15;
    // XCD-aware block -> (pair, path group) map: workgroups are dealt round-robin over the 8 XCDs, so
    // block b lands on XCD b % 8.  A multiple of 8 pairs: every XCD gets its own pairs (pair % 8 == b % 8),
    // so the census images an XCD's L2 has to hold are those of the few pairs it is working on, and the 8
    // XCDs write neighbouring E volumes.  Otherwise the pair-major list of (pair, group of 16 paths) items is
    // cut into 8 equal contiguous chunks, one per XCD (one 4K pair spreads over the whole chip).
    // Speed only: any placement is correct.
    const int ngroups = (a.npaths + 15) >> 4;
    const int xcd = (int)(blockIdx.x & 7), bi = (int)(blockIdx.x >> 3);
    int pair, grp;
    if ((a.npairs & 7) == 0) {
        pair = (bi / ngroups) * 8 + xcd;
        grp = bi % ngroups;
        if (pair >= a.npairs)
            return;
    } else {
        const int items = ngroups * a.npairs, per = (items + 7) >> 3;
        const int item = xcd * per + bi;
        if (bi >= per || item >= items)
            return;
        pair = item / ngroups;
        grp = item - pair * ngroups;
    }
    const int wave = grp * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave * 4 >= a.npaths)
        return;
    const int w = a.w, h = a.h, wp = a.wp;
    const size_t npix = (size_t)w * h;
    // rows past the last path repeat it (same loads, same E bytes) with their ring stores disabled
    const int praw = wave * 4 + (lane >> 4);
    const bool valid = praw < a.npaths;
    static_assert(!WRAP || HZ == 0, "wrapped paths are diagonal");
    LaneGeom g;
    if constexpr (WRAP) { // composite path c starts in column c of the first image row the sweep visits
        g.x0 = valid ? praw : a.npaths - 1;
        g.y0 = a.dy > 0 ? 0 : h - 1;
        g.len = h;
        g.edge = false;
    } else {
        g = lane_geometry(valid ? praw : a.npaths - 1, w, h, a.dx, a.dy);
    }
    const int len = g.len;
    int minlen = __builtin_amdgcn_readlane(len, 0), maxlen = minlen, minx = __builtin_amdgcn_readlane(g.x0, 0);
#pragma unroll
    for (int r = 1; r < 4; r++) {
        const int l = __builtin_amdgcn_readlane(len, 16 * r), x = __builtin_amdgcn_readlane(g.x0, 16 * r);
        minlen = l < minlen ? l : minlen;
        maxlen = l > maxlen ? l : maxlen;
        minx = x < minx ? x : minx;
    }
    const bool any_edge = __builtin_amdgcn_readfirstlane((int)(__ballot(g.edge) != 0ull)) != 0;
    // A wave with a path along an image edge stores L of every pixel into the ring: ~30 % more work per step
    // than its neighbours on the SIMD, and with every wave of a horizontal sweep resident from the start the
    // launch lasts as long as those few waves (0.52 ms against 0.42 for the other directions).  They go first.
#ifndef SGMX_NO_EDGE_PRIO
    if (any_edge)
        __builtin_amdgcn_s_setprio(3);
#endif

    // per-step byte strides and stream bases
    const int sRi = (a.dy * wp + a.dx) * 4, sLi = (a.dy * w + a.dx) * 4, sEi = (a.dy * w + a.dx) * Dp;
    const unsigned T = (unsigned)(maxlen - 1);
    const unsigned soR0 = sRi < 0 ? T * (unsigned)(-sRi) : 0u, soL0 = sLi < 0 ? T * (unsigned)(-sLi) : 0u,
                   soE0 = sEi < 0 ? T * (unsigned)(-sEi) : 0u;
    const unsigned sR = (unsigned)sRi, sL = (unsigned)sLi, sE = (unsigned)sEi;
    // WRAP, dx > 0: a row that wraps moves its voffsets DOWN by w pixels; they start this many pixels up (and the
    // bases as many down) so that no voffset goes below zero: (w + h - 2) / w wraps at most
    const unsigned wbias = WRAP && a.dx > 0 ? (unsigned)((w + h - 2) / w) * (unsigned)w : 0u;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(cr + (size_t)pair * (size_t)h * wp) - soR0 - (size_t)wbias * 4), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(cl + (size_t)pair * npix) - soL0 - (size_t)wbias * 4), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(E + (size_t)pair * npix * Dp) - soE0 - (size_t)wbias * Dp, 0, -1, 0x00020000);
    // lane parts: start pixel of the lane's path + its window / line slice (WRAP: moved when the row wraps)
    unsigned voRl = (unsigned)((g.y0 * wp + Dp + g.x0 - RP * q - (RP - 1)) * 4) + wbias * 4; // low-half window
    unsigned voRh = (unsigned)((g.y0 * wp + g.x0 + 1 + RP * q) * 4) + wbias * 4;             // high-half window
    unsigned voL = (unsigned)((g.y0 * w + g.x0) * 4) + wbias * 4;
    unsigned voE = (unsigned)((g.y0 * w + g.x0) * Dp + 2 * RP * q) + wbias * Dp;
    // WRAP: what a wrap adds to a voffset (one image row of pixels back or forth), and the row's column at step 0
    // "unwrapped" (column at step t = x0 + t dx): x0e for the step being computed, x0p for the step being prefetched
    const unsigned wrapR = (unsigned)(-a.dx * w * 4), wrapE = (unsigned)(-a.dx * w * Dp);
    int x0e = g.x0, x0p = g.x0;
    const int xend = a.dx > 0 ? w - 1 : 0; // WRAP: the column where a diagonal ends
    const size_t ring = (size_t)(2 * w + 2 * h) * (Dp / 2); // uint32 units
    // Paths along an image edge write L of every pixel into the ring.  Away from the corners (steps
    // 1 .. len-2) the ring slot is linear in the step, so the steady-state loop stores through a
    // descriptor with a walking soffset like the other streams.
    const int sBi = (a.dy == 0 ? a.dx : a.dy) * Dp * 2;
    const unsigned soB0 = sBi < 0 ? T * (unsigned)(-sBi) : 0u, sB = (unsigned)sBi;
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(reinterpret_cast<u32 *>(a.fuse == 2 ? a.BLout : BL) + (size_t)pair * ring) - soB0, 0, -1, 0x00020000);
    const int bi0 = a.dy == 0 ? (g.y0 == 0 ? 0 : w) + g.x0 : 2 * w + (g.x0 == 0 ? 0 : h) + g.y0;
    const unsigned voB = (unsigned)((bi0 * (Dp / 2) + RP * q) * 4);
    const bool edge_store = g.edge && valid;
    u32 *BLf = reinterpret_cast<u32 *>(BL) + (size_t)pair * ring + RP * q;   // ring reads
    u32 *BLw = reinterpret_cast<u32 *>(a.fuse == 2 ? a.BLout : BL) + (size_t)pair * ring + RP * q; // ring writes
    u32 *SBf = reinterpret_cast<u32 *>(SB) + (size_t)pair * ring + RP * q;
    const bool ringw = valid && a.fuse != 1; // this lane stores into the ring at all
    const int xscol = a.dx > 0 ? 0 : w - 1;  // the image column paths of this direction start on

    const u32 P1p = (u32)a.P1 * 0x10001u, P2p = (u32)a.P2 * 0x10001u;
    int tl0 = g.x0 - RP * q;          // x - RP q at step 0 (WRAP: of the unwrapped column)
    int th0 = Dp - 1 - RP * q - g.x0; // Dp-1 - RP q - x at step 0
    u32 padfloor[RP], padmask[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) {
        padfloor[j] = (RP * q + j >= a.D ? INF16 : 0u) | (Dp - 1 - (RP * q + j) >= a.D ? INF16 << 16 : 0u);
        padmask[j] = (RP * q + j >= a.D ? 0u : 0xFFFFu) | (Dp - 1 - (RP * q + j) >= a.D ? 0u : 0xFFFF0000u);
    }
    // The state carried from pixel to pixel is not L but Q = K - L for a constant K (bigger is better; pad and
    // far-away disparities sit at 0).  aggregate_path only looks at differences of the previous pixel's costs,
    // so K may change from step to step; what every producer of Q guarantees is max_d Q >= P2.
    //   unmasked steady state: Q = (P2 - E) + (32 - C)          (K = 32 + P2; 32 - C is a popcount too)
    //   steps with d > x costs / last steps:  Q = KM - L         (KM = 255 + 2 P2 >= every L there, + P2)
    //   start pixels:                         Q = sat(KS - L)    (KS = 0x7000 > every start value; pads -> 0)
    const u32 KMp = (u32)(255 + 2 * a.P2) * 0x10001u, K32p = (u32)(32 + a.P2) * 0x10001u, K255p = (u32)(255 + a.P2) * 0x10001u;
    // Steady-state steps with d > x costs (MASK): Q = G + (255 - C) (K = 255 + P2): halves with d <= x get
    // 223 + popcount(~(censL ^ censR)) = 255 - C on top of G, halves with d > x (C = 255) nothing.  Which halves are
    // alive comes from ONE packed value per lane and step, alv = (tl, -th) as two int16 (tl = x - RP q, th = Dp-1 - RP q - x):
    // register j's low half is alive iff j <= tl, its high half iff j >= th, i.e. iff the halves of
    // (j - 1, -j - 1) - alv are negative -- a packed subtraction from a constant and a packed arithmetic shift.
    // (All of it modulo 2^16: x and Dp stay far below 2^15, check_size.)
    u32 alv0 = ((u32)tl0 & 0xFFFFu) | ((u32)(-th0) << 16);

    // ---- step 0, the start pixels: L_k = L_{k-1} + C; their E bytes are 0, the carry goes to SB
    u32 prior[RP];
    {
        u32 wl[RP], wh[RP], c[RP];
        buf_load_win<RP>(rsR, voRl, soR0, wl);
        buf_load_win<RP>(rsR, voRh, soR0, wh);
        const u32 clv = __builtin_amdgcn_raw_buffer_load_b32(rsL, voL, soL0, 0);
        cost_from_census<RP, true>(wl, wh, clv, tl0, th0, c);
        const size_t bo = (size_t)border_index(g.y0, g.x0, w, h) * (Dp / 2);
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < RP; j++)
                prior[j] = c[j];
        } else {
            u32 bl[RP], sb[RP];
            __builtin_memcpy(bl, BLf + bo, 4 * RP);
            __builtin_memcpy(sb, SBf + bo, 4 * RP);
#pragma unroll
            for (int j = 0; j < RP; j++) {
                prior[j] = pk_add(bl[j], c[j]);
                sb[j] = pk_add(sb[j], bl[j]);
                if (a.fuse == 2) { // L_{k-1} + 2 C; the carries of directions k and k+1: L_{k-1} + (L_{k-1} + C)
                    sb[j] = pk_add(sb[j], prior[j]);
                    prior[j] = pk_add(prior[j], c[j]);
                }
            }
            if (valid && !(a.fuse == 1 && g.x0 == xscol))
                __builtin_memcpy(SBf + bo, sb, 4 * RP);
        }
#pragma unroll
        for (int j = 0; j < RP; j++) {
            if (HAS_PAD)
                prior[j] = pk_max(prior[j], padfloor[j]);
        }
        // the byte volumes hold G = P2 - E (what the steps compute): E = 0 on start pixels
        u32 gstart[NW];
#pragma unroll
        for (int m = 0; m < NW; m++)
            gstart[m] = (u32)a.P2 * 0x01010101u;
        buf_store_ev<RP>(rsE, voE, soE0, gstart);
        if (ringw)
            __builtin_memcpy(BLw + bo, prior, 4 * RP);
#pragma unroll
        for (int j = 0; j < RP; j++)
            prior[j] = pk_sub_sat(0x70007000u, prior[j]); // L -> Q (pads: INF16 -> 0)
    }
    if (maxlen == 1)
        return;

    // ---- steps 1 .. len-1; ring slot i holds a step t with (t-1) % PF == i
    static_assert(HZ == 0 || PF % RP == 0, "the window shifts become renaming only over a multiple of RP steps");
    u32 qwl[HZ ? 1 : PF][RP], qwh[HZ ? 1 : PF][RP], qcl[PF];
    u32 W[RP], WH[RP], qnw[PF]; // HZ: the windows of the next step; slot i: the word that enters after its step
    // HZ: the word that enters lane 0 when the path moves on to pixel x: censR[x] (low half, dx > 0) or
    // censR[x - (Dp-1)] (high half, dx < 0); every lane of the row loads it (one address per row)
    const unsigned voN = (unsigned)((g.y0 * wp + (HZ > 0 ? Dp + g.x0 : g.x0 + 1)) * 4);
    if constexpr (HZ != 0) {
        buf_load_win<RP>(rsR, voRl, soR0 + sR, W);
        buf_load_win<RP>(rsR, voRh, soR0 + sR, WH);
#pragma unroll
        for (int i = 0; i < PF; i++) {
            if (1 + i < len)
                qcl[i] = __builtin_amdgcn_raw_buffer_load_b32(rsL, voL, soL0 + (1 + i) * sL, 0);
            if (2 + i < len)
                qnw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsR, voN, soR0 + (2 + i) * sR, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            if constexpr (WRAP) { // a row whose step 1 + i lies past the image column it ends in continues w pixels back
                const int xp = x0p + (1 + i) * a.dx;
                if (a.dx > 0 ? xp >= w : xp < 0) {
                    x0p -= a.dx * w;
                    voRl += wrapR;
                    voRh += wrapR;
                    voL += wrapR;
                }
            }
            if (1 + i < len) {
                buf_load_win<RP>(rsR, voRl, soR0 + (1 + i) * sR, qwl[i]);
                buf_load_win<RP>(rsR, voRh, soR0 + (1 + i) * sR, qwh[i]);
                qcl[i] = __builtin_amdgcn_raw_buffer_load_b32(rsL, voL, soL0 + (1 + i) * sL, 0);
            }
        }
    }
    u32 *bl_last = nullptr; // the ring slot of the path's last pixel
    if constexpr (!WRAP)
        bl_last = BLw + (size_t)border_index(g.y0 + (len - 1) * a.dy, g.x0 + (len - 1) * a.dx, w, h) * (Dp / 2);

    unsigned soRp = soR0 + (1 + PF) * sR, soLp = soL0 + (1 + PF) * sL; // soffsets of the step PF ahead
    unsigned soE = soE0, soB = soB0;
    // guardc: a steady-state (not SLOW) step near the end of the path, whose prefetch target may lie past it
    auto step = [&](auto maskc, auto slowc, auto edgec, auto guardc, int t, int i) {
        constexpr bool SLOW = decltype(slowc)::value, EDGE = decltype(edgec)::value, GUARD = decltype(guardc)::value;
        soE += sE;
        soB += sB;
        // WRAP, slow steps: does this row start a new diagonal at step t (the column after the one diagonals end in),
        // and is its pixel the last of a diagonal (its L goes into the ring)?  Both are border pixels.
        bool wstart = false, wend = false;
        int xt = 0;
        if constexpr (WRAP && SLOW) {
            xt = x0e + t * a.dx;
            if (a.dx > 0 ? xt >= w : xt < 0) {
                x0e -= a.dx * w;
                xt -= a.dx * w;
                tl0 -= a.dx * w;
                th0 += a.dx * w;
                alv0 = ((u32)tl0 & 0xFFFFu) | ((u32)(-th0) << 16);
                voE += wrapE;
                wstart = true;
            }
            wend = xt == xend || t == len - 1;
        }
        u32 wl[RP], wh[RP];
#pragma unroll
        for (int j = 0; j < RP; j++) {
            wl[j] = HZ ? W[j] : qwl[i][j];
            wh[j] = HZ ? WH[j] : qwh[i][j];
        }
        const u32 clv = qcl[i];
        if constexpr (HZ != 0) { // move the windows on to the next pixel (the step works on its copies wl / wh)
            const u32 nw = qnw[i];
            if constexpr (HZ > 0) {
                const u32 inl = dpp<0x111>(nw, W[0]);   // row_shr:1 -- lane q-1's oldest low word; lane 0: the new word
                const u32 inh = dpp<0x101>(W[0], WH[0]); // row_shl:1 -- lane q+1's oldest high word; lane 15: its own low one
#pragma unroll
                for (int j = 0; j + 1 < RP; j++) {
                    W[j] = W[j + 1];
                    WH[j] = WH[j + 1];
                }
                W[RP - 1] = inl;
                WH[RP - 1] = inh;
            } else {
                const u32 inl = dpp<0x101>(WH[RP - 1], W[RP - 1]);  // row_shl:1 -- lane 15: its own top high word
                const u32 inh = dpp<0x111>(nw, WH[RP - 1]);         // row_shr:1 -- lane 0: the new word
#pragma unroll
                for (int j = RP - 1; j > 0; j--) {
                    W[j] = W[j - 1];
                    WH[j] = WH[j - 1];
                }
                W[0] = inl;
                WH[0] = inh;
            }
        }
        auto prefetch = [&]() { // census words PF steps ahead, into the ring slot this step has just consumed
            if constexpr (HZ != 0) {
                if ((!SLOW && !GUARD) || t + PF < len)
                    qcl[i] = __builtin_amdgcn_raw_buffer_load_b32(rsL, voL, soLp, 0);
                if ((!SLOW && !GUARD) || t + PF + 1 < len) // (fast blocks end two blocks and a step before the path does)
                    qnw[i] = __builtin_amdgcn_raw_buffer_load_b32(rsR, voN, soRp + sR, 0);
            } else {
            if constexpr (WRAP && (SLOW || GUARD)) { // (fast blocks end before any row's prefetch target wraps)
                const int xp = x0p + (t + PF) * a.dx;
                if (a.dx > 0 ? xp >= w : xp < 0) {
                    x0p -= a.dx * w;
                    voRl += wrapR;
                    voRh += wrapR;
                    voL += wrapR;
                }
            }
#ifdef SGMX_EXP_NOLOAD // timing-only build (wrong results): the steady-state loops without their census loads
            if (SLOW && t + PF < len) {
#else
            if ((!SLOW && !GUARD) || t + PF < len) { // (fast blocks: in range for every row)
#endif
                buf_load_win<RP>(rsR, voRl, soRp, qwl[i]);
                buf_load_win<RP>(rsR, voRh, soRp, qwh[i]);
                qcl[i] = __builtin_amdgcn_raw_buffer_load_b32(rsL, voL, soLp, 0);
            }
            }
#ifndef SGMX_EXP_FIXLOAD // (defined: timing-only, every step loads the census words of the path's first pixels again)
            soRp += sR;
            soLp += sL;
#endif
        };
        if (SLOW)
            prefetch();
        constexpr bool MASK = decltype(maskc)::value;
        // stereo_sgm.h:129-146 in the mirrored layout (see the head of this file), on Q = K - L:
        //   G = P2 - E = max(P2 - (L[d] - m), P2 - P1 - (L[d+-1] - m), 0)
        //     = max(g[d], h[d-1], h[d+1])  with  g = sat(Q - (maxQ - P2)),  h = sat(g - P1)
        // (saturating subtractions do the clamping at 0 = "P2 or more above the minimum").
        u32 mx = prior[0];
#pragma unroll
        for (int j = 1; j + 1 < RP; j += 2)
            mx = pk_max3(mx, prior[j], prior[j + 1]);
        if constexpr (RP % 2 == 0)
            mx = pk_max(mx, prior[RP - 1]);
        const u32 mq = row_max_u32(pk_max_halves(mx));      // (max Q, max Q) in every lane of the row
        const u32 tq = pk_sub(mq, P2p);                      // >= 0: every producer of Q keeps max Q >= P2
        u32 gq[RP], hh[RP];
#pragma unroll
        for (int j = 0; j < RP; j++) {
            gq[j] = pk_sub_sat(prior[j], tq);
            hh[j] = pk_sub_sat(gq[j], P1p);
        }
        // "register -1": lane q-1's top register; lane 0 has no neighbour on either half -> 0 (bound_ctrl)
        const u32 hbelow = (u32)__builtin_amdgcn_update_dpp(0, (int)hh[RP - 1], 0x111, 0xF, 0xF, true);   // row_shr:1
        // "register RP": lane q+1's register 0; lane 15 keeps its own top register with the halves swapped
        const u32 habove = dpp<0x101>(__builtin_amdgcn_alignbit(hh[RP - 1], hh[RP - 1], 16), hh[0]);      // row_shl:1
        u32 e[RP]; // G = P2 - E
#pragma unroll
        for (int j = 0; j < RP; j++)
            e[j] = pk_max3(gq[j], j ? hh[j - 1] : hbelow, j < RP - 1 ? hh[j + 1] : habove);
        u32 Lv[RP] = {}; // the true path costs, where a ring store may need them
        if constexpr (MASK && !SLOW) {
            u32 ncl;
            asm("v_not_b32 %0, %1" : "=v"(ncl) : "v"(clv));
            const u32 alv = pk_add(alv0, ((u32)(t * a.dx) & 0xFFFFu) * 0x10001u);
            const u32 c223 = 223u;
#pragma unroll
            for (int j = 0; j < RP; j++) {
                u32 lo, hi, pc;
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(lo) : "v"(wl[RP - 1 - j] ^ ncl), "s"(c223));
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(hi) : "v"(wh[j] ^ ncl), "s"(c223));
                asm("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(pc) : "v"(hi), "v"(lo));
                const u32 cj = ((u32)(j - 1) & 0xFFFFu) | ((u32)(-j - 1) << 16);
                const u32 alive = __builtin_bit_cast(u32, (ss2)(__builtin_bit_cast(ss2, pk_sub(cj, alv)) >> 15));
                prior[j] = pk_add(e[j], pc & alive);
                if (HAS_PAD)
                    prior[j] &= padmask[j];
                if (EDGE)
                    Lv[j] = pk_sub(K255p, prior[j]);
            }
        } else if (MASK) {
            u32 c[RP];
            cost_from_census<RP, true>(wl, wh, clv, tl0 + t * a.dx, th0 - t * a.dx, c);
            u32 bl[RP] = {};
            if constexpr (WRAP && SLOW && !FIRST) {
                if (wstart) { // start pixel: L_k = L_{k-1} + C, the carry L_{k-1} goes to SB (as in step 0)
                    const size_t bo = (size_t)border_index(g.y0 + t * a.dy, xt, w, h) * (Dp / 2);
                    u32 sb[RP];
                    __builtin_memcpy(bl, BLf + bo, 4 * RP);
                    __builtin_memcpy(sb, SBf + bo, 4 * RP);
#pragma unroll
                    for (int j = 0; j < RP; j++)
                        sb[j] = pk_add(sb[j], bl[j]);
                    if (valid && a.fuse != 1) // (fuse = 1: a column start, its carry is direction k+1's to add)
                        __builtin_memcpy(SBf + bo, sb, 4 * RP);
                }
            }
#pragma unroll
            for (int j = 0; j < RP; j++) {
                Lv[j] = pk_add(pk_sub(P2p, e[j]), c[j]);
                if (WRAP && SLOW && wstart) {
                    Lv[j] = pk_add(bl[j], c[j]);
                    e[j] = P2p; // E = 0 on start pixels
                }
                if (HAS_PAD)
                    Lv[j] = pk_max(Lv[j], padfloor[j]);
                prior[j] = pk_sub_sat(WRAP && SLOW && wstart ? 0x70007000u : KMp, Lv[j]);
            }
        } else {
            // Q = G + (32 - C): 32 - popcount(x) = popcount(~x), the add folded into the popcount (v_bcnt
            // accumulates) and into the merge of the halves
            // (through asm: left to itself the compiler fuses the complement into eight v_xnor_b32, which issue at
            //  half the rate of v_xor_b32 -- profiles/r02_valu_issue.txt)
            u32 ncl;
            asm("v_not_b32 %0, %1" : "=v"(ncl) : "v"(clv));
#pragma unroll
            for (int j = 0; j < RP; j++) {
                u32 lo, hi;
#ifdef SGMX_EXP_NOLOAD // (timing-only: the census words are loop invariant there; volatile keeps the popcounts in the loop)
                asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(lo) : "v"(wl[RP - 1 - j] ^ ncl), "v"(e[j]));
                asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(hi) : "v"(wh[j] ^ ncl));
#else
                asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(lo) : "v"(wl[RP - 1 - j] ^ ncl), "v"(e[j]));
                asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(hi) : "v"(wh[j] ^ ncl));
#endif
                asm("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(prior[j]) : "v"(hi), "v"(lo));
                if (HAS_PAD)
                    prior[j] &= padmask[j];
                if (EDGE)
                    Lv[j] = pk_sub(K32p, prior[j]);
            }
        }
        if (!SLOW) {
            // after the slot's last use, and not hoisted above it: otherwise the compiler copies the slot
            __builtin_amdgcn_sched_barrier(0);
            prefetch();
        }
        u32 eb[NW];
        pack_e<RP>(e, eb);
        if (!SLOW) {
#ifndef SGMX_EXP_NOSTORE // (defined: timing-only build, wrong results, the steady-state loops without their E stores)
            buf_store_ev<RP>(rsE, voE, soE, eb);
#else
            if (a.npairs < 0) // keeps eb alive
                buf_store_ev<RP>(rsE, voE, soE, eb);
#endif
            if (EDGE && edge_store) {
                if constexpr (RP == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(Lv[0], rsB, voB, soB, 0);
                } else if constexpr (RP == 2) {
                    u32x2 v;
                    v.x = Lv[0]; v.y = Lv[1];
                    __builtin_amdgcn_raw_buffer_store_b64(v, rsB, voB, soB, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < RP / 4; m++) {
                        u32x4 v;
                        v.x = Lv[4 * m]; v.y = Lv[4 * m + 1]; v.z = Lv[4 * m + 2]; v.w = Lv[4 * m + 3];
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsB, voB + 16 * m, soB, 0);
                    }
                }
            }
        } else if (t < len) {
            buf_store_ev<RP>(rsE, voE, soE, eb);
            if (ringw) {
                if constexpr (WRAP) {
                    if (wstart || wend) // first or last pixel of a diagonal
                        __builtin_memcpy(BLw + (size_t)border_index(g.y0 + t * a.dy, xt, w, h) * (Dp / 2), Lv, 4 * RP);
                } else if (g.edge) // path along an image edge: every pixel feeds a later direction's start
                    __builtin_memcpy(BLw + (size_t)border_index(g.y0 + t * a.dy, g.x0 + t * a.dx, w, h) * (Dp / 2), Lv, 4 * RP);
                else if (t == len - 1)
                    __builtin_memcpy(bl_last, Lv, 4 * RP);
            }
        }
    };

    // every d <= x for all four rows in steps t .. t+PF-1 (otherwise the costs need the 255 mask)
    auto unmasked = [&](int t) {
        const int xa = minx + t * a.dx, xb = minx + (t + PF - 1) * a.dx;
        return (xa < xb ? xa : xb) >= a.D - 1;
    };
    // A run of full blocks of one kind is a loop of its own: sharing one loop between the kinds made the
    // compiler copy the whole prefetch ring and drain vmcnt in every block.
    int t = 1;
    // WRAP: zs .. ze = the steps at which the rows of this wave wrap next (the row with the column nearest the end
    // column first); the step before each is the last pixel of a diagonal.  lim = the step the steady-state loops
    // stop short of: a path end, or the next wrap.
    int zs = 0x7fffffff, ze = 0x7fffffff, lim = minlen;
    if constexpr (WRAP) {
        const int c0 = __builtin_amdgcn_readlane(g.x0, 0), c3 = __builtin_amdgcn_readlane(g.x0, 48);
        zs = a.dx > 0 ? w - c3 : c0 + 1;
        ze = a.dx > 0 ? w - c0 : c3 + 1;
        lim = zs < len ? zs : len;
    }
    auto run = [&](auto maskc, auto edgec) {
        do {
#pragma unroll
            for (int i = 0; i < PF; i++)
                step(maskc, BoolC<false>(), edgec, BoolC<false>(), t + i, i);
            t += PF;
        } while (t + 2 * PF + (HZ ? 1 : 0) <= lim && unmasked(t) != decltype(maskc)::value);
    };
    while (t < maxlen) {
        if constexpr (WRAP) {
            while (ze < t) { // every row has wrapped: on to the next zone, w steps later, with all columns w further back
                zs += w;
                ze += w;
                minx -= a.dx * w;
            }
            lim = zs < len ? zs : len;
        }
        if (t + 2 * PF + (HZ ? 1 : 0) <= lim) { // every row: steps t..t+PF-1 and their prefetch targets exist, none is last
            if (!unmasked(t))
                run(BoolC<true>(), BoolC<true>());
            else if (any_edge)
                run(BoolC<false>(), BoolC<true>());
            else
                run(BoolC<false>(), BoolC<false>());
        } else if (WRAP && t + PF < lim) {
            // blocks before a wrap or the end: no step of theirs is a diagonal's first or last one, but what they
            // prefetch may lie past the wrap (the row's load voffsets move there) or past the end
            if (unmasked(t)) {
#pragma unroll
                for (int i = 0; i < PF; i++)
                    step(BoolC<false>(), BoolC<false>(), BoolC<false>(), BoolC<true>(), t + i, i);
            } else {
#pragma unroll
                for (int i = 0; i < PF; i++)
                    step(BoolC<true>(), BoolC<false>(), BoolC<false>(), BoolC<true>(), t + i, i);
            }
            t += PF;
        } else if (!WRAP && t + PF < minlen && !any_edge && unmasked(t)) {
            // the last full blocks of the paths: none of their steps is a row's last one, but what they prefetch
            // may lie past a path's end -- the steady-state steps with guarded loads
#pragma unroll
            for (int i = 0; i < PF; i++)
                step(BoolC<false>(), BoolC<false>(), BoolC<false>(), BoolC<true>(), t + i, i);
            t += PF;
        } else {
#pragma unroll
            for (int i = 0; i < PF; i++)
                if (t + i < maxlen)
                    step(BoolC<true>(), BoolC<true>(), BoolC<true>(), BoolC<false>(), t + i, i);
            t += PF;
        }
    }
}

// ---------------------------------------------------------------------------
// S(p,.) of one pixel in the lane layout above (RP packed pairs per lane, 16 lanes per pixel),
// rebuilt from the census images, the N byte volumes and the border accumulator.
// ---------------------------------------------------------------------------
struct SumArgs {
    const u32 *cl, *cr;
    const uint8_t *E;
    size_t estride; // bytes between the volumes of two directions
    const uint16_t *SB;
    int w, h, D, wp, N;
    int P2; // the byte volumes hold G = P2 - E
};

// Raw inputs of one pixel's S for one lane: loaded one iteration ahead of their use.
template <int RP> struct PixelRaw {
    u32 clv, wl[RP], wh[RP];
    u32 ev[8][(RP + 1) / 2]; // RP = 1: the two bytes sit in the low half of ev[k][0]
};

template <int RP>
__device__ __forceinline__ void load_pixel(const SumArgs &a, int pair, int y, int x, int q, PixelRaw<RP> &r)
{
    constexpr int Dp = 32 * RP;
    const size_t pix = ((size_t)pair * a.h + y) * a.w + x;
    r.clv = a.cl[pix];
    const u32 *row = a.cr + ((size_t)pair * a.h + y) * a.wp;
    __builtin_memcpy(r.wl, row + (Dp + x - RP * q - (RP - 1)), 4 * RP);
    __builtin_memcpy(r.wh, row + (x + 1 + RP * q), 4 * RP);
    const uint8_t *ep = a.E + pix * Dp + 2 * RP * q;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < a.N) {
            const uint8_t *pk = ep + (size_t)k * a.estride;
            // read once: non-temporal, like the stores that wrote them
            if constexpr (RP == 1) {
                r.ev[k][0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(pk));
            } else if constexpr (RP == 2) {
                r.ev[k][0] = __builtin_nontemporal_load(reinterpret_cast<const u32 *>(pk));
            } else if constexpr (RP == 4) {
                const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk));
                r.ev[k][0] = v.x;
                r.ev[k][1] = v.y;
            } else { // 8-byte pieces: the lane's 2 RP bytes start at a multiple of 8, not always of 16
#pragma unroll
                for (int m = 0; m < RP / 2; m += 2) {
                    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + 4 * m));
                    r.ev[k][m] = v.x;
                    r.ev[k][m + 1] = v.y;
                }
            }
        }
}

// S = N C + sum_k E_k = N (C + P2) - sum_k G_k: the byte volumes hold G = P2 - E.  The bytes of G volumes are
// added as packed bytes first (no carry while G * P2 <= 255), then widened to the packed uint16 pairs.
template <int RP, int G, bool MASK>
__device__ __forceinline__ void sum_pixel(const SumArgs &a, const PixelRaw<RP> &r, int x, int q, u32 (&S)[RP])
{
    constexpr int NW = (RP + 1) / 2;
    u32 c[RP];
    cost_from_census<RP, MASK>(r.wl, r.wh, r.clv, x - RP * q, 32 * RP - 1 - RP * q - x, c);
    const int sh = a.N == 8 ? 3 : 2;
    const u32 np2 = (u32)(a.N * a.P2) * 0x10001u;
#pragma unroll
    for (int j = 0; j < RP; j++)
        S[j] = pk_add(c[j] << sh, np2); // N * (C + P2)
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += G)
        if (k0 < a.N) {
            u32 acc[NW];
#pragma unroll
            for (int m = 0; m < NW; m++)
                acc[m] = r.ev[k0][m];
#pragma unroll
            for (int k = k0 + 1; k < k0 + G; k++)
                if (k < a.N) // (4 paths with a group of 8: volumes 4..7 do not exist, load_pixel left their registers alone)
#pragma unroll
                    for (int m = 0; m < NW; m++)
                        acc[m] += r.ev[k][m];
#pragma unroll
            for (int j = 0; j < RP; j++)
                S[j] = pk_sub(S[j], __builtin_amdgcn_perm(0u, acc[j >> 1], (j & 1) ? 0x0c030c02u : 0x0c010c00u));
        }
}

template <int RP>
__device__ __forceinline__ void add_border(const SumArgs &a, int pair, int y, int x, int q, u32 (&S)[RP])
{
    constexpr int Dp = 32 * RP;
    if (y == 0 || y == a.h - 1 || x == 0 || x == a.w - 1) {
        u32 sb[RP];
        __builtin_memcpy(sb, reinterpret_cast<const u32 *>(a.SB) +
                                 ((size_t)pair * (2 * a.w + 2 * a.h) + border_index(y, x, a.w, a.h)) * (Dp / 2) + RP * q,
                         4 * RP);
#pragma unroll
        for (int j = 0; j < RP; j++)
            S[j] = pk_add(S[j], sb[j]);
    }
}

template <int RP>
__device__ __forceinline__ void pixel_sums(const SumArgs &a, int pair, int y, int x, int q, u32 (&S)[RP])
{
    PixelRaw<RP> r;
    load_pixel<RP>(a, pair, y, x, q, r);
    sum_pixel<RP, 1, true>(a, r, x, q, S);
    add_border<RP>(a, pair, y, x, q, S);
}

// ---------------------------------------------------------------------------
// K4+K5' WTA, right-view argmin, LR check, subpixel (stereo_sgm.h:209-241, 275-317) on the
// rebuilt S.  One workgroup per image row; a wave handles 4 pixels at a time (one per DPP row),
// with the next 4 pixels' loads in flight.  Right view: LDS atomicMin of (cost << 16 | d) keys
// into slot x-d, i.e. the reference's diagonal scan S[y][xr+i][i] regrouped by xr; the smaller d
// wins ties like its strict '<' scan.
// ---------------------------------------------------------------------------
template <int RP, bool HAS_PAD>
__global__ __launch_bounds__(256) void wta3_kernel(const SumArgs a, int P2, int m, uint16_t *__restrict__ raw)
{
    constexpr int Dp = 32 * RP, HALF = 16 * RP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = a.w, h = a.h, D = a.D;
    u32 *tile = reinterpret_cast<u32 *>(smem);       // [16 pixels in flight][Dp/2] packed S
    u32 *rkey = tile + 16 * (Dp / 2);                // [w] right-view best key
    u32 *lkey = rkey + w;                            // [w] left-view best key
    u32 *s3a = lkey + w;                             // [w] S[d-1] | S[d] << 16 at the left winner
    u32 *s3b = s3a + w;                              // [w] S[d+1]
    const int y = blockIdx.x, pair = blockIdx.y;
    const int W = w + 2 * m;
    uint16_t *out = raw + ((size_t)pair * (h + 2 * m) + (y + m)) * (size_t)W + m;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane >> 4, q = lane & 15;
    for (int x = threadIdx.x; x < w; x += blockDim.x)
        rkey[x] = 0xFFFFFFFFu;
    __syncthreads();
    u32 *mytile = tile + (wave * 4 + r) * (Dp / 2);
    const uint16_t *t16 = reinterpret_cast<const uint16_t *>(mytile);
    const int G = P2 <= 31 ? 8 : (P2 <= 63 ? 4 : (P2 <= 127 ? 2 : 1)); // volumes whose bytes add without carry
    const bool border_row = y == 0 || y == h - 1;

    auto clampx = [&](int x0) { const int x = x0 + r; return x < w ? x : w - 1; };
    // the next 4 pixels' loads are in flight while these are summed, except for the widest layouts
    // (RP >= 12, D > 256), where a second PixelRaw does not fit the register file
    constexpr bool AHEAD = RP <= 8;
    PixelRaw<RP> cur;
    if (AHEAD && wave * 4 < w)
        load_pixel<RP>(a, pair, y, clampx(wave * 4), q, cur);
    for (int x0 = wave * 4; x0 < w; x0 += 16) {
        const int x = x0 + r;
        PixelRaw<AHEAD ? RP : 1> nxt;
        const bool more = x0 + 16 < w;
        if constexpr (AHEAD) {
            if (more)
                load_pixel<RP>(a, pair, y, clampx(x0 + 16), q, nxt);
        } else {
            load_pixel<RP>(a, pair, y, clampx(x0), q, cur);
        }
        const bool full = x0 >= D - 1 && x0 + 3 < w; // every lane: pixel inside the row, every d <= x
        const int xc = x < w ? x : w - 1;
        const int xb = xc - RP * q;                  // x - d_lo of register 0
        u32 S[RP];
        if (full) {
            if (G == 8)
                sum_pixel<RP, 8, false>(a, cur, xc, q, S);
            else if (G == 4)
                sum_pixel<RP, 4, false>(a, cur, xc, q, S);
            else if (G == 2)
                sum_pixel<RP, 2, false>(a, cur, xc, q, S);
            else
                sum_pixel<RP, 1, false>(a, cur, xc, q, S);
        } else {
            sum_pixel<RP, 1, true>(a, cur, xc, q, S);
        }
        if (border_row || x0 == 0 || x0 + 3 >= w - 1)
            add_border<RP>(a, pair, y, xc, q, S);
        u32 best = 0xFFFFFFFFu;
        // right-view slots x - d of this lane: low half of register j at rl[-j], high half at rh[j]
        // (both may lie below rkey when !full; those are never touched)
        u32 *rl = rkey + xb, *rh = rkey + (xc - (Dp - 1) + RP * q);
        if (full) {
#pragma unroll
            for (int j = 0; j < RP; j++) {
                const int dl = RP * q + j, dh = Dp - 1 - dl;
                const u32 kl = (S[j] << 16) | (u32)dl, kh = (S[j] & 0xFFFF0000u) | (u32)dh;
                if (!HAS_PAD || dl < D) {
                    best = min(best, kl);
                    atomicMin(rl - j, kl);
                }
                if (!HAS_PAD || dh < D) {
                    best = min(best, kh);
                    atomicMin(rh + j, kh);
                }
            }
        } else {
            const bool active = x < w;
#pragma unroll
            for (int j = 0; j < RP; j++) {
                const int dl = RP * q + j, dh = Dp - 1 - dl;
                const u32 kl = (S[j] << 16) | (u32)dl, kh = (S[j] & 0xFFFF0000u) | (u32)dh;
                if (active && dl < D) {
                    best = min(best, kl);
                    if (dl <= x)
                        atomicMin(rl - j, kl);
                }
                if (active && dh < D) {
                    best = min(best, kh);
                    if (dh <= x)
                        atomicMin(rh + j, kh);
                }
            }
        }
        __builtin_memcpy(mytile + RP * q, S, 4 * RP);
        best = row_min_u32(best);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (x < w && q == 0) {
            lkey[x] = best;
            const int d = (int)(best & 0xFFFFu);
            if (d >= 1 && d < D - 1) {
                // element d of the tile: low half of word d, or high half of word Dp-1 - d
                auto at = [&](int dd) { return (u32)t16[dd < HALF ? 2 * dd : 2 * (Dp - 1 - dd) + 1]; };
                s3a[x] = at(d - 1) | (at(d) << 16);
                s3b[x] = at(d + 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); // the tile is overwritten by the next iteration
        if constexpr (AHEAD) {
            if (more)
                cur = nxt;
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += blockDim.x) {
        const int d = (int)(lkey[x] & 0xFFFFu);
        uint16_t o = 0;
        if (x - d >= 0) {
            const int dr = (int)(rkey[x - d] & 0xFFFFu);
            if (abs(d - dr) <= 2) {
                float cl = 0.f, cc = 0.f, cr = 0.f;
                if (d >= 1 && d < D - 1) {
                    const u32 ab = s3a[x];
                    cl = (float)(ab & 0xFFFFu);
                    cc = (float)(ab >> 16);
                    cr = (float)s3b[x];
                }
                o = subpixel_u16(d, D, cl, cc, cr, nullptr);
            }
        }
        out[x] = o;
    }
}

// S tap: the rebuilt summed path costs of one pair, dense [h][w][D] uint16.
template <int RP>
__global__ __launch_bounds__(256) void sums_tap_kernel(const SumArgs a, int pair, uint16_t *__restrict__ out)
{
    constexpr int Dp = 32 * RP;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pixi = t / 16;
    const int q = (int)(t % 16);
    if (pixi >= (size_t)a.w * a.h)
        return;
    const int y = (int)(pixi / a.w), x = (int)(pixi % a.w);
    u32 S[RP];
    pixel_sums<RP>(a, pair, y, x, q, S);
#pragma unroll
    for (int j = 0; j < RP; j++) {
        const int dl = RP * q + j, dh = Dp - 1 - dl;
        if (dl < a.D)
            out[pixi * a.D + dl] = (uint16_t)(S[j] & 0xFFFFu);
        if (dh < a.D)
            out[pixi * a.D + dh] = (uint16_t)(S[j] >> 16);
    }
}

} // namespace sgmx
