// sgmx_kernels_v2.hpp -- the census/Hamming fast path (P2 <= 255): no data-cost volume and
// no summed-cost volume in HBM.
//
//   * The Hamming data cost C(p,d) = popcount(censL[p] ^ censR[p-d]) (stereo_sgm.cc:75-101,
//     stereo_costs.h:120-132) is recomputed inside the sweeps from the two census images, which
//     stay cache resident (2 x 1.8 MB per pair), instead of being streamed from a 59 MB volume.
//   * aggregate_path (stereo_sgm.h:119-147) gives L = C + E with
//         E = min(prior[d], prior[d+-1] + P1, m + P2) - m,   0 <= E <= P2,
//     so with P2 <= 255 every direction stores only E as ONE byte per voxel into its own volume
//     (write-only, 128-byte lines) instead of read-modify-writing a uint16 sum (sum_costs,
//     stereo_sgm.h:105-117).  On start pixels L = L_{k-1} + C (stereo_sgm.cc:201,218,236,254) can be
//     larger: they are image-border pixels, their E byte is 0 and L_{k-1} goes into the small
//     border accumulator SB (uint16 ring).  The final kernel rebuilds
//         S(p,d) = N C(p,d) + sum_k E_k(p,d) (+ SB on border pixels)
//     exactly (integers), and does WTA / right-view argmin / LR check / subpixel on it.
//   HBM bytes per voxel: N (E writes) + N (E reads) = 16 at N = 8, against 41 for the materialised
//   C + S formulation that the algorithmic-byte model counts.
//
// Census layout for this path: left uint32 [h][w]; right uint32 [h][wp], wp = Dp + w, pixel x in
// column Dp + x, so that the 2R values censR[x-d] of a lane are one contiguous, in-row load for
// every x >= 0 (columns < Dp are padding; costs with d > x are forced to 255, stereo_sgm.cc:43).
#pragma once
#include "sgmx_kernels.hpp"

namespace sgmx {

struct Agg2Args {
    int w, h, D, wp;   // cropped size, real disparity count, census-right row pitch
    int dx, dy;        // predecessor of (y,x) is (y-dy, x-dx)
    int P1, P2;
    int npaths;
};

// 2R census words of one lane: v[i] = censR[x - d0 - (2R-1) + i], i.e. disparity d0 + 2R-1 - i.
template <int R> struct CensRaw { u32 v[2 * R]; };

template <int R> __device__ __forceinline__ CensRaw<R> load_cens(const u32 *p)
{
    CensRaw<R> r;
    __builtin_memcpy(r.v, p, 8 * R); // 4-byte aligned multi-dword load
    return r;
}

template <int R, bool MASK>
__device__ __forceinline__ void cost_from_census(const CensRaw<R> &raw, u32 cl, int x, int d0, u32 (&c)[R])
{
#pragma unroll
    for (int j = 0; j < R; j++) {
        const u32 lo = (u32)__builtin_popcount(raw.v[2 * R - 1 - 2 * j] ^ cl);
        const u32 hi = (u32)__builtin_popcount(raw.v[2 * R - 2 - 2 * j] ^ cl);
        c[j] = lo | (hi << 16);
        if (MASK) { // d > x: no such pixel in the right view -> 255 (cost <= 32, so OR works)
            const int d = d0 + 2 * j;
            c[j] |= (d > x ? 0xFFu : 0u) | (d + 1 > x ? 0xFF0000u : 0u);
        }
    }
}

// R packed uint16 pairs (each value <= 255) -> 2R bytes
template <int R> __device__ __forceinline__ void store_e(uint8_t *p, const u32 (&e)[R]);
template <> __device__ __forceinline__ void store_e<1>(uint8_t *p, const u32 (&e)[1])
{
    *reinterpret_cast<unsigned short *>(p) = (unsigned short)__builtin_amdgcn_perm(0u, e[0], 0x0c0c0200u);
}
template <> __device__ __forceinline__ void store_e<2>(uint8_t *p, const u32 (&e)[2])
{
    *reinterpret_cast<u32 *>(p) = __builtin_amdgcn_perm(e[1], e[0], 0x06040200u);
}
template <> __device__ __forceinline__ void store_e<4>(uint8_t *p, const u32 (&e)[4])
{
    *reinterpret_cast<uint2 *>(p) =
        make_uint2(__builtin_amdgcn_perm(e[1], e[0], 0x06040200u), __builtin_amdgcn_perm(e[3], e[2], 0x06040200u));
}

// ---------------------------------------------------------------------------
// K3' one direction sweep.  One wave per path, disparities across lanes (2R per lane),
// stereo_sgm.cc:191-268 + stereo_sgm.h:119-147.  cl/cr: census images of the launch group,
// E: this direction's byte volume [pairs][h][w][Dp], BL: border ring of L_{k-1} (in) / L_k (out),
// SB: border accumulator of the start-pixel carry.
// ---------------------------------------------------------------------------
template <int R, bool FIRST, bool HAS_PAD, int PF>
__global__ __launch_bounds__(256) void aggregate2_kernel(const u32 *__restrict__ cl, const u32 *__restrict__ cr,
                                                         uint8_t *__restrict__ E, uint16_t *__restrict__ BL,
                                                         uint16_t *__restrict__ SB, const Agg2Args a)
{
    constexpr int Dp = 128 * R;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (p >= a.npaths)
        return;
    const int w = a.w, h = a.h, wp = a.wp;
    const size_t npix = (size_t)w * h;
    const int d0 = lane * (2 * R);           // this lane's first disparity
    const unsigned loff = lane * (2 * R);    // byte offset into a pixel's E line / element offset into a ring line
    const unsigned coff = Dp - d0 - 2 * R;   // element offset of this lane's census window relative to column x+1
    const u32 *clf = cl + (size_t)blockIdx.y * npix;
    const u32 *crf = cr + (size_t)blockIdx.y * (size_t)h * wp + 1;
    uint8_t *Ef = E + (size_t)blockIdx.y * npix * Dp;
    const size_t ring = (size_t)(2 * w + 2 * h) * Dp;
    uint16_t *BLf = BL + (size_t)blockIdx.y * ring;
    uint16_t *SBf = SB + (size_t)blockIdx.y * ring;

    const PathGeom g = path_geometry(p, w, h, a.dx, a.dy);
    const int len = g.len;
    const int pstride = a.dy * w + a.dx;   // pixels per step in the dense images
    const int cstride = a.dy * wp + a.dx;  // elements per step in the padded census-right image
    const int pix0 = g.y0 * w + g.x0;
    const int cix0 = g.y0 * wp + g.x0;

    const u32 P1p = (u32)a.P1 * 0x10001u, P2p = (u32)a.P2 * 0x10001u;
    u32 padfloor[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int d = d0 + 2 * j;
        padfloor[j] = (d >= a.D ? INF16 : 0u) | (d + 1 >= a.D ? INF16 << 16 : 0u);
    }

    // ---- step 0, the start pixel: L_k = L_{k-1} + C; its E byte is 0, the carry goes to SB
    u32 prior[R];
    {
        const size_t bo = (size_t)border_index(g.y0, g.x0, w, h) * Dp + loff;
        u32 c[R];
        cost_from_census<R, true>(load_cens<R>(crf + cix0 + coff), clf[pix0], g.x0, d0, c);
        u32 zero[R];
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < R; j++)
                prior[j] = c[j];
        } else {
            const SRaw<R> bl = load_s<R>(BLf + bo);
            const SRaw<R> sb = load_s<R>(SBf + bo);
            u32 sbn[R];
#pragma unroll
            for (int j = 0; j < R; j++) {
                prior[j] = pk_add(bl.v[j], c[j]);
                sbn[j] = pk_add(sb.v[j], bl.v[j]);
            }
            store_s<R>(SBf + bo, sbn);
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (HAS_PAD)
                prior[j] = pk_max(prior[j], padfloor[j]);
            zero[j] = 0u;
        }
        store_e<R>(Ef + (size_t)pix0 * Dp + loff, zero);
        store_s<R>(BLf + bo, prior);
    }
    if (len == 1)
        return;

    // ---- steps 1 .. len-1; ring slot i holds a step t with (t-1) % PF == i
    CensRaw<R> cq[PF];
    u32 lq[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        const int t = 1 + i < len ? 1 + i : len - 1;
        cq[i] = load_cens<R>(crf + (cix0 + t * cstride) + coff);
        lq[i] = clf[pix0 + t * pstride];
    }
    const int xl = g.x0 + (len - 1) * a.dx, yl = g.y0 + (len - 1) * a.dy; // last pixel
    uint16_t *bl_last = BLf + (size_t)border_index(yl, xl, w, h) * Dp;

    int pix = pix0, cix = cix0, x = g.x0;
    auto step = [&](int t, int i, bool prefetch) {
        pix += pstride;
        cix += cstride;
        x += a.dx;
        const CensRaw<R> raw = cq[i];
        const u32 clv = lq[i];
        if (prefetch) { // census words PF steps ahead
            cq[i] = load_cens<R>(crf + (cix + PF * cstride) + coff);
            lq[i] = clf[pix + PF * pstride];
        }
        u32 c[R];
        if (x < a.D - 1) // wave-uniform: only the first D-1 columns have d > x
            cost_from_census<R, true>(raw, clv, x, d0, c);
        else
            cost_from_census<R, false>(raw, clv, x, d0, c);

        // stereo_sgm.h:129-146
        const u32 left = dpp<0x138>(INF2, prior[R - 1]); // lane-1's top pair; lane 0 has no d-1
        const u32 right = dpp<0x130>(INF2, prior[0]);    // lane+1's bottom pair; lane 63 has no d+1
        u32 mn = prior[0];
#pragma unroll
        for (int j = 1; j < R; j++)
            mn = pk_min(mn, prior[j]);
        const u32 m = wave_min_u32(min(mn & 0xFFFFu, mn >> 16));
        const u32 mp = m * 0x10001u;
        u32 e[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const u32 prev = __builtin_amdgcn_alignbit(prior[j], j ? prior[j - 1] : left, 16);
            const u32 next = __builtin_amdgcn_alignbit(j < R - 1 ? prior[j + 1] : right, prior[j], 16);
            const u32 A = pk_min(prior[j], pk_add(pk_min(prev, next), P1p));
            e[j] = pk_min(pk_sub(A, mp), P2p);
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            prior[j] = pk_add(e[j], c[j]);
            if (HAS_PAD)
                prior[j] = pk_max(prior[j], padfloor[j]);
        }
        store_e<R>(Ef + (size_t)pix * Dp + loff, e);
        if (g.edge) { // path along an image edge: every pixel feeds a later direction's start
            const int y = g.y0 + t * a.dy;
            store_s<R>(BLf + (size_t)border_index(y, x, w, h) * Dp + loff, prior);
        } else if (t == len - 1) {
            store_s<R>(bl_last + loff, prior);
        }
    };

    int t = 1;
    for (; t + 2 * PF <= len; t += PF) { // every prefetch target t+i+PF <= len-1
#pragma unroll
        for (int i = 0; i < PF; i++)
            step(t + i, i, true);
    }
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (t + i < len)
            step(t + i, i, t + i + PF < len);
    t += PF;
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (t + i < len)
            step(t + i, i, false);
}

// ---------------------------------------------------------------------------
// S(p, 16g .. 16g+15) of one pixel as 8 packed uint16 pairs, rebuilt from the census images, the
// N byte volumes and the border accumulator.  16 disparities per lane, 8R lanes per pixel.
// ---------------------------------------------------------------------------
struct SumArgs {
    const u32 *cl, *cr;
    const uint8_t *E;
    size_t estride; // bytes between the volumes of two directions
    const uint16_t *SB;
    int w, h, D, wp, N;
};

template <int R>
__device__ __forceinline__ void pixel_sums(const SumArgs &a, int pair, int y, int x, int g, u32 (&S)[8])
{
    constexpr int Dp = 128 * R;
    const size_t pix = ((size_t)pair * a.h + y) * a.w + x;
    const u32 clv = a.cl[pix];
    u32 r[16]; // r[i] = censR[x - 16g - 15 + i]  <->  d = 16g + 15 - i
    __builtin_memcpy(r, a.cr + ((size_t)pair * a.h + y) * a.wp + (Dp + x - 16 * g - 15), 64);
    const uint8_t *ep = a.E + pix * Dp + 16 * g;
    uint4 ev[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < a.N)
            ev[k] = *reinterpret_cast<const uint4 *>(ep + (size_t)k * a.estride);
    const int sh = a.N == 8 ? 3 : 2;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int d = 16 * g + 2 * q;
        u32 lo = (u32)__builtin_popcount(r[15 - 2 * q] ^ clv);
        u32 hi = (u32)__builtin_popcount(r[14 - 2 * q] ^ clv);
        lo = d > x ? 255u : lo;
        hi = d + 1 > x ? 255u : hi;
        S[q] = (lo | (hi << 16)) << sh; // N * C
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (k < a.N) {
            const u32 e4[4] = {ev[k].x, ev[k].y, ev[k].z, ev[k].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                S[2 * q] = pk_add(S[2 * q], __builtin_amdgcn_perm(0u, e4[q], 0x0c010c00u));
                S[2 * q + 1] = pk_add(S[2 * q + 1], __builtin_amdgcn_perm(0u, e4[q], 0x0c030c02u));
            }
        }
    if (y == 0 || y == a.h - 1 || x == 0 || x == a.w - 1) {
        const uint16_t *sp = a.SB + ((size_t)pair * (2 * a.w + 2 * a.h) + border_index(y, x, a.w, a.h)) * Dp + 16 * g;
        const uint4 s0 = *reinterpret_cast<const uint4 *>(sp), s1 = *reinterpret_cast<const uint4 *>(sp + 8);
        S[0] = pk_add(S[0], s0.x); S[1] = pk_add(S[1], s0.y); S[2] = pk_add(S[2], s0.z); S[3] = pk_add(S[3], s0.w);
        S[4] = pk_add(S[4], s1.x); S[5] = pk_add(S[5], s1.y); S[6] = pk_add(S[6], s1.z); S[7] = pk_add(S[7], s1.w);
    }
}

// ---------------------------------------------------------------------------
// K4+K5' WTA, right-view argmin, LR check, subpixel (stereo_sgm.h:209-241, 275-317) on the
// rebuilt S.  One workgroup per image row; a wave handles 8/R pixels at a time, 16 disparities
// per lane.  Right view: LDS atomicMin of (cost << 16 | d) keys into slot x-d, i.e. the
// reference's diagonal scan S[y][xr+i][i] regrouped by xr; the smaller d wins ties like its
// strict '<' scan.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void wta2_kernel(const SumArgs a, int m, uint16_t *__restrict__ raw)
{
    constexpr int Dp = 128 * R;
    constexpr int LPP = 8 * R;    // lanes per pixel
    constexpr int PPW = 64 / LPP; // pixels per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = a.w, h = a.h, D = a.D;
    u32 *tile = reinterpret_cast<u32 *>(smem);       // [4 waves][PPW][Dp/2] packed S of the pixels in flight
    u32 *rkey = tile + 4 * PPW * (Dp / 2);           // [w] right-view best key
    u32 *lkey = rkey + w;                            // [w] left-view best key
    u32 *s3a = lkey + w;                             // [w] S[d-1] | S[d] << 16 at the left winner
    u32 *s3b = s3a + w;                              // [w] S[d+1]
    const int y = blockIdx.x, pair = blockIdx.y;
    const int W = w + 2 * m;
    uint16_t *out = raw + ((size_t)pair * (h + 2 * m) + (y + m)) * (size_t)W + m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane / LPP, g = lane % LPP;
    for (int x = threadIdx.x; x < w; x += blockDim.x)
        rkey[x] = 0xFFFFFFFFu;
    __syncthreads();
    u32 *mytile = tile + (wave * PPW + j) * (Dp / 2);
    for (int x0 = wave * PPW; x0 < w; x0 += 4 * PPW) {
        const int x = x0 + j;
        const bool active = x < w;
        const int xc = active ? x : w - 1;
        u32 S[8];
        pixel_sums<R>(a, pair, y, xc, g, S);
        u32 best = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int d = 16 * g + i;
            const u32 v = (i & 1) ? (S[i >> 1] >> 16) : (S[i >> 1] & 0xFFFFu);
            if (active && d < D) {
                const u32 key = (v << 16) | (u32)d;
                best = min(best, key);
                if (d <= x)
                    atomicMin(&rkey[x - d], key);
            }
        }
        *reinterpret_cast<uint4 *>(mytile + 8 * g) = make_uint4(S[0], S[1], S[2], S[3]);
        *reinterpret_cast<uint4 *>(mytile + 8 * g + 4) = make_uint4(S[4], S[5], S[6], S[7]);
#pragma unroll
        for (int msk = 1; msk < LPP; msk <<= 1)
            best = min(best, (u32)__shfl_xor((int)best, msk));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active && g == 0) {
            lkey[x] = best;
            const int d = (int)(best & 0xFFFFu);
            if (d >= 1 && d < D - 1) {
                const uint16_t *t16 = reinterpret_cast<const uint16_t *>(mytile);
                s3a[x] = (u32)t16[d - 1] | ((u32)t16[d] << 16);
                s3b[x] = (u32)t16[d + 1];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); // the tile is overwritten by the next iteration
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
template <int R>
__global__ __launch_bounds__(256) void sums_tap_kernel(const SumArgs a, int pair, uint16_t *__restrict__ out)
{
    constexpr int LPP = 8 * R;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t pixi = t / LPP;
    const int g = (int)(t % LPP);
    if (pixi >= (size_t)a.w * a.h)
        return;
    const int y = (int)(pixi / a.w), x = (int)(pixi % a.w);
    u32 S[8];
    pixel_sums<R>(a, pair, y, x, g, S);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int d = 16 * g + i;
        if (d < a.D)
            out[pixi * a.D + d] = (uint16_t)((i & 1) ? (S[i >> 1] >> 16) : (S[i >> 1] & 0xFFFFu));
    }
}

} // namespace sgmx
