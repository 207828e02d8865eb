// sgmx_kernels.hpp -- gfx950 (CDNA4, wave64) kernels of the SGM engine.
//
// Data layout in HBM (per stereo pair, pairs of a launch group back to back):
//   census  uint32 [h][w]                         (stereo_costs.cc:36-61)
//   C       uint8  [h][w][Dp]   Dp = 128*R >= D   (stereo_sgm.cc:75-101; pad d >= D holds 0)
//   S       uint16 [h][w][Dp]   summed path costs (aggr_costs, stereo_sgm.cc:133)
//   BL      uint16 [2w+2h][Dp]  path costs L_k of the image-border pixels only
//   raw/out uint16 [H][W]
// The reference keeps a full path-cost volume that it never clears between
// directions (stereo_sgm.cc:46, :142): direction k starts from L_{k-1}(p) + C(p)
// on its start pixels.  Start pixels are always border pixels, so carrying the
// border ring BL from direction to direction reproduces that exactly without a
// [h][w][D] path-cost volume in memory.
//
// Aggregation mapping: one wavefront per path (scanline / column / diagonal),
// the disparity axis across the 64 lanes, 2R disparities per lane as R packed
// uint16 pairs held in registers; neighbours d+-1 come from DPP wave shifts,
// min_k from a DPP row reduction; the cost volume is streamed with loads issued
// PF steps ahead of their use.  No LDS, no MFMA (the path is a min-plus
// recurrence, not a contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgmx {

typedef uint32_t u32;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// "no such disparity": larger than any reachable path cost, and INF + P1 still
// fits uint16 (the host validates the penalties against these guards).
constexpr u32 INF16 = 0x7FFFu;
constexpr u32 INF2 = 0x7FFF7FFFu;

__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return __builtin_bit_cast(u32, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }

// DPP controls (gfx9 family): row_shr:n = 0x110+n, wave_shl:1 = 0x130, wave_shr:1 = 0x138,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ u32 dpp(u32 old, u32 src)
{
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}

// Minimum of a 32-bit value over the 64 lanes, returned wave-uniform (SGPR).
// `old` = UINT_MAX is the identity of umin, which lets the compiler fold each DPP
// move into a single v_min_u32_dpp (lanes without a source keep their value).
__device__ __forceinline__ u32 wave_min_u32(u32 x)
{
    x = min(x, dpp<0x111>(0xFFFFFFFFu, x));          // row_shr:1
    x = min(x, dpp<0x112>(0xFFFFFFFFu, x));          // row_shr:2
    x = min(x, dpp<0x114>(0xFFFFFFFFu, x));          // row_shr:4
    x = min(x, dpp<0x118>(0xFFFFFFFFu, x));          // row_shr:8  -> lane 15 of each row
    x = min(x, dpp<0x142, 0xA>(0xFFFFFFFFu, x));     // row_bcast:15 -> lanes 31, 63
    x = min(x, dpp<0x143, 0xC>(0xFFFFFFFFu, x));     // row_bcast:31 -> lane 63
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
}

// float <-> uint32 whose unsigned order is the float order (-0 folded onto +0)
__device__ __forceinline__ u32 float_key(float f)
{
    const u32 b = __builtin_bit_cast(u32, f + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(u32 k)
{
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ float wave_min_f32(float x) { return key_float(wave_min_u32(float_key(x))); }

// ---------------------------------------------------------------------------
// K1 census transform.  One thread per cropped pixel; z = image index (n left
// images then n right images).  stereo_costs.cc:36-61.
// ---------------------------------------------------------------------------
// The right census image may be stored with a row pitch wpr >= w and a column offset xoffr (the
// padded layout of sgmx_kernels_v2.hpp); the dense layout is wpr = w, xoffr = 0.
__global__ __launch_bounds__(256) void census_kernel(const uint8_t *__restrict__ left, const uint8_t *__restrict__ right,
                                                     int n, int W, int H, int wsz, u32 *__restrict__ cens_left,
                                                     u32 *__restrict__ cens_right, int wpr, int xoffr)
{
    const int m = (wsz - 1) / 2;
    const int w = W - 2 * m, h = H - 2 * m;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= w || y >= h)
        return;
    const uint8_t *img = (z < n ? left + (size_t)z * W * H : right + (size_t)(z - n) * W * H);
    u32 *out = (z < n ? cens_left + (size_t)z * w * h + (size_t)y * w
                      : cens_right + (size_t)(z - n) * wpr * h + (size_t)y * wpr + xoffr);
    const int cy = y + m, cx = x + m;
    const uint8_t c = img[(size_t)cy * W + cx];
    u32 code = 0;
    for (int py = y; py < y + wsz; py++) {
        const uint8_t *row = img + (size_t)py * W;
        for (int px = x; px < x + wsz; px++) {
            if (px == cx && py == cy)
                continue;
            code = (code << 1) + (row[px] > c ? 1u : 0u);
        }
    }
    out[x] = code;
}

// Census for the common small windows: one thread produces 4 adjacent pixels from WSZ row
// segments of WSZ + 3 bytes each, loaded as two (unaligned) dwords instead of byte by byte.
// Same bit order as above.  Needs w >= 1; columns past the row end are clamped and discarded.
template <int WSZ>
__global__ __launch_bounds__(256) void census4_kernel(const uint8_t *__restrict__ left, const uint8_t *__restrict__ right,
                                                      int n, int W, int H, u32 *__restrict__ cens_left,
                                                      u32 *__restrict__ cens_right, int wpr, int xoffr)
{
    constexpr int M = (WSZ - 1) / 2;
    const int w = W - 2 * M, h = H - 2 * M;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    const int z = blockIdx.z;
    if (x >= w || y >= h)
        return;
    const uint8_t *img = (z < n ? left + (size_t)z * W * H : right + (size_t)(z - n) * W * H);
    u32 *out = (z < n ? cens_left + (size_t)z * w * h + (size_t)y * w
                      : cens_right + (size_t)(z - n) * wpr * h + (size_t)y * wpr + xoffr);
    // bytes x .. x+7 of each window row (WSZ + 3 <= 8 of them are used); the last threads of a row
    // would reach past the image row, so their segment start is pulled back (W >= 8) and the pixels
    // are re-indexed by `sh`
    const int xs = x + 8 <= W ? x : W - 8;
    const int sh = x - xs;
    unsigned long long seg[WSZ];
#pragma unroll
    for (int r = 0; r < WSZ; r++) {
        __builtin_memcpy(&seg[r], img + (size_t)(y + r) * W + xs, 8);
        seg[r] >>= 8 * sh; // byte i + k of the segment = column k of pixel x + i's window
    }
    // The four pixels' comparisons run side by side in the four bytes of a dword (the sweeps' kernels are the
    // budget; this one was 330 VALU instructions per thread with a compare and a select per bit).  Unsigned
    // "centre < neighbour" per byte: t = (c | H) - (n & ~H) has no borrow between bytes and its bit 7 says
    // c_low7 >= n_low7; where the top bits are equal the answer is ~t7, else it is the neighbour's top bit.
    constexpr u32 Hb = 0x80808080u;
    const u32 cen = (u32)(seg[M] >> (8 * M)), cenH = cen | Hb;
    constexpr int NPOS = WSZ * WSZ - 1, NACC = NPOS / 8;
    static_assert(NPOS % 8 == 0, "3x3 and 5x5 windows: 8 or 24 comparisons");
    u32 acc[NACC] = {};
    int pos = 0;
#pragma unroll
    for (int r = 0; r < WSZ; r++)
#pragma unroll
        for (int k = 0; k < WSZ; k++) {
            if (r == M && k == M)
                continue;
            const u32 nb = (u32)(seg[r] >> (8 * k));
            const u32 t = cenH - (nb & ~Hb);
            const u32 lt = (~cen & nb) | (~(cen ^ nb) & ~t); // bit 7 of every byte: centre < neighbour
            acc[pos / 8] = (acc[pos / 8] << 1) | ((lt >> 7) & 0x01010101u); // 8 bits per byte: nothing crosses a byte
            pos++;
        }
    u32 code[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        code[i] = 0;
#pragma unroll
        for (int g = 0; g < NACC; g++)
            code[i] = (code[i] << 8) | ((acc[g] >> (8 * i)) & 0xFFu); // first comparison = highest bit, as stereo_costs.cc:55
    }
    if (x + 4 <= w) {
        __builtin_memcpy(out + x, code, 16); // one 16-byte store (4-byte aligned)
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (x + i < w)
                out[x + i] = code[i];
    }
}

// ---------------------------------------------------------------------------
// K2 Hamming data-cost volume.  One thread per (pixel, group of 16 disparities),
// one 16-byte store each, disparity groups fastest so a wave writes whole lines.
// stereo_sgm.cc:43,75-101 + stereo_costs.h:120-132.  Pad disparities d >= D get 0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hamming_volume_kernel(const u32 *__restrict__ cens_left,
                                                             const u32 *__restrict__ cens_right, int wpr, int xoffr,
                                                             int w, int h, int D, int Dp, uint8_t *__restrict__ C,
                                                             size_t total_groups)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_groups)
        return;
    const int gpp = Dp >> 4; // groups per pixel
    const size_t pix = t / gpp; // pixel index over all pairs of the launch group
    const int g = (int)(t - pix * gpp);
    const int x = (int)(pix % (size_t)w);
    const u32 cl = cens_left[pix];
    const u32 *cr = cens_right + (pix / (size_t)w) * (size_t)wpr + xoffr + x; // pix / w = row over all pairs
    u32 o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int d = g * 16 + q * 4 + b;
            u32 c;
            if (d >= D)
                c = 0u;
            else if (d > x)
                c = 255u;
            else
                c = (u32)__builtin_popcount(cl ^ cr[-d]);
            v |= c << (8 * b);
        }
        o[q] = v;
    }
    *reinterpret_cast<uint4 *>(C + pix * (size_t)Dp + (size_t)g * 16) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---------------------------------------------------------------------------
// K3 path aggregation, census build (uint8 C, uint16 L and S).
// stereo_sgm.cc:191-268 (sweep), stereo_sgm.h:119-147 (recurrence), :85-117 (sum).
// ---------------------------------------------------------------------------
struct AggArgs {
    const void *C;    // data cost, [pairs][h][w][Dp]
    void *S;          // summed path cost, same shape
    void *BL;         // border ring, [pairs][2w+2h][Dp]
    int w, h, D;      // cropped size, real disparity count
    int dx, dy;       // direction; predecessor of (y,x) is (y-dy, x-dx)
    int P1, P2;
    int npaths;       // paths of this direction in one pair
    int first;        // 1: S = L (first direction), 0: S += L
    int wrap;         // diagonal directions: 1 = npaths = w wrapped diagonals of length h (WrapWalk below)
};

// Wrapped diagonals for the one-path-per-wave kernels (the volume-free kernels' WRAP, sgmx_kernels_v2.hpp): a wave walks
// the pixels ((c + t dx) mod w, y0 + t dy), t = 0 .. h-1 -- diagonal c and then, from the opposite image column on, the
// diagonal that starts one row further: w equally long paths instead of w + h - 1 of lengths 1 .. min(w, h).  The pixel
// before a wrap ends a diagonal (its L goes into the ring), the pixel after it starts one (L_{k-1} + C from the ring,
// stereo_sgm.cc:201,236).  Everything here is wave-uniform.
struct WrapWalk {
    int w, dx, stride, back; // image width, column step, voxels per step, voxels a wrap takes back (dx * w * Dp)
    bool on;
    // one step on: column x and voxel offset off; true when the step wrapped (x is then the start column)
    __device__ __forceinline__ bool advance(int &x, int &off) const
    {
        x += dx;
        off += stride;
        if (on && (x == w || x < 0)) {
            x -= dx * w;
            off -= back;
            return true;
        }
        return false;
    }
};

// Path p of direction (dx,dy): start pixel and length.  A pixel starts a path iff
// its predecessor is outside the image: the whole edge column x = (dx>0 ? 0 : w-1)
// when dx != 0, plus the edge row y = (dy>0 ? 0 : h-1) when dy != 0 without the
// corner the column already owns (the de-duplication at stereo_sgm.cc:216,234,252).
// All values are wave-uniform and forced into SGPRs.
struct PathGeom {
    int x0, y0, len;
    bool edge; // the whole path runs along an image edge (every pixel is a border pixel)
};
__device__ __forceinline__ PathGeom path_geometry(int p, int w, int h, int dx, int dy)
{
    const int xe = dx > 0 ? 0 : w - 1; // start column when the path starts on the column edge
    const int ye = dy > 0 ? 0 : h - 1; // start row when it starts on the row edge
    const bool on_col = (dy == 0) || (dx != 0 && p < h);
    const int q = p - h;
    int x0 = on_col ? xe : (dx == 0 ? p : (dx > 0 ? q + 1 : q));
    int y0 = on_col ? p : ye;
    const int sx = dx == 0 ? 0x7fffffff : (dx > 0 ? w - x0 : x0 + 1);
    const int sy = dy == 0 ? 0x7fffffff : (dy > 0 ? h - y0 : y0 + 1);
    PathGeom g;
    g.x0 = __builtin_amdgcn_readfirstlane(x0);
    g.y0 = __builtin_amdgcn_readfirstlane(y0);
    g.len = __builtin_amdgcn_readfirstlane(sx < sy ? sx : sy);
    g.edge = (dy == 0 && (g.y0 == 0 || g.y0 == h - 1)) || (dx == 0 && (g.x0 == 0 || g.x0 == w - 1));
    return g;
}

__device__ __forceinline__ int border_index(int y, int x, int w, int h)
{
    if (y == 0)
        return x;
    if (y == h - 1)
        return w + x;
    if (x == 0)
        return 2 * w + y;
    return 2 * w + h + y;
}

// 2R cost bytes of one lane -> R packed uint16 pairs.
template <int R> struct CRaw;
template <> struct CRaw<1> { unsigned short v; };
template <> struct CRaw<2> { u32 v; };
template <> struct CRaw<4> { uint2 v; };
template <int R> struct SRaw { u32 v[R]; };

__device__ __forceinline__ void unpack_c(const CRaw<1> &c, u32 (&o)[1]) { o[0] = __builtin_amdgcn_perm(0u, (u32)c.v, 0x0c010c00u); }
__device__ __forceinline__ void unpack_c(const CRaw<2> &c, u32 (&o)[2])
{
    o[0] = __builtin_amdgcn_perm(0u, c.v, 0x0c010c00u);
    o[1] = __builtin_amdgcn_perm(0u, c.v, 0x0c030c02u);
}
__device__ __forceinline__ void unpack_c(const CRaw<4> &c, u32 (&o)[4])
{
    o[0] = __builtin_amdgcn_perm(0u, c.v.x, 0x0c010c00u);
    o[1] = __builtin_amdgcn_perm(0u, c.v.x, 0x0c030c02u);
    o[2] = __builtin_amdgcn_perm(0u, c.v.y, 0x0c010c00u);
    o[3] = __builtin_amdgcn_perm(0u, c.v.y, 0x0c030c02u);
}

template <int R> __device__ __forceinline__ CRaw<R> load_c(const uint8_t *p);
template <> __device__ __forceinline__ CRaw<1> load_c<1>(const uint8_t *p) { return CRaw<1>{*reinterpret_cast<const unsigned short *>(p)}; }
template <> __device__ __forceinline__ CRaw<2> load_c<2>(const uint8_t *p) { return CRaw<2>{*reinterpret_cast<const u32 *>(p)}; }
template <> __device__ __forceinline__ CRaw<4> load_c<4>(const uint8_t *p) { return CRaw<4>{*reinterpret_cast<const uint2 *>(p)}; }

template <int R> __device__ __forceinline__ SRaw<R> load_s(const uint16_t *p);
template <> __device__ __forceinline__ SRaw<1> load_s<1>(const uint16_t *p) { return SRaw<1>{{*reinterpret_cast<const u32 *>(p)}}; }
template <> __device__ __forceinline__ SRaw<2> load_s<2>(const uint16_t *p)
{
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    return SRaw<2>{{t.x, t.y}};
}
template <> __device__ __forceinline__ SRaw<4> load_s<4>(const uint16_t *p)
{
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    return SRaw<4>{{t.x, t.y, t.z, t.w}};
}
template <int R> __device__ __forceinline__ void store_s(uint16_t *p, const u32 (&v)[R]);
template <> __device__ __forceinline__ void store_s<1>(uint16_t *p, const u32 (&v)[1]) { *reinterpret_cast<u32 *>(p) = v[0]; }
template <> __device__ __forceinline__ void store_s<2>(uint16_t *p, const u32 (&v)[2]) { *reinterpret_cast<uint2 *>(p) = make_uint2(v[0], v[1]); }
template <> __device__ __forceinline__ void store_s<4>(uint16_t *p, const u32 (&v)[4]) { *reinterpret_cast<uint4 *>(p) = make_uint4(v[0], v[1], v[2], v[3]); }

// One step of the recurrence for the R register pairs of a lane (stereo_sgm.h:129-146):
// L = C + min(prior[d], prior[d-1]+P1, prior[d+1]+P1, m+P2) - m with m = min_k prior[k].
template <int R>
__device__ __forceinline__ void path_step_u16(const u32 (&prior)[R], const u32 (&c)[R], u32 P1p, u32 P2p, u32 (&L)[R])
{
    const u32 left = dpp<0x138>(INF2, prior[R - 1]); // lane-1's top pair; lane 0 has no d-1
    const u32 right = dpp<0x130>(INF2, prior[0]);    // lane+1's bottom pair; lane 63 has no d+1
    u32 mn = prior[0];
#pragma unroll
    for (int j = 1; j < R; j++)
        mn = pk_min(mn, prior[j]);
    const u32 m = wave_min_u32(min(mn & 0xFFFFu, mn >> 16));
    const u32 mp = m * 0x10001u;
#pragma unroll
    for (int j = 0; j < R; j++) {
        const u32 prev = __builtin_amdgcn_alignbit(prior[j], j ? prior[j - 1] : left, 16);
        const u32 next = __builtin_amdgcn_alignbit(j < R - 1 ? prior[j + 1] : right, prior[j], 16);
        const u32 A = pk_min(prior[j], pk_add(pk_min(prev, next), P1p));
        L[j] = pk_add(pk_min(pk_sub(A, mp), P2p), c[j]);
    }
}

template <int R, bool FIRST, bool HAS_PAD, int PF>
__global__ __launch_bounds__(256) void aggregate_u16_kernel(const AggArgs a)
{
    constexpr int Dp = 128 * R;
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (p >= a.npaths)
        return;
    const int w = a.w, h = a.h;
    const size_t npix = (size_t)w * h;
    const unsigned loff = lane * (2 * R); // this lane's first disparity
    // wave-uniform bases; per-lane part is the 32-bit offset loff (saddr + voffset addressing)
    const uint8_t *Cf = static_cast<const uint8_t *>(a.C) + (size_t)blockIdx.y * npix * Dp;
    uint16_t *Sf = static_cast<uint16_t *>(a.S) + (size_t)blockIdx.y * npix * Dp;
    uint16_t *BLf = static_cast<uint16_t *>(a.BL) + (size_t)blockIdx.y * (size_t)(2 * w + 2 * h) * Dp;

    PathGeom g = path_geometry(p, w, h, a.dx, a.dy);
    if (a.wrap) {
        g.x0 = p, g.y0 = a.dy > 0 ? 0 : h - 1, g.len = h, g.edge = false;
    }
    const int len = g.len;
    const int stride = (a.dy * w + a.dx) * Dp; // voxels per step; one pair holds < 2^31 voxels
    const int off0 = (g.y0 * w + g.x0) * Dp;
    const WrapWalk walk{w, a.dx, stride, a.dx * w * Dp, a.wrap != 0};
    const int xend = a.dx > 0 ? w - 1 : 0;

    const u32 P1p = (u32)a.P1 * 0x10001u, P2p = (u32)a.P2 * 0x10001u;
    u32 padfloor[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int d = (int)loff + 2 * j;
        padfloor[j] = (d >= a.D ? INF16 : 0u) | (d + 1 >= a.D ? INF16 << 16 : 0u);
    }

    // ---- step 0, the start pixel: buf += C (stereo_sgm.cc:201,218,236,254); buf is
    // L_{k-1} there (zero before the first direction, stereo_sgm.cc:46)
    u32 prior[R];
    {
        uint16_t *blp = BLf + (size_t)border_index(g.y0, g.x0, w, h) * Dp;
        const SRaw<R> bl = load_s<R>(blp + loff);
        u32 c[R];
        unpack_c(load_c<R>(Cf + off0 + loff), c);
        SRaw<R> s;
        if (!FIRST)
            s = load_s<R>(Sf + off0 + loff);
        u32 sn[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            prior[j] = pk_add(bl.v[j], c[j]);
            if (HAS_PAD)
                prior[j] = pk_max(prior[j], padfloor[j]);
            sn[j] = FIRST ? prior[j] : pk_add(s.v[j], prior[j]);
        }
        store_s<R>(Sf + off0 + loff, sn);
        store_s<R>(blp + loff, prior);
    }
    if (len == 1)
        return;

    // ---- steps 1 .. len-1; ring slot i holds a step t with (t-1) % PF == i
    CRaw<R> cq[PF];
    SRaw<R> sq[PF];
    int xp = g.x0, op = off0; // column and voxel offset of the newest prefetched step
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (1 + i < len)
            walk.advance(xp, op);
        cq[i] = load_c<R>(Cf + op + loff);
        if (!FIRST)
            sq[i] = load_s<R>(Sf + op + loff);
    }
    const int xl = g.x0 + (len - 1) * a.dx, yl = g.y0 + (len - 1) * a.dy; // last pixel (paths that do not wrap)
    uint16_t *bl_last = BLf + (size_t)border_index(yl, a.wrap ? 0 : xl, w, h) * Dp;

    int o = off0, xc = g.x0; // voxel offset and column of the current step, advanced by the step lambda
    auto step = [&](int t, int i, bool prefetch) {
        const bool started = walk.advance(xc, o); // wrapped: this pixel starts the next diagonal
        u32 c[R];
        unpack_c(cq[i], c);
        SRaw<R> s;
        if (!FIRST)
            s = sq[i];
        if (prefetch) { // stream the volume PF steps ahead
            walk.advance(xp, op);
            cq[i] = load_c<R>(Cf + op + loff);
            if (!FIRST)
                sq[i] = load_s<R>(Sf + op + loff);
        }
        const int y = g.y0 + t * a.dy;
        u32 L[R];
        if (started) { // buf += C on a start pixel (stereo_sgm.cc:201,236), as in step 0
            const SRaw<R> bl = load_s<R>(BLf + (size_t)border_index(y, xc, w, h) * Dp + loff);
#pragma unroll
            for (int j = 0; j < R; j++)
                L[j] = pk_add(bl.v[j], c[j]);
        } else {
            path_step_u16<R>(prior, c, P1p, P2p, L);
        }
        u32 sn[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (HAS_PAD)
                L[j] = pk_max(L[j], padfloor[j]);
            sn[j] = FIRST ? L[j] : pk_add(s.v[j], L[j]);
            prior[j] = L[j];
        }
        store_s<R>(Sf + o + loff, sn);
        if (a.wrap) { // first or last pixel of a diagonal: a border pixel that feeds a later direction's start
            if (started || xc == xend || t == len - 1)
                store_s<R>(BLf + (size_t)border_index(y, xc, w, h) * Dp + loff, L);
        } else if (g.edge) { // path along an image edge: every pixel feeds a later direction's start
            const int x = g.x0 + t * a.dx;
            store_s<R>(BLf + (size_t)border_index(y, x, w, h) * Dp + loff, L);
        } else if (t == len - 1) {
            store_s<R>(bl_last + loff, L);
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
// K3f path aggregation, float build (float C, L, S).  sgm.cc:177-254, sgm.h:129-157.
// Lane l holds d = 2Rl .. 2Rl+2R-1 as 2R floats.  Same op order as the reference:
// e = mp + P2; e = min(e, prior[d]); e = min(e, prior[d-1] + P1); e = min(e,
// prior[d+1] + P1); L = C + (e - mp).  No multiply anywhere, so no contraction.
// ---------------------------------------------------------------------------
template <int R, bool FIRST, bool HAS_PAD, int PF>
__global__ __launch_bounds__(256) void aggregate_f32_kernel(const AggArgs a)
{
    constexpr int N = 2 * R; // floats per lane
    constexpr int Dp = 64 * N;
    typedef float fvec __attribute__((ext_vector_type(N)));
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (p >= a.npaths)
        return;
    const int w = a.w, h = a.h;
    const size_t npix = (size_t)w * h;
    const unsigned loff = lane * N;
    const float *Cf = static_cast<const float *>(a.C) + (size_t)blockIdx.y * npix * Dp;
    float *Sf = static_cast<float *>(a.S) + (size_t)blockIdx.y * npix * Dp;
    float *BLf = static_cast<float *>(a.BL) + (size_t)blockIdx.y * (size_t)(2 * w + 2 * h) * Dp;

    PathGeom g = path_geometry(p, w, h, a.dx, a.dy);
    if (a.wrap) {
        g.x0 = p, g.y0 = a.dy > 0 ? 0 : h - 1, g.len = h, g.edge = false;
    }
    const int len = g.len;
    const int stride = (a.dy * w + a.dx) * Dp; // voxels per step; one pair holds < 2^31 voxels
    const int off0 = (g.y0 * w + g.x0) * Dp;
    const WrapWalk walk{w, a.dx, stride, a.dx * w * Dp, a.wrap != 0};
    const int xend = a.dx > 0 ? w - 1 : 0;
    const float P1f = (float)a.P1, P2f = (float)a.P2;
    const float inf = __builtin_inff();

    auto ld = [](const float *q) { return *reinterpret_cast<const fvec *>(q); };
    auto st = [](float *q, const float (&v)[N]) {
        fvec t;
#pragma unroll
        for (int j = 0; j < N; j++)
            t[j] = v[j];
        *reinterpret_cast<fvec *>(q) = t;
    };
    // the volumes are streamed once per direction: non-temporal (835 -> 859 volumes/s at KITTI size)
    auto ldv = [](const float *q) { return __builtin_nontemporal_load(reinterpret_cast<const fvec *>(q)); };
    auto stv = [](float *q, const float (&v)[N]) {
        fvec t;
#pragma unroll
        for (int j = 0; j < N; j++)
            t[j] = v[j];
        __builtin_nontemporal_store(t, reinterpret_cast<fvec *>(q));
    };

    float prior[N];
    {
        const size_t bo = (size_t)border_index(g.y0, g.x0, w, h) * Dp + loff;
        const fvec bl = ld(BLf + bo);
        const fvec c = ldv(Cf + off0 + loff);
        fvec s;
        if (!FIRST)
            s = ldv(Sf + off0 + loff);
        float sn[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            prior[j] = bl[j] + c[j]; // sgm.h:95-102 sum_vectors on the start pixel
            if (HAS_PAD && (int)loff + j >= a.D)
                prior[j] = inf;
            sn[j] = FIRST ? prior[j] : s[j] + prior[j];
        }
        stv(Sf + off0 + loff, sn);
        st(BLf + bo, prior);
    }
    if (len == 1)
        return;

    fvec cq[PF], sq[PF];
    int xp = g.x0, op = off0; // column and voxel offset of the newest prefetched step
#pragma unroll
    for (int i = 0; i < PF; i++) {
        if (1 + i < len)
            walk.advance(xp, op);
        cq[i] = ldv(Cf + op + loff);
        if (!FIRST)
            sq[i] = ldv(Sf + op + loff);
    }
    const int xl = g.x0 + (len - 1) * a.dx, yl = g.y0 + (len - 1) * a.dy; // last pixel (paths that do not wrap)
    const size_t bl_last = (size_t)border_index(yl, a.wrap ? 0 : xl, w, h) * Dp;

    int o = off0, xc = g.x0;
    auto step = [&](int t, int i, bool prefetch) {
        const bool started = walk.advance(xc, o); // wrapped: this pixel starts the next diagonal
        const fvec c = cq[i];
        fvec s;
        if (!FIRST)
            s = sq[i];
        if (prefetch) {
            walk.advance(xp, op);
            cq[i] = ldv(Cf + op + loff);
            if (!FIRST)
                sq[i] = ldv(Sf + op + loff);
        }
        const int y = g.y0 + t * a.dy;
        fvec blv;
        if (started) // sum_vectors on a start pixel (sgm.h:95-102), as in step 0
            blv = ld(BLf + (size_t)border_index(y, xc, w, h) * Dp + loff);
        const float left = __builtin_bit_cast(float, dpp<0x138>(__builtin_bit_cast(u32, inf), __builtin_bit_cast(u32, prior[N - 1])));
        const float right = __builtin_bit_cast(float, dpp<0x130>(__builtin_bit_cast(u32, inf), __builtin_bit_cast(u32, prior[0])));
        float mn = prior[0];
#pragma unroll
        for (int j = 1; j < N; j++)
            mn = fminf(mn, prior[j]);
        const float mp = wave_min_f32(mn);
        const float e0 = mp + P2f;
        float L[N], sn[N];
#pragma unroll
        for (int j = 0; j < N; j++) {
            const float pv = j ? prior[j - 1] : left;
            const float nx = j < N - 1 ? prior[j + 1] : right;
            float e = fminf(e0, prior[j]);
            e = fminf(e, pv + P1f);
            e = fminf(e, nx + P1f);
            L[j] = c[j] + (e - mp);
            if (started)
                L[j] = blv[j] + c[j];
            if (HAS_PAD && (int)loff + j >= a.D)
                L[j] = inf;
            sn[j] = FIRST ? L[j] : s[j] + L[j];
        }
#pragma unroll
        for (int j = 0; j < N; j++)
            prior[j] = L[j];
        stv(Sf + o + loff, sn);
        if (a.wrap) { // first or last pixel of a diagonal
            if (started || xc == xend || t == len - 1)
                st(BLf + (size_t)border_index(y, xc, w, h) * Dp + loff, L);
        } else if (g.edge) {
            const int x = g.x0 + t * a.dx;
            st(BLf + (size_t)border_index(y, x, w, h) * Dp + loff, L);
        } else if (t == len - 1) {
            st(BLf + bl_last + loff, L);
        }
    };

    int t = 1;
    for (; t + 2 * PF <= len; t += PF) {
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
// K4+K5 winner-takes-all, left-right check, subpixel.  stereo_sgm.h:209-241,
// 275-317 (sgm.h:219-251,285-332 for float).  One workgroup per image row.
//  pass 1: one wave per pixel x: first-min argmin over d (key = cost<<16|d, so the
//          smaller d wins ties like the strict '<' scan) and, for the right view,
//          an LDS atomic min of the same keys into slot x-d -- the reference's
//          diagonal scan S[y][xr+i][i] regrouped by xr.
//  pass 2: one thread per pixel: LR test |d - dR(x-d)| > 2, equiangular subpixel
//          in float32, 256.0 * (double), round half away, uint16.
// Float keys: a monotone float->uint32 map in the high word of a 64-bit key.
// ---------------------------------------------------------------------------
// A floating-point value to uint16 the way the reference's x86-64 builds do where the C++ conversion is undefined
// (value outside [0, 65535]: 256 d >= 65536 once disp_range > 256, NaN / infinite sub-pixel terms on the float
// path): cvttsd2si and the low 16 bits, i.e. modulo 65536 while |r| < 2^31, else the low half of the "integer
// indefinite" 0x80000000 = 0.  In range it is the plain conversion.
__device__ __forceinline__ uint16_t cast_u16_like_x86(double r)
{
    return fabs(r) < 2147483648.0 ? (uint16_t)(u32)(int)r : (uint16_t)0;
}

__device__ __forceinline__ uint16_t subpixel_u16(int d, int D, float cl, float cc, float cr, bool *undefined)
{
    if (d >= 1 && d < D - 1) {
        const float num = 0.5f * (cr - cl);
        const float den = (cr < cl) ? (cc - cl) : (cc - cr);
        const float ds = (float)((double)num / (double)den); // correctly rounded float quotient
        const float dsub = (float)d + ds;
        const double r = round(256.0 * (double)dsub);
        if (!(r >= 0.0 && r <= 65535.0) && undefined) // an undefined cast in the reference (see above)
            *undefined = true;
        return cast_u16_like_x86(r);
    }
    return (uint16_t)(256 * d); // modulo 65536 like the cast above
}

template <int R>
__global__ __launch_bounds__(256) void wta_u16_kernel(const uint16_t *__restrict__ S, int w, int h, int D, int m,
                                                      uint16_t *__restrict__ raw)
{
    constexpr int Dp = 128 * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32 *rkey = reinterpret_cast<u32 *>(smem);       // [w] right-view best key
    u32 *lkey = rkey + w;                            // [w] left-view best key
    const int y = blockIdx.x;
    const int W = w + 2 * m;
    const uint16_t *Srow = S + ((size_t)blockIdx.y * h + y) * (size_t)w * Dp;
    uint16_t *out = raw + ((size_t)blockIdx.y * (h + 2 * m) + (y + m)) * (size_t)W + m;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < w; x += blockDim.x)
        rkey[x] = 0xFFFFFFFFu;
    __syncthreads();
    for (int x = wave; x < w; x += 4) {
        const SRaw<R> s = load_s<R>(Srow + (size_t)x * Dp + lane * 2 * R);
        u32 best = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 2 * R; j++) {
            const int d = lane * 2 * R + j;
            const u32 v = (j & 1) ? (s.v[j >> 1] >> 16) : (s.v[j >> 1] & 0xFFFFu);
            if (d < D) {
                const u32 key = (v << 16) | (u32)d;
                best = min(best, key);
                if (d <= x)
                    atomicMin(&rkey[x - d], key);
            }
        }
        best = wave_min_u32(best);
        if (lane == 0)
            lkey[x] = best;
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
                    const uint16_t *v = Srow + (size_t)x * Dp + d;
                    cl = (float)v[-1];
                    cc = (float)v[0];
                    cr = (float)v[1];
                }
                o = subpixel_u16(d, D, cl, cc, cr, nullptr);
            }
        }
        out[x] = o;
    }
}

template <int R>
__global__ __launch_bounds__(256) void wta_f32_kernel(const float *__restrict__ S, int w, int h, int D,
                                                      uint16_t *__restrict__ raw)
{
    constexpr int N = 2 * R;
    constexpr int Dp = 64 * N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *rkey = reinterpret_cast<unsigned long long *>(smem);
    unsigned long long *lkey = rkey + w;
    const int y = blockIdx.x;
    const float *Srow = S + ((size_t)blockIdx.y * h + y) * (size_t)w * Dp;
    uint16_t *out = raw + ((size_t)blockIdx.y * h + y) * (size_t)w;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int x = threadIdx.x; x < w; x += blockDim.x)
        rkey[x] = ~0ull;
    __syncthreads();
    typedef float fvec __attribute__((ext_vector_type(N)));
    // the next pixel's costs are in flight while this one's keys are built
    fvec nxt = *reinterpret_cast<const fvec *>(Srow + (size_t)(wave < w ? wave : w - 1) * Dp + lane * N);
    for (int x = wave; x < w; x += 4) {
        const fvec s = nxt;
        nxt = *reinterpret_cast<const fvec *>(Srow + (size_t)(x + 4 < w ? x + 4 : w - 1) * Dp + lane * N);
        unsigned long long best = ~0ull;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int d = lane * N + j;
            if (d < D) {
                const unsigned long long key = ((unsigned long long)float_key(s[j]) << 32) | (u32)d;
                best = key < best ? key : best;
                if (d <= x)
                    atomicMin(&rkey[x - d], key);
            }
        }
        // 64-bit wave min: high words first, then low words among the lanes that tie
        const u32 hi = wave_min_u32((u32)(best >> 32));
        const u32 lo = wave_min_u32((u32)(best >> 32) == hi ? (u32)best : 0xFFFFFFFFu);
        if (lane == 0)
            lkey[x] = ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += blockDim.x) {
        const int d = (int)(lkey[x] & 0xFFFFFFFFull);
        uint16_t o = 0;
        if (x - d >= 0) {
            const int dr = (int)(rkey[x - d] & 0xFFFFFFFFull);
            if (abs(d - dr) <= 2) {
                float cl = 0.f, cc = 0.f, cr = 0.f;
                if (d >= 1 && d < D - 1) {
                    const float *v = Srow + (size_t)x * Dp + d;
                    cl = v[-1];
                    cc = v[0];
                    cr = v[1];
                }
                o = subpixel_u16(d, D, cl, cc, cr, nullptr);
            }
        }
        out[x] = o;
    }
}

// ---------------------------------------------------------------------------
// K6 3x3 median, replicate border (cv::medianBlur(disp, disp, 3) at
// stereo_sgm.cc:187 / sgm.cc:172).  One thread per pixel, 19-exchange network.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void cswap(u32 &a, u32 &b)
{
    const u32 lo = min(a, b), hi = max(a, b);
    a = lo;
    b = hi;
}

__global__ __launch_bounds__(256) void median3_kernel(const uint16_t *__restrict__ in, int W, int H,
                                                      uint16_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W)
        return;
    const uint16_t *img = in + (size_t)blockIdx.z * W * H;
    const int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
    const int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : H - 1;
    u32 v0 = img[(size_t)ym * W + xm], v1 = img[(size_t)ym * W + x], v2 = img[(size_t)ym * W + xp];
    u32 v3 = img[(size_t)y * W + xm], v4 = img[(size_t)y * W + x], v5 = img[(size_t)y * W + xp];
    u32 v6 = img[(size_t)yp * W + xm], v7 = img[(size_t)yp * W + x], v8 = img[(size_t)yp * W + xp];
    cswap(v1, v2); cswap(v4, v5); cswap(v7, v8);
    cswap(v0, v1); cswap(v3, v4); cswap(v6, v7);
    cswap(v1, v2); cswap(v4, v5); cswap(v7, v8);
    cswap(v0, v3); cswap(v5, v8); cswap(v4, v7);
    cswap(v3, v6); cswap(v1, v4); cswap(v2, v5);
    cswap(v4, v7); cswap(v4, v2); cswap(v6, v4);
    cswap(v4, v2);
    out[((size_t)blockIdx.z * H + y) * W + x] = (uint16_t)v4;
}

// ---------------------------------------------------------------------------
// Caller-side pre-filter cv::GaussianBlur(img, img, Size(3,3), sigma) on 8-bit images
// (dense_stereo/main.cc:83-85, stereo_eval/main.cc:36-38): separable 3-tap kernel with 8 fractional
// bits (k0 centre, k1 sides), int32 row pass, (v + 2^15) >> 16 column pass, reflect-101 border.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss3_u8_kernel(const uint8_t *__restrict__ in, int W, int H, int k0, int k1,
                                                        uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W)
        return;
    const uint8_t *img = in + (size_t)blockIdx.z * W * H;
    const int xm = W == 1 ? 0 : (x == 0 ? 1 : x - 1), xp = W == 1 ? 0 : (x == W - 1 ? W - 2 : x + 1);
    int acc = 0;
#pragma unroll
    for (int j = -1; j <= 1; j++) {
        int yy = y + j;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
        yy = H == 1 ? 0 : yy;
        const uint8_t *r = img + (size_t)yy * W;
        const int row = k1 * ((int)r[xm] + (int)r[xp]) + k0 * (int)r[x];
        acc += (j == 0 ? k0 : k1) * row;
    }
    const int v = (acc + (1 << 15)) >> 16;
    out[((size_t)blockIdx.z * H + y) * W + x] = (uint8_t)(v > 255 ? 255 : v);
}

// ---------------------------------------------------------------------------
// Small helpers: layout conversion for taps / the float input, debug WTA on C,
// Hamming matrix (core/hamming.c:215-236).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void repitch_kernel(const T *__restrict__ src, int Dsrc, T *__restrict__ dst, int Ddst,
                                                      size_t npix, T fill)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npix * (size_t)Ddst)
        return;
    const size_t pix = t / Ddst;
    const int d = (int)(t - pix * Ddst);
    dst[t] = d < Dsrc ? src[pix * Dsrc + d] : fill;
}

__global__ __launch_bounds__(256) void data_cost_wta_kernel(const uint8_t *__restrict__ C, int w, int h, int D, int Dp,
                                                            int m, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w)
        return;
    const uint8_t *v = C + ((size_t)y * w + x) * Dp;
    int d = 0;
    for (int i = 1; i < D; i++)
        if (v[i] < v[d])
            d = i;
    out[(size_t)(y + m) * (w + 2 * m) + (x + m)] = (uint8_t)d;
}

// Descriptor data cost (sgm.cc:76-84, initialisation sgm.cc:55): C[y][x][d] = ||fL[y][x] - fR[y][x-d]||_2
// for x >= d, FLT_MAX otherwise; pad disparities d >= D get 0.  The K differences are summed left to right
// in float32, one rounding per operation (Eigen's own reduction order is build dependent: parity unpinned).
// REPLICATE: recon::SGMStereo's variant (new_sgm/sgm_stereo.cc:158-181), which repeats the cost of
// the last valid disparity d = x where x < d instead of FLT_MAX (:174).
//
// A banded contraction: output pixel x needs the right descriptors x-D+1 .. x.  One workgroup of 1024 threads owns
// TX = 32 NI neighbouring pixels of one image row and all Dp = 32 NJ disparities; the TX left descriptors and the
// TX + Dp - 1 right descriptors go through LDS in chunks of KC = 32 components (row pitch 36 floats: lanes along d
// read neighbouring rows as 16-byte pieces without bank conflicts).  Thread (tx, td), td = lane-fast, keeps the
// NI x NJ running sums of pixels tx + 32 i and disparities td + 32 j: both strides are 32, so the right descriptor
// of voxel (i, j), pixel - disparity = (tx - td) + 32 (i - j), is one of only NI + NJ - 1 LDS rows, each read
// once per four components and used for every (i, j) on its diagonal -- 11 LDS reads of 16 bytes per 192 float
// operations at Dp = 128 (the round-2 first version read two per voxel and was bound by LDS bandwidth).  Each
// descriptor is read from HBM once per tile.  HBM bytes: 2 H W K 4 in (the right rows (TX + Dp - 1) / TX times,
// mostly from L2), H W Dp 4 out.
constexpr int DESC_KC = 32, DESC_KCP = 36, DESC_THREADS = 1024;
inline int desc_ni(int Dp) { return Dp <= 256 ? 4 : 2; }
inline int desc_tile_pixels(int Dp) { return 32 * desc_ni(Dp); }
inline size_t desc_lds_bytes(int Dp) { return (size_t)(2 * desc_tile_pixels(Dp) + Dp - 1) * DESC_KCP * sizeof(float); }

template <bool REPLICATE, int NI, int NJ>
__global__ __launch_bounds__(DESC_THREADS) __attribute__((amdgpu_waves_per_eu(NI * NJ <= 16 ? 8 : 4))) void descriptor_cost_kernel(const float *__restrict__ fl, const float *__restrict__ fr,
                                                                       int w, int K, int D, float *__restrict__ C)
{
    extern __shared__ __attribute__((aligned(16))) float desc_lds[];
    constexpr int TX = 32 * NI, Dp = 32 * NJ, NB = TX + Dp - 1;
    float *A = desc_lds;                   // [TX][KCP]   left descriptors x0 .. x0 + TX - 1
    float *B = desc_lds + TX * DESC_KCP;   // [NB][KCP]   right descriptors x0 - Dp + 1 .. x0 + TX - 1 (zeros left of the image)
    const int x0 = blockIdx.x * TX, y = blockIdx.y;
    const size_t row = (size_t)y * w;
    const int tid = threadIdx.x, td = tid & 31, tx = tid >> 5;
    float s[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
            s[i][j] = 0.f;
    const float *Ap = A + tx * DESC_KCP;
    const float *Bp = B + (tx - td + Dp - 1) * DESC_KCP; // the row of i - j = 0
    for (int k0 = 0; k0 < K; k0 += DESC_KC) {
        const int kc = K - k0 < DESC_KC ? K - k0 : DESC_KC;
        __syncthreads(); // the previous chunk has been consumed
        // fill: (TX + NB) rows of 32 components; components past K and rows left / right of the image hold 0 (a
        // zero difference leaves a sum unchanged).  16-byte pieces, all of a thread's loads in flight before its
        // first LDS store.
        if ((K & 3) == 0) {
            constexpr int PIECES = DESC_KC / 4, NPIECE = (TX + NB) * PIECES, MAXP = (NPIECE + DESC_THREADS - 1) / DESC_THREADS;
            float4 v[MAXP];
#pragma unroll
            for (int m = 0; m < MAXP; m++) {
                const int e = tid + DESC_THREADS * m, r = e / PIECES, k = (e % PIECES) * 4;
                v[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < NPIECE && k < kc) {
                    const int x = r < TX ? x0 + r : x0 - (Dp - 1) + (r - TX);
                    if (x >= 0 && x < w)
                        v[m] = *reinterpret_cast<const float4 *>((r < TX ? fl : fr) + (row + x) * K + k0 + k);
                }
            }
#pragma unroll
            for (int m = 0; m < MAXP; m++) {
                const int e = tid + DESC_THREADS * m, r = e / PIECES, k = (e % PIECES) * 4;
                if (e < NPIECE)
                    *reinterpret_cast<float4 *>((r < TX ? A + r * DESC_KCP : B + (r - TX) * DESC_KCP) + k) = v[m];
            }
        } else {
            for (int e = tid; e < (TX + NB) * DESC_KC; e += DESC_THREADS) {
                const int r = e / DESC_KC, k = e % DESC_KC;
                const int x = r < TX ? x0 + r : x0 - (Dp - 1) + (r - TX);
                float v = 0.f;
                if (x >= 0 && x < w && k < kc)
                    v = (r < TX ? fl : fr)[(row + x) * K + k0 + k];
                (r < TX ? A + r * DESC_KCP : B + (r - TX) * DESC_KCP)[k] = v;
            }
        }
        __syncthreads();
        for (int k = 0; k < kc; k += 4) { // (components past kc are zeros)
            float4 a[NI];
#pragma unroll
            for (int i = 0; i < NI; i++)
                a[i] = *reinterpret_cast<const float4 *>(Ap + 32 * i * DESC_KCP + k);
#pragma unroll
            for (int dl = -(NJ - 1); dl <= NI - 1; dl++) { // the diagonal i - j = dl shares one right descriptor
                // two diagonals' reads in flight at a time, not all of them: 64 VGPRs let two workgroups share a CU,
                // one filling its chunk while the other computes
                if ((dl + NJ - 1) % 2 == 0)
                    __builtin_amdgcn_sched_barrier(0);
                const float4 b = *reinterpret_cast<const float4 *>(Bp + 32 * dl * DESC_KCP + k);
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const int j = i - dl;
                    if (j < 0 || j >= NJ)
                        continue;
                    // left to right over the components, one rounding per operation (no contraction)
                    float df = a[i].x - b.x;
                    s[i][j] += df * df;
                    df = a[i].y - b.y;
                    s[i][j] += df * df;
                    df = a[i].z - b.z;
                    s[i][j] += df * df;
                    df = a[i].w - b.w;
                    s[i][j] += df * df;
                }
            }
        }
    }
    // REPLICATE: the voxels with d > x take the cost of d = x, which another thread of this workgroup holds
    float *diag = desc_lds; // [TX]
    if (REPLICATE && x0 < D) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
                if (x0 + tx + 32 * i == td + 32 * j)
                    diag[tx + 32 * i] = __builtin_sqrtf(s[i][j]);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int xl = tx + 32 * i, x = x0 + xl;
        if (x >= w)
            continue;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int d = td + 32 * j;
            float o;
            if (d >= D)
                o = 0.f;
            else if (d > x)
                o = REPLICATE ? diag[xl] : 3.402823466e+38f;
            else
                o = __builtin_sqrtf(s[i][j]);
            C[(row + x) * Dp + d] = o;
        }
    }
}

__global__ __launch_bounds__(256) void hamming_matrix_kernel(uint16_t *__restrict__ dis, const uint8_t *__restrict__ a,
                                                             const uint8_t *__restrict__ b, int na, int nb, int ncodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= na || j >= nb)
        return;
    const uint8_t *pa = a + (size_t)i * ncodes, *pb = b + (size_t)j * ncodes;
    u32 s = 0;
    int k = 0;
    if ((ncodes & 3) == 0)
        for (; k < ncodes; k += 4)
            s += __builtin_popcount(*reinterpret_cast<const u32 *>(pa + k) ^ *reinterpret_cast<const u32 *>(pb + k));
    for (; k < ncodes; k++)
        s += __builtin_popcount((u32)(pa[k] ^ pb[k]));
    dis[(size_t)j * na + i] = (uint16_t)s;
}

} // namespace sgmx
