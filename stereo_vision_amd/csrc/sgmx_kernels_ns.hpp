// sgmx_kernels_ns.hpp -- kernels of the recon::SGMStereo pipeline (reconstruction/base/new_sgm/
// sgm_stereo.cc): right data-cost volume, plain WTA with its sub-pixel estimate, speckle filter,
// left/right states, interpolation, 16-bit scaling and the 5x5 median.  The eight path sweeps
// themselves are aggregate_f32_kernel (sgmx_kernels.hpp) started from an all-zero border ring,
// which is the textbook start L = C this pipeline uses (sgm_stereo.cc:333).
#pragma once
#include "sgmx_kernels.hpp"

namespace sgmx {

// sgm_stereo.cc:208-246 ComputeRightCostImage(): right[y][x][d] = left[y][x+d][d] while x + d < w,
// beyond that the last valid value left[y][w-1][w-1-x] repeats (:235-244).  Both volumes have
// pitch Dp; pad disparities get 0.  One thread per output voxel.
__global__ __launch_bounds__(256) void ns_right_cost_kernel(const float *__restrict__ left, int w, int D, int Dp,
                                                            float *__restrict__ right, size_t total)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total)
        return;
    const size_t pix = t / Dp;
    const int d = (int)(t - pix * Dp);
    const int x = (int)(pix % (size_t)w);
    float v = 0.f;
    if (d < D) {
        const int dd = x + d < w ? d : w - 1 - x;
        v = left[(pix + dd) * Dp + dd];
    }
    right[t] = v;
}

// sgm_stereo.cc:362-402: first-minimum argmin over d of the summed costs and, for the left image,
// bestDisparity + (float)(r - l) / (c - l or c - r) / 2.0 + 0.5 evaluated as the reference's types
// dictate (float quotient, then double), cast to float; d = 0 and d = D-1 stay unrefined.
// One workgroup per row, one wave per pixel.
template <int R>
__global__ __launch_bounds__(256) void ns_wta_kernel(const float *__restrict__ S, int w, int h, int D,
                                                     int *__restrict__ disp, float *__restrict__ subpix)
{
    constexpr int N = 2 * R;
    constexpr int Dp = 64 * N;
    typedef float fvec __attribute__((ext_vector_type(N)));
    const int y = blockIdx.x;
    const size_t row = (size_t)blockIdx.y * h + y;
    const float *Srow = S + row * (size_t)w * Dp;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int x = wave; x < w; x += 4) {
        const fvec s = *reinterpret_cast<const fvec *>(Srow + (size_t)x * Dp + lane * N);
        unsigned long long best = ~0ull;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int d = lane * N + j;
            if (d < D) {
                const unsigned long long key = ((unsigned long long)float_key(s[j]) << 32) | (u32)d;
                best = key < best ? key : best;
            }
        }
        const u32 hi = wave_min_u32((u32)(best >> 32));
        const u32 lo = wave_min_u32((u32)(best >> 32) == hi ? (u32)best : 0xFFFFFFFFu);
        if (lane == 0) {
            const int bd = (int)lo;
            disp[row * w + x] = bd;
            if (subpix) {
                float o = (float)bd;
                if (bd > 0 && bd < D - 1) {
                    const float *v = Srow + (size_t)x * Dp + bd;
                    const float l = v[-1], c = v[0], r = v[1];
                    const float q = r < l ? (r - l) / (c - l) : (r - l) / (c - r);
                    o = (float)((double)bd + (double)q / 2.0 + 0.5);
                }
                subpix[row * w + x] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Speckle filter (sgm_stereo.cc:416-488): 4-connected regions of non-zero pixels whose neighbouring
// values differ by at most max_diff; regions of at most max_size pixels are zeroed.  The reference
// flood-fills in raster order; the regions are the connected components of a symmetric relation,
// so a union-find labelling gives the same sets: horizontal runs first (one wave per row), then
// vertical unions by atomicMin on the roots, then component sizes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool ns_linked(int a, int b, int max_diff) { return a != 0 && b != 0 && abs(a - b) <= max_diff; }

// label = index (within the image) of the first pixel of the pixel's horizontal run, -1 for zero pixels.
__global__ __launch_bounds__(256) void ns_ccl_rows_kernel(const int *__restrict__ img, int w, int h, int max_diff,
                                                          int *__restrict__ label)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h)
        return;
    const size_t base = ((size_t)blockIdx.y * h + y) * w;
    int carry = 0; // run start when the chunk holds no start at or before the lane
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const int v = x < w ? img[base + x] : 0;
        const int vl = (x > 0 && x < w) ? img[base + x - 1] : 0;
        const bool start = !ns_linked(vl, v, max_diff);
        const unsigned long long m = __ballot(start);
        const unsigned long long le = m & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
        const int s = le ? x0 + 63 - __builtin_clzll(le) : carry;
        if (x < w)
            label[base + x] = v != 0 ? y * w + s : -1;
        if (m)
            carry = x0 + 63 - __builtin_clzll(m);
    }
}

__device__ __forceinline__ int ns_find(const int *L, int i)
{
    int p;
    while ((p = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != i)
        i = p;
    return i;
}

__device__ __forceinline__ void ns_unite(int *L, int a, int b)
{
    for (;;) {
        a = ns_find(L, a);
        b = ns_find(L, b);
        if (a == b)
            return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(L + a, b); // a was a root iff old == a
        if (old == a)
            return;
        a = old; // a had been linked to `old` meanwhile: unite that with b as well
    }
}

__global__ __launch_bounds__(256) void ns_ccl_cols_kernel(const int *__restrict__ img, int w, int h, int max_diff,
                                                          int *__restrict__ label)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h - 1)
        return;
    const size_t base = (size_t)blockIdx.z * w * h;
    const int i = y * w + x;
    if (ns_linked(img[base + i], img[base + i + w], max_diff))
        ns_unite(label + base, label[base + i], label[base + i + w]);
}

__global__ __launch_bounds__(256) void ns_ccl_count_kernel(int *__restrict__ label, int npix, int *__restrict__ size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix)
        return;
    const size_t base = (size_t)blockIdx.y * npix;
    if (label[base + i] < 0)
        return;
    const int r = ns_find(label + base, i);
    atomicAdd(size + base + r, 1);
}

__global__ __launch_bounds__(256) void ns_ccl_apply_kernel(int *__restrict__ img, const int *__restrict__ label, int npix,
                                                           const int *__restrict__ size, int max_size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix)
        return;
    const size_t base = (size_t)blockIdx.y * npix;
    if (label[base + i] < 0)
        return;
    const int r = ns_find(label + base, i);
    if (size[base + r] <= max_size)
        img[base + i] = 0;
}

// sgm_stereo.cc:490-531 EnforceLeftRightConsistency: 'c' / 'm' / 'o' per pixel.
__global__ __launch_bounds__(256) void ns_lr_kernel(const int *__restrict__ left, const int *__restrict__ right, int w, int h,
                                                    int D, int thr, unsigned char *__restrict__ states)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w)
        return;
    const size_t row = ((size_t)blockIdx.z * h + y) * w;
    const int ld = left[row + x];
    unsigned char st = 'o';
    if (ld <= x) {
        if (abs(ld - right[row + x - ld]) <= thr) {
            st = 'c';
        } else {
            const int dmax = x < D ? x : D; // inclusive (:512)
            for (int d = 0; d <= dmax; d++)
                if (d != ld && abs(ld - right[row + x - d]) <= thr) {
                    st = 'm';
                    break;
                }
        }
    }
    states[row + x] = st;
}

// sgm_stereo.cc:534-557 InterpolateDisparities: every 'o' pixel with a 'c' pixel somewhere to its
// left in the row takes that pixel's value and becomes 'c' (the reference's fill loop stops after
// the pixel itself, :548; filled pixels then serve the next ones with the same value, 'm' pixels
// are skipped).  One wave per row, 64 pixels per step.
__global__ __launch_bounds__(256) void ns_interpolate_kernel(unsigned char *__restrict__ states, float *__restrict__ disp,
                                                             int w, int h)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h)
        return;
    const size_t base = ((size_t)blockIdx.y * h + y) * w;
    bool have = false;
    float carry = 0.f;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const unsigned char st = x < w ? states[base + x] : (unsigned char)'m';
        const float v = x < w ? disp[base + x] : 0.f;
        const unsigned long long m = __ballot(st == 'c');
        const unsigned long long lt = m & ((1ull << lane) - 1);
        const int src = lt ? 63 - __builtin_clzll(lt) : 0;
        const float from_chunk = __shfl(v, src);
        if (st == 'o' && (lt || have)) {
            disp[base + x] = lt ? from_chunk : carry;
            states[base + x] = 'c';
        }
        if (m) {
            carry = __shfl(v, 63 - __builtin_clzll(m));
            have = true;
        }
    }
}

// sgm_stereo.cc:79-80: (uint16)(float)std::round(subpix * 256.0), the product in double.
__global__ __launch_bounds__(256) void ns_scale_kernel(const float *__restrict__ subpix, size_t n, uint16_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float scaled = (float)__builtin_round((double)subpix[i] * 256.0);
    out[i] = cast_u16_like_x86((double)scaled); // past 65535 (D > 256) the reference's conversion is undefined: low 16 bits
}

// cv::medianBlur(img, img, 5) on CV_16U (sgm_stereo.cc:85): median of the 25 replicate-bordered
// neighbours.  One thread per pixel; odd-even transposition sort in registers.
__global__ __launch_bounds__(256) void median5_kernel(const uint16_t *__restrict__ in, int W, int H,
                                                      uint16_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W)
        return;
    const uint16_t *img = in + (size_t)blockIdx.z * W * H;
    u32 v[25];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int yy = y + j - 2;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
        for (int i = 0; i < 5; i++) {
            int xx = x + i - 2;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            v[j * 5 + i] = img[(size_t)yy * W + xx];
        }
    }
#pragma unroll
    for (int pass = 0; pass < 25; pass++) {
#pragma unroll
        for (int i = pass & 1; i + 1 < 25; i += 2)
            cswap(v[i], v[i + 1]);
    }
    out[((size_t)blockIdx.z * H + y) * W + x] = (uint16_t)v[12];
}

} // namespace sgmx
