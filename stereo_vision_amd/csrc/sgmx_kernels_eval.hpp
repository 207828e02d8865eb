// sgmx_kernels_eval.hpp -- the KITTI stereo evaluation of reconstruction/evaluation on disparity maps
// in the on-disk format the SGM mains write (16-bit, value = 256 * d, 0 = invalid; io_disp.h:220-243):
// background interpolation (io_disp.h:108-186) and the outlier / average error statistics
// (evaluate_stereo.cc:21-119).  Disparities are multiples of 1/256, so every per-pixel value is exact;
// the only order-dependent arithmetic is the reference's float sum of the absolute errors, which walks
// the image column by column -- reproduced here by one lane per sum adding in that same order.
#pragma once
#include "sgmx_kernels.hpp"

namespace sgmx {

__device__ __forceinline__ float kitti_value(uint16_t v) { return v == 0 ? -1.f : (float)v / 256.0f; } // exact

// Row part of interpolateBackground (io_disp.h:111-163), one wave per row: a gap between two valid
// pixels takes the smaller of them, a gap at the start / end of the row repeats the first / last
// valid pixel, a row without valid pixels stays invalid.  Forward sweep leaves the nearest valid value
// to the left (or -1) in `ipol`, the backward sweep combines it with the nearest one to the right.
__global__ __launch_bounds__(256) void kitti_rows_kernel(const uint16_t *__restrict__ png, int w, int h,
                                                         float *__restrict__ ipol)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h)
        return;
    const size_t base = ((size_t)blockIdx.y * h + y) * w;
    float carry = -1.f;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const float v = x < w ? kitti_value(png[base + x]) : -1.f;
        const unsigned long long m = __ballot(v >= 0.f);
        const unsigned long long lt = m & ((1ull << lane) - 1);
        const float from_chunk = __shfl(v, lt ? 63 - __builtin_clzll(lt) : 0);
        if (x < w)
            ipol[base + x] = v >= 0.f ? v : (lt ? from_chunk : carry);
        if (m)
            carry = __shfl(v, 63 - __builtin_clzll(m));
    }
    carry = -1.f;
    for (int x0 = ((w - 1) / 64) * 64; x0 >= 0; x0 -= 64) {
        const int x = x0 + lane;
        const float v = x < w ? kitti_value(png[base + x]) : -1.f;
        const unsigned long long m = __ballot(v >= 0.f);
        const unsigned long long gt = lane == 63 ? 0ull : (m & ~((2ull << lane) - 1));
        const float from_chunk = __shfl(v, gt ? __builtin_ctzll(gt) : 0);
        const float right = gt ? from_chunk : carry;
        if (x < w && v < 0.f && right >= 0.f) {
            const float left = ipol[base + x];
            ipol[base + x] = left >= 0.f ? (right < left ? right : left) : right;
        }
        if (m)
            carry = __shfl(v, __builtin_ctzll(m));
    }
}

// Column part (io_disp.h:165-186), one thread per column: pixels above the first valid one of the
// column take its value, pixels below the last valid one take that one's; gaps in between stay.
__global__ __launch_bounds__(256) void kitti_cols_kernel(float *__restrict__ ipol, int w, int h)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w)
        return;
    float *col = ipol + (size_t)blockIdx.y * w * h + x;
    int first = 0;
    while (first < h && col[(size_t)first * w] < 0.f)
        first++;
    if (first == h)
        return;
    const float top = col[(size_t)first * w];
    for (int v = 0; v < first; v++)
        col[(size_t)v * w] = top;
    int last = h - 1;
    while (col[(size_t)last * w] < 0.f)
        last--;
    const float bottom = col[(size_t)last * w];
    for (int v = last + 1; v < h; v++)
        col[(size_t)v * w] = bottom;
}

// D_ipol.write (io_disp.h:234-243): (uint16)max(d * 256.0, 1.0) for valid pixels, 0 otherwise.
__global__ __launch_bounds__(256) void kitti_write_kernel(const float *__restrict__ d, size_t n, uint16_t *__restrict__ png)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double s = (double)d[i] * 256.0;
    png[i] = d[i] >= 0.f ? (uint16_t)(s > 1.0 ? s : 1.0) : (uint16_t)0;
}

// disparityErrorsOutlier + disparityErrorsAverage (evaluate_stereo.cc:21-119) against the two
// ground-truth maps of one image; one workgroup per image.  out[26] = noc_out[11], occ_out[11],
// noc_avg[2], occ_avg[2]; out[0] = -1 flags a ground truth without valid pixels (the reference throws).
constexpr int KITTI_TILE = 3072; // pixels per LDS tile of the ordered sums (4 addends each)

__global__ __launch_bounds__(256) void kitti_errors_kernel(const uint16_t *__restrict__ disp, const float *__restrict__ ipol,
                                                           const uint16_t *__restrict__ gt_noc,
                                                           const uint16_t *__restrict__ gt_occ, int w, int h,
                                                           float *__restrict__ out)
{
    __shared__ float4 tile[KITTI_TILE];
    __shared__ int cnt[2][12]; // per ground truth: 5 x (all, result), num_pixels, num_pixels_result
    const size_t base = (size_t)blockIdx.x * w * h;
    const int npix = w * h;
    if (threadIdx.x < 24)
        (&cnt[0][0])[threadIdx.x] = 0;
    __syncthreads();
    {
        int c[2][12] = {{0}};
        for (int i = threadIdx.x; i < npix; i += blockDim.x) {
            const float ip = ipol[base + i];
            const bool res = disp[base + i] != 0;
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const float gv = kitti_value(g == 0 ? gt_noc[base + i] : gt_occ[base + i]);
                if (gv >= 0.f) {
                    const float e = __builtin_fabsf(gv - ip);
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        const int over = e > (float)(k + 1);
                        c[g][2 * k] += over;
                        c[g][2 * k + 1] += over & (int)res;
                    }
                    c[g][10]++;
                    c[g][11] += res;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int k = 0; k < 12; k++)
                if (c[g][k])
                    atomicAdd(&cnt[g][k], c[g][k]);
    }
    // the four float sums in the reference's pixel order: u outer, v inner
    float sum = 0.f;
    const int cols_per_tile = KITTI_TILE / h > 0 ? KITTI_TILE / h : 1;
    const int rows_per_tile = KITTI_TILE / h > 0 ? h : KITTI_TILE; // a column taller than a tile goes in pieces
    for (int u0 = 0; u0 < w; u0 += cols_per_tile) {
        const int nc = w - u0 < cols_per_tile ? w - u0 : cols_per_tile;
        for (int v0 = 0; v0 < h; v0 += rows_per_tile) {
            const int nr = h - v0 < rows_per_tile ? h - v0 : rows_per_tile;
            __syncthreads();
            for (int t = threadIdx.x; t < nc * nr; t += blockDim.x) {
                const int v = v0 + t / nc, u = u0 + t % nc; // neighbouring threads read neighbouring pixels
                const size_t i = base + (size_t)v * w + u;
                const float ip = ipol[i];
                const bool res = disp[i] != 0;
                const float gn = kitti_value(gt_noc[i]), go = kitti_value(gt_occ[i]);
                const float en = gn >= 0.f ? __builtin_fabsf(gn - ip) : 0.f, eo = go >= 0.f ? __builtin_fabsf(go - ip) : 0.f;
                tile[(u - u0) * nr + (v - v0)] = make_float4(en, res ? en : 0.f, eo, res ? eo : 0.f); // + 0 leaves a sum as it is
            }
            __syncthreads();
            if (threadIdx.x < 4) {
                const float *tf = reinterpret_cast<const float *>(tile) + threadIdx.x;
                for (int k = 0; k < nc * nr; k++)
                    sum += tf[4 * k];
            }
        }
    }
    __syncthreads();
    float *o = out + (size_t)blockIdx.x * 26;
    if (threadIdx.x < 2) { // outlier shares of ground truth g
        const int g = threadIdx.x;
        const float np = fmaxf((float)cnt[g][10], 1.0f), nr = fmaxf((float)cnt[g][11], 1.0f);
        for (int k = 0; k < 5; k++) {
            o[g * 11 + 2 * k] = (float)cnt[g][2 * k] / np;
            o[g * 11 + 2 * k + 1] = cnt[g][11] > 0 ? (float)cnt[g][2 * k + 1] / nr : (float)cnt[g][2 * k + 1];
        }
        o[g * 11 + 10] = (float)cnt[g][11] / np;
        if (cnt[g][10] == 0)
            o[g * 11] = -1.f;
    }
    __syncthreads();
    if (threadIdx.x < 4) { // averages: lanes 0..3 = (noc, all), (noc, result), (occ, all), (occ, result)
        const int g = threadIdx.x >> 1;
        const float n = fmaxf((float)cnt[g][10 + (threadIdx.x & 1)], 1.0f);
        o[22 + threadIdx.x] = sum / n;
    }
}

// CalculateDepthImage (reconstruction/main/create_rgbd_dataset/main.cc:12-27) on the 16-bit disparity
// image: d = v / 256 (the CV_64F conversion at :93); d >= 1 gives round(100 * f * b / d) centimetres,
// everything else 0; *overflow is set when a depth does not fit 16 bits (the reference throws).
__global__ __launch_bounds__(256) void depth_from_disparity_kernel(const uint16_t *__restrict__ disp, size_t n, double f,
                                                                   double b, uint16_t *__restrict__ depth,
                                                                   int *__restrict__ overflow)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double d = (double)disp[i] * (1.0 / 256.0);
    uint16_t o = 0;
    if (d >= 1.0) {
        const double z = 100.0 * f * b / d;
        if (z >= 65536)
            atomicOr(overflow, 1);
        else
            o = (uint16_t)__builtin_round(z);
    }
    depth[i] = o;
}

} // namespace sgmx
