// sgmx.hip -- context management and the C ABI (include/sgmx.h) of the SGM engine.
// HIP only: there is no CPU fallback anywhere in this library; sgmx_create fails
// with SGMX_ERR_NO_DEVICE when no GPU is usable.
#include "../../include/sgmx.h"
#include "sgmx_kernels.hpp"
#include "sgmx_kernels_v2.hpp"
#include "sgmx_kernels_ns.hpp"
#include "sgmx_kernels_eval.hpp"
#include "sgmx_kernels_ham.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace sgmx;

namespace {

thread_local std::string g_create_error;

const int DIRS8[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}}; // stereo_sgm.cc:120-122
const int DIRS4[4][2] = {{-1, 0}, {0, -1}, {0, 1}, {1, 0}};
// recon::SGMStereo (new_sgm/sgm_stereo.cc:283-316,357): the order in which its two raster passes add the paths
// into sum_cost -- previous pixel of the row, previous row's x-step / x / x+step, forwards then backwards.
const int DIRS_NS[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};

struct ProfRec {
    int cls;
    int dir; // direction sweeps: index of the direction in the schedule, else -1
    hipEvent_t a, b;
    bool owns_a; // false: a is the previous record's b (SGMX_PROFILE_SWEEPS chains the sweeps of a launch group)
};

// Device buffers and geometry of one census launch group in the volume-free formulation.  The context's own
// scratch is one such view (ensure_scratch); sgmx_compute_census_u8_dev_noalloc builds another over the static
// array below.
struct FastView {
    int W = 0, H = 0, w = 0, h = 0, m = 0, wp = 0;
    int cap = 0;                 // pairs the view holds (the E volumes of two directions lie cap pairs apart)
    size_t bytes_E = 0;          // one pair of one direction
    size_t bytes_BL = 0;         // one pair's border ring
    u32 *cl = nullptr, *cr = nullptr;
    uint8_t *E = nullptr;
    uint16_t *BL = nullptr, *SB = nullptr, *raw = nullptr;
    uint16_t *BL2 = nullptr;     // second border ring (directions that run side by side, Agg2Args::fuse), or null
};

} // namespace

// Scratch of the entry point that allocates nothing (sgmx_compute_census_u8_dev_noalloc): one array per GPU,
// part of the code object, held by every device the library runs on until the process ends.  In host code this
// name is only a handle for hipGetSymbolAddress -- never a device address; the per-device addresses live in
// g_scratch_ptr.
#define SGMX_NOALLOC_SCRATCH_BYTES ((size_t)64 << 20)
static __device__ __align__(256) unsigned char sgmx_noalloc_scratch[SGMX_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[MAX_DEVICES] = {nullptr};

// Address of this device's copy of the static scratch; looked up once per device, with that device current.
hipError_t noalloc_scratch_for_device(int device, unsigned char **out)
{
    if (device < 0 || device >= MAX_DEVICES)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess)
            return e;
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(sgmx_noalloc_scratch));
        if (e != hipSuccess)
            return e;
        g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
} // namespace

enum { HAM_A = 0, HAM_B, HAM_CNT, HAM_IDX, HAM_H, HAM_DIS, HAM_REC, HAM_SLOTS };

struct sgmx_ctx {
    sgmx_params prm;
    int R = 1;  // packed uint16 pairs per lane (census) / float pairs per lane (float path)
    int Dp = 128;
    std::string err;

    // geometry the scratch is currently sized for
    int W = 0, H = 0;          // census: full image; float: volume size
    int w = 0, h = 0, m = 0;   // cropped
    int mode = 0;              // 0 none, 1 census, 2 float, 3 float SGMStereo pipeline
    bool fast = false;         // census path without C / S volumes (sgmx_kernels_v2.hpp): P2 <= 255
    int RP = 4, DpF = 128;     // fast path: packed pairs per lane, padded disparity count 32 * RP
    int wp = 0;                // row pitch of the padded right census image (fast path)
    int last_n = 0;            // pairs in the last launch group
    int last_first = 0;        // index of that group's first pair within the call that ran it
    int num_cus = 256;
    unsigned char *noalloc_scratch = nullptr; // this device's copy of sgmx_noalloc_scratch
    // The scratch below is shared by every call on this context.  ev_scratch is recorded behind the last
    // launch group; a call on another stream waits for it before it touches the scratch.
    hipEvent_t ev_scratch = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;

    // device scratch (sized for max_batch pairs)
    uint8_t *d_left = nullptr, *d_right = nullptr; // staged inputs for the host entry points
    uint8_t *d_left2 = nullptr, *d_right2 = nullptr; // second staging set of the pipelined batch entry point
    // sgmx_set_preblur3: the callers' 3x3 Gaussian pre-filter fused in front of the census transform
    bool preblur = false;
    int blur_k0 = 0, blur_k1 = 0; // the kernel's centre and side taps, 8 fractional bits
    uint8_t *d_blur = nullptr;    // blurred left and right images of one launch group
    size_t blur_cap = 0;
    unsigned char *h_stage = nullptr; // page-locked staging of the one-pair host entry point: left, right, disparity
    size_t h_stage_bytes = 0;
    uint16_t *d_out2 = nullptr;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    u32 *d_cl = nullptr, *d_cr = nullptr;
    void *d_C = nullptr;   // uint8 or float, pitch Dp
    void *d_S = nullptr;   // uint16 or float, pitch Dp
    void *d_BL = nullptr;  // border ring
    uint8_t *d_E = nullptr;     // fast path: num_paths byte volumes, [k][max_batch][h][w][Dp]
    uint16_t *d_SB = nullptr;   // fast path: border accumulator, same shape as d_BL
    uint16_t *d_BL2 = nullptr;  // fast path: second border ring (two directions side by side in small launch groups)
    hipStream_t aux_stream = nullptr; // the second of two directions that run side by side
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    size_t bytes_E = 0;         // one pair of one direction
    uint16_t *d_raw = nullptr, *d_out = nullptr;
    size_t bytes_C = 0, bytes_S = 0, bytes_BL = 0;
    // SGMStereo pipeline (mode 3): right cost volume, integer disparities, labels / region sizes of the
    // speckle filter, sub-pixel disparities, left/right states
    float *d_C2 = nullptr, *d_sub = nullptr;
    int *d_ld = nullptr, *d_rd = nullptr, *d_label = nullptr, *d_size = nullptr;
    uint8_t *d_states = nullptr;

    // Grow-only device buffers of the Hamming-library entry points (signatures, counts, matches, the distance
    // matrix): kept from call to call instead of a hipMalloc / hipFree per buffer and call; freed with the context.
    void *ham_buf[HAM_SLOTS] = {nullptr};
    size_t ham_cap[HAM_SLOTS] = {0};

    hipStream_t own_stream = nullptr;
    int profiling = 0; // 0, SGMX_PROFILE_ALL or SGMX_PROFILE_SWEEPS
    std::vector<ProfRec> prof;
    double prof_ms[SGMX_K_COUNT] = {0};
    long prof_n[SGMX_K_COUNT] = {0};
    double prof_dir_ms[16] = {0}; // the sweeps again, per direction of the schedule
    long prof_dir_n[16] = {0};
};

namespace {

int fail(sgmx_ctx *c, int code, const std::string &msg)
{
    if (c)
        c->err = msg;
    else
        g_create_error = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                                                \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return fail((c), e_ == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP,                                  \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                            \
    } while (0)

void free_scratch(sgmx_ctx *c)
{
    void *ptrs[] = {c->d_left, c->d_right, c->d_cl,  c->d_cr, c->d_C,  c->d_S,     c->d_BL,   c->d_raw,   c->d_out,
                    c->d_E,    c->d_SB,    c->d_C2,  c->d_sub, c->d_ld, c->d_rd,   c->d_label, c->d_size, c->d_states,
                    c->d_left2, c->d_right2, c->d_out2, c->d_blur, c->d_BL2};
    c->d_BL2 = nullptr;
    c->d_blur = nullptr;
    c->blur_cap = 0;
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    if (c->h_stage)
        (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    c->d_left = c->d_right = nullptr;
    c->d_left2 = c->d_right2 = nullptr;
    c->d_out2 = nullptr;
    c->d_cl = c->d_cr = nullptr;
    c->d_C = c->d_S = c->d_BL = nullptr;
    c->d_raw = c->d_out = nullptr;
    c->d_E = nullptr;
    c->d_SB = nullptr;
    c->d_C2 = c->d_sub = nullptr;
    c->d_ld = c->d_rd = c->d_label = c->d_size = nullptr;
    c->d_states = nullptr;
    c->mode = 0;
    c->W = c->H = 0;
}

// Order the scratch between streams: a call on a stream other than the last user's waits for that user's
// last launch group (sgmx.h: "_dev" entry points).
int scratch_acquire(sgmx_ctx *c, hipStream_t st)
{
    if (c->scratch_used && st != c->scratch_stream)
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_scratch, 0));
    return SGMX_OK;
}
int scratch_release(sgmx_ctx *c, hipStream_t st)
{
    HIPCHK(c, hipEventRecord(c->ev_scratch, st));
    c->scratch_stream = st;
    c->scratch_used = true;
    return SGMX_OK;
}

// (Re)allocate scratch for `mode` at image/volume size W x H; `st` is the stream of the call that needs it
// (the buffers' initial memsets are ordered on it).
int ensure_scratch(sgmx_ctx *c, int mode, int W, int H, hipStream_t st)
{
    if (c->mode == mode && c->W == W && c->H == H)
        return scratch_acquire(c, st);
    if (c->scratch_used) // the previous geometry's buffers may still be in use on another stream
        HIPCHK(c, hipEventSynchronize(c->ev_scratch));
    c->scratch_used = false;
    free_scratch(c);
    const int B = c->prm.max_batch;
    const int m = mode == 1 ? (c->prm.window_sz - 1) / 2 : 0;
    const int w = W - 2 * m, h = H - 2 * m;
    const size_t npix = (size_t)w * h, nfull = (size_t)W * H;
    const size_t esz_c = mode == 1 ? 1 : 4, esz_s = mode == 1 ? 2 : 4;
    c->bytes_C = npix * c->Dp * esz_c;
    c->bytes_S = npix * c->Dp * esz_s;
    c->bytes_BL = (size_t)(2 * w + 2 * h) * c->Dp * esz_s;
    if (mode == 1 && c->fast)
        c->bytes_BL = (size_t)(2 * w + 2 * h) * c->DpF * 2;
    const bool fast = mode == 1 && c->fast;
    c->wp = fast ? w + c->DpF : w;
    if (mode == 1) {
        HIPCHK(c, hipMalloc((void **)&c->d_left, nfull * B));
        HIPCHK(c, hipMalloc((void **)&c->d_right, nfull * B));
        HIPCHK(c, hipMalloc((void **)&c->d_left2, nfull * B));
        HIPCHK(c, hipMalloc((void **)&c->d_right2, nfull * B));
        HIPCHK(c, hipMalloc((void **)&c->d_out2, nfull * 2 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_cl, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_cr, (size_t)c->wp * h * 4 * B));
        HIPCHK(c, hipMemsetAsync(c->d_cr, 0, (size_t)c->wp * h * 4 * B, st)); // the padding columns are never written again
    }
    HIPCHK(c, hipMalloc(&c->d_BL, c->bytes_BL * B));
    if (fast) {
        c->bytes_E = npix * c->DpF;
        HIPCHK(c, hipMalloc((void **)&c->d_E, c->bytes_E * B * c->prm.num_paths));
        HIPCHK(c, hipMalloc((void **)&c->d_SB, c->bytes_BL * B));
        HIPCHK(c, hipMalloc((void **)&c->d_BL2, c->bytes_BL * B));
    } else {
        HIPCHK(c, hipMalloc(&c->d_C, c->bytes_C * B));
        HIPCHK(c, hipMalloc(&c->d_S, c->bytes_S * B));
    }
    if (mode == 3) {
        HIPCHK(c, hipMalloc((void **)&c->d_C2, c->bytes_C * B));
        HIPCHK(c, hipMalloc((void **)&c->d_sub, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_ld, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_rd, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_label, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_size, npix * 4 * B));
        HIPCHK(c, hipMalloc((void **)&c->d_states, npix * B));
    }
    HIPCHK(c, hipMalloc((void **)&c->d_raw, nfull * 2 * B));
    HIPCHK(c, hipMalloc((void **)&c->d_out, nfull * 2 * B));
    // the zero border of the disparity image (stereo_sgm.h:280) is written once here;
    // the WTA kernel only ever touches the interior
    HIPCHK(c, hipMemsetAsync(c->d_raw, 0, nfull * 2 * B, st));
    c->mode = mode;
    c->W = W;
    c->H = H;
    c->w = w;
    c->h = h;
    c->m = m;
    return SGMX_OK;
}

struct ProfScope {
    sgmx_ctx *c;
    hipStream_t st;
    ProfRec r;
    bool on;
    ProfScope(sgmx_ctx *c_, int cls, hipStream_t s, int dir = -1) : c(c_), st(s), on(c_->profiling == SGMX_PROFILE_ALL)
    {
        if (on) {
            r.cls = cls;
            r.dir = dir;
            r.owns_a = true;
            on = hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess;
            if (on)
                (void)hipEventRecord(r.a, st);
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(r.b, st);
            c->prof.push_back(r);
        }
    }
};

template <int R, bool FIRST, bool HAS_PAD>
void launch_agg_u16(const AggArgs &a, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((aggregate_u16_kernel<R, FIRST, HAS_PAD, (R == 4 ? 4 : 8)>), grid, dim3(256), 0, st, a);
}
template <int R, bool FIRST, bool HAS_PAD>
void launch_agg_f32(const AggArgs &a, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((aggregate_f32_kernel<R, FIRST, HAS_PAD, (R == 4 ? 2 : (R == 2 ? 4 : 8))>), grid, dim3(256), 0,
                       st, a);
}

template <bool F32> void launch_agg(int R, bool first, bool pad, const AggArgs &a, dim3 grid, hipStream_t st)
{
#define SGMX_DISPATCH(RR)                                                                                              \
    if (first) {                                                                                                       \
        if (pad)                                                                                                       \
            F32 ? launch_agg_f32<RR, true, true>(a, grid, st) : launch_agg_u16<RR, true, true>(a, grid, st);            \
        else                                                                                                           \
            F32 ? launch_agg_f32<RR, true, false>(a, grid, st) : launch_agg_u16<RR, true, false>(a, grid, st);          \
    } else {                                                                                                           \
        if (pad)                                                                                                       \
            F32 ? launch_agg_f32<RR, false, true>(a, grid, st) : launch_agg_u16<RR, false, true>(a, grid, st);          \
        else                                                                                                           \
            F32 ? launch_agg_f32<RR, false, false>(a, grid, st) : launch_agg_u16<RR, false, false>(a, grid, st);        \
    }
    if (R == 1) {
        SGMX_DISPATCH(1)
    } else if (R == 2) {
        SGMX_DISPATCH(2)
    } else {
        SGMX_DISPATCH(4)
    }
#undef SGMX_DISPATCH
}

// The 8 (or 4) ordered direction sweeps (materialised formulation) over the n cost volumes at Cvol.
// `order` != nullptr: eight sweeps in that order, each from an all-zero border ring, i.e. without the
// reference's carry-over between directions (the textbook start L = C of new_sgm/sgm_stereo.cc:333).
int run_aggregation(sgmx_ctx *c, int n, bool f32, hipStream_t st, const void *Cvol, const int (*order)[2] = nullptr)
{
    const int np = order ? 8 : c->prm.num_paths;
    HIPCHK(c, hipMemsetAsync(c->d_BL, 0, c->bytes_BL * n, st)); // path buffer starts at zero (stereo_sgm.cc:46)
    for (int k = 0; k < np; k++) {
        if (order && k > 0)
            HIPCHK(c, hipMemsetAsync(c->d_BL, 0, c->bytes_BL * n, st));
        AggArgs a;
        a.C = Cvol;
        a.S = c->d_S;
        a.BL = c->d_BL;
        a.w = c->w;
        a.h = c->h;
        a.D = c->prm.disp_range;
        a.dx = order ? order[k][0] : (np == 8 ? DIRS8[k][0] : DIRS4[k][0]);
        a.dy = order ? order[k][1] : (np == 8 ? DIRS8[k][1] : DIRS4[k][1]);
        a.P1 = c->prm.penalty1;
        a.P2 = c->prm.penalty2;
        a.wrap = SGMX_WRAP && a.dx != 0 && a.dy != 0; // diagonal sweeps: w wrapped diagonals (WrapWalk)
        a.npaths = a.dy == 0 ? c->h : (a.dx == 0 || a.wrap ? c->w : c->w + c->h - 1);
        a.first = k == 0;
        const dim3 grid((a.npaths + 3) / 4, n);
        ProfScope ps(c, SGMX_K_AGGREGATE, st, k);
        if (f32)
            launch_agg<true>(c->R, k == 0, c->prm.disp_range != c->Dp, a, grid, st);
        else
            launch_agg<false>(c->R, k == 0, c->prm.disp_range != c->Dp, a, grid, st);
        HIPCHK(c, hipGetLastError());
    }
    return SGMX_OK;
}

int run_wta_median(sgmx_ctx *c, int n, bool f32, uint16_t *d_disp, hipStream_t st)
{
    const int W = c->W, H = c->H;
    {
        ProfScope ps(c, SGMX_K_WTA, st);
        const dim3 grid(c->h, n);
        if (!f32) {
            const size_t lds = (size_t)c->w * 8;
            if (lds > 64 * 1024) {
                const void *fn = c->R == 1 ? (const void *)wta_u16_kernel<1> : (c->R == 2 ? (const void *)wta_u16_kernel<2> : (const void *)wta_u16_kernel<4>);
                HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (c->R == 1)
                hipLaunchKernelGGL(wta_u16_kernel<1>, grid, dim3(256), lds, st, (const uint16_t *)c->d_S, c->w, c->h, c->prm.disp_range, c->m, c->d_raw);
            else if (c->R == 2)
                hipLaunchKernelGGL(wta_u16_kernel<2>, grid, dim3(256), lds, st, (const uint16_t *)c->d_S, c->w, c->h, c->prm.disp_range, c->m, c->d_raw);
            else
                hipLaunchKernelGGL(wta_u16_kernel<4>, grid, dim3(256), lds, st, (const uint16_t *)c->d_S, c->w, c->h, c->prm.disp_range, c->m, c->d_raw);
        } else {
            const size_t lds = (size_t)c->w * 16;
            if (lds > 64 * 1024) {
                const void *fn = c->R == 1 ? (const void *)wta_f32_kernel<1> : (c->R == 2 ? (const void *)wta_f32_kernel<2> : (const void *)wta_f32_kernel<4>);
                HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (c->R == 1)
                hipLaunchKernelGGL(wta_f32_kernel<1>, grid, dim3(256), lds, st, (const float *)c->d_S, c->w, c->h, c->prm.disp_range, c->d_raw);
            else if (c->R == 2)
                hipLaunchKernelGGL(wta_f32_kernel<2>, grid, dim3(256), lds, st, (const float *)c->d_S, c->w, c->h, c->prm.disp_range, c->d_raw);
            else
                hipLaunchKernelGGL(wta_f32_kernel<4>, grid, dim3(256), lds, st, (const float *)c->d_S, c->w, c->h, c->prm.disp_range, c->d_raw);
        }
        HIPCHK(c, hipGetLastError());
    }
    {
        ProfScope ps(c, SGMX_K_MEDIAN, st);
        hipLaunchKernelGGL(median3_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, st, c->d_raw, W, H, d_disp);
        HIPCHK(c, hipGetLastError());
    }
    return SGMX_OK;
}

// lds: dynamic LDS of the launch (the staged variants); above 64 KiB the function attribute has to allow it.
template <int RP> hipError_t launch_agg3(bool first, bool pad, dim3 grid, hipStream_t st, const u32 *cl, const u32 *cr,
                                         uint8_t *E, uint16_t *BL, uint16_t *SB, const Agg2Args &a, size_t lds)
{
#ifndef SGMX_PF
#define SGMX_PF 4
#endif
#ifndef SGMX_PF8
#define SGMX_PF8 2
#endif
    constexpr int PF = RP >= 12 ? 2 : (RP == 8 ? SGMX_PF8 : SGMX_PF); // census prefetch depth in steps (2, 3, 6, 8 measured slower at RP = 4)
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, cl, cr, E, BL, SB, a);
        return hipSuccess;
    };
    // horizontal sweeps: census windows as a shift register across the lanes (aggregate3_kernel, HZ); the first
    // direction of either schedule is never horizontal
#ifndef SGMX_HZ
#define SGMX_HZ 1
#endif
    // (RP = 12, e.g. the 270 disparities of dense_stereo/main.cc:93: 0.58 -> 0.50 ms per launch of 8 KITTI-size pairs;
    //  RP = 16 gains nothing from it)
    if constexpr (SGMX_HZ && RP <= 12) {
        constexpr int PFH = RP <= 8 ? 8 : RP; // a multiple of RP
        if (a.dy == 0 && !first) {
            if (a.dx > 0)
                return pad ? go(aggregate3_kernel<RP, false, true, PFH, 1>) : go(aggregate3_kernel<RP, false, false, PFH, 1>);
            return pad ? go(aggregate3_kernel<RP, false, true, PFH, -1>) : go(aggregate3_kernel<RP, false, false, PFH, -1>);
        }
    }
    // diagonal sweeps: w wrapped diagonals of equal length h instead of w + h - 1 diagonals (aggregate3_kernel, WRAP)
    if (SGMX_WRAP && a.dx != 0 && a.dy != 0) {
        if (first)
            return pad ? go(aggregate3_kernel<RP, true, true, PF, 0, true>) : go(aggregate3_kernel<RP, true, false, PF, 0, true>);
        return pad ? go(aggregate3_kernel<RP, false, true, PF, 0, true>) : go(aggregate3_kernel<RP, false, false, PF, 0, true>);
    }
    if (first)
        return pad ? go(aggregate3_kernel<RP, true, true, PF>) : go(aggregate3_kernel<RP, true, false, PF>);
    return pad ? go(aggregate3_kernel<RP, false, true, PF>) : go(aggregate3_kernel<RP, false, false, PF>);
}

SumArgs sum_args(const sgmx_ctx *c, const FastView &v)
{
    SumArgs a;
    a.cl = v.cl;
    a.cr = v.cr;
    a.E = v.E;
    a.estride = v.bytes_E * v.cap;
    a.SB = v.SB;
    a.w = v.w;
    a.h = v.h;
    a.D = c->prm.disp_range;
    a.wp = v.wp;
    a.N = c->prm.num_paths;
    a.P2 = c->prm.penalty2;
    return a;
}

size_t wta3_lds_bytes(int DpF, int w) { return (size_t)16 * (DpF / 2) * 4 + (size_t)w * 16; }

// The context's own scratch as a view (valid after ensure_scratch(mode 1) on a fast context).
FastView ctx_view(const sgmx_ctx *c)
{
    FastView v;
    v.W = c->W; v.H = c->H; v.w = c->w; v.h = c->h; v.m = c->m; v.wp = c->wp;
    v.cap = c->prm.max_batch;
    v.bytes_E = c->bytes_E;
    v.bytes_BL = c->bytes_BL;
    v.cl = c->d_cl; v.cr = c->d_cr; v.E = c->d_E;
    v.BL = (uint16_t *)c->d_BL; v.SB = c->d_SB; v.raw = c->d_raw;
    v.BL2 = c->d_BL2;
    return v;
}

// Fast census path: the ordered direction sweeps writing byte volumes, then the fused
// sum + WTA / LR / subpixel kernel and the median.  prof == false: no event is created (the entry point
// that allocates nothing).
int run_fast(sgmx_ctx *c, const FastView &v, int n, uint16_t *d_disp, hipStream_t st, bool prof)
{
    const int np = c->prm.num_paths;
    const int was_profiling = c->profiling;
    if (!prof)
        c->profiling = 0;
    struct Restore {
        sgmx_ctx *c;
        int v;
        ~Restore() { c->profiling = v; }
    } restore{c, was_profiling};
    HIPCHK(c, hipMemsetAsync(v.BL, 0, v.bytes_BL * n, st)); // path buffer starts at zero (stereo_sgm.cc:46)
    HIPCHK(c, hipMemsetAsync(v.SB, 0, v.bytes_BL * n, st));
    // the launch's census images into the memory-side cache before the first sweep (cache_warm_kernel); a few
    // megabytes are still in the L2s the census kernel wrote them through
    if (((size_t)n * v.w * v.h + (size_t)n * v.wp * v.h) * 4 > ((size_t)16 << 20)) {
        const size_t ncl = (size_t)n * v.w * v.h * 4 / 16, ncr = (size_t)n * v.wp * v.h * 4 / 16;
        hipLaunchKernelGGL(cache_warm_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(v.cl), ncl, (u32 *)v.BL);
        hipLaunchKernelGGL(cache_warm_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(v.cr), ncr, (u32 *)v.BL);
        HIPCHK(c, hipGetLastError());
    }
    const bool pad = c->prm.disp_range != c->DpF;
    // SGMX_PROFILE_SWEEPS: the sweeps of the group share their boundary events (np + 1 markers on the stream
    // instead of two per launch of every kernel class)
    hipEvent_t chain = nullptr;
    if (c->profiling == SGMX_PROFILE_SWEEPS && hipEventCreate(&chain) == hipSuccess)
        (void)hipEventRecord(chain, st);
    // One direction sweep of the group on stream `on`: ring reads from `ring`, ring writes into `ring_out` when fuse == 2.
    auto sweep = [&](int k, hipStream_t on, int fuse, uint16_t *ring, uint16_t *ring_out) -> hipError_t {
        Agg2Args a;
        a.w = v.w;
        a.h = v.h;
        a.D = c->prm.disp_range;
        a.wp = v.wp;
        a.dx = np == 8 ? DIRS8[k][0] : DIRS4[k][0];
        a.dy = np == 8 ? DIRS8[k][1] : DIRS4[k][1];
        a.P1 = c->prm.penalty1;
        a.P2 = c->prm.penalty2;
        a.npaths = a.dy == 0 ? v.h : (a.dx == 0 || SGMX_WRAP ? v.w : v.w + v.h - 1); // (wrapped diagonals: w of them)
        a.npairs = n;
        a.fuse = fuse;
        a.BLout = ring_out;
        const int nwg = ((a.npaths + 15) / 16) * n; // 4 waves x 4 paths per workgroup
        const dim3 grid(8 * ((nwg + 7) / 8));       // 8 equal XCD chunks
        const size_t lds = 0;
        uint8_t *Ek = v.E + (size_t)k * v.bytes_E * v.cap;
        ProfScope ps(c, SGMX_K_AGGREGATE, on, k);
        switch (c->RP) {
        case 1: return launch_agg3<1>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        case 2: return launch_agg3<2>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        case 4: return launch_agg3<4>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        case 8: return launch_agg3<8>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        case 12: return launch_agg3<12>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        default: return launch_agg3<16>(k == 0, pad, grid, on, v.cl, v.cr, Ek, ring, v.SB, a, lds);
        }
    };
    // Small launch groups leave most of the chip idle during a sweep (one KITTI pair: 93 - 310 waves for 1 024 SIMDs,
    // and the two horizontal sweeps are 1 237 dependent steps each).  There the diagonal directions 0 and 5 run side
    // by side with the horizontal directions 1 and 6 that follow them (Agg2Args::fuse): the horizontal sweep on a
    // second stream, writing the second ring, which then becomes the ring.  Not under profiling (the events of a
    // class would overlap).  The fork and the join are events between the two streams, which a stream capture
    // records as branches of the graph (the entry point that allocates nothing runs this way too).
    // Measured (exp/small_groups.py, KITTI size): 1 / 2 / 4 / 6 pairs per call 1.06 / 1.11 / 1.34 / 1.55 ms
    // -> 0.96 / 1.01 / 1.23 / 1.45; 3840x2160 D=256 one pair 8.92 -> 7.97 ms.  Once the two launches together come
    // near the waves the chip holds it stops paying and then costs (KITTI, 32 pairs: -1.7 %; 3840x2160, 2 or 3 pairs:
    // -1 % / -4 %; 1241x376 D=270, 6 pairs: -2 %): bounded to 0.7 of the resident waves (1 024 SIMDs x the kernels'
    // waves per SIMD).
    const long two = (long)n * (((v.h + 3) / 4) + ((v.w + 3) / 4));
    const long resident = 1024L * (c->RP <= 4 ? 4 : (c->RP == 8 ? (pad ? 3 : 4) : 2));
    const bool side_by_side = SGMX_FUSE && SGMX_WRAP && np == 8 && !c->profiling && v.BL2 && c->aux_stream && 10 * two <= 7 * resident;
    uint16_t *ring = v.BL, *spare = v.BL2;
    for (int k = 0; k < np; k++) {
        hipError_t le;
        if (side_by_side && (k == 0 || k == 5)) {
            HIPCHK(c, hipEventRecord(c->ev_fork, st));
            HIPCHK(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            le = sweep(k + 1, c->aux_stream, 2, ring, spare);
            if (le == hipSuccess)
                le = sweep(k, st, 1, ring, nullptr);
            HIPCHK(c, le);
            HIPCHK(c, hipEventRecord(c->ev_join, c->aux_stream));
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
            std::swap(ring, spare);
            k++;
        } else {
            le = sweep(k, st, 0, ring, nullptr);
        }
        if (chain) {
            hipEvent_t next = nullptr;
            if (hipEventCreate(&next) == hipSuccess) {
                (void)hipEventRecord(next, st);
                c->prof.push_back(ProfRec{SGMX_K_AGGREGATE, k, chain, next, k == 0});
                chain = next;
            } else {
                if (k == 0)
                    (void)hipEventDestroy(chain);
                chain = nullptr;
            }
        }
        HIPCHK(c, le);
        HIPCHK(c, hipGetLastError());
    }
    {
        ProfScope ps(c, SGMX_K_WTA, st);
        const SumArgs sa = sum_args(c, v);
        const size_t lds = wta3_lds_bytes(c->DpF, v.w);
        const dim3 grid(v.h, n);
        const bool wpad = c->prm.disp_range != c->DpF;
        const int P2 = c->prm.penalty2;
#define SGMX_WTA(RPV)                                                                                                  \
    do {                                                                                                               \
        if (wpad) {                                                                                                    \
            if (lds > 64 * 1024)                                                                                       \
                HIPCHK(c, hipFuncSetAttribute((const void *)wta3_kernel<RPV, true>,                                    \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            hipLaunchKernelGGL((wta3_kernel<RPV, true>), grid, dim3(256), lds, st, sa, P2, v.m, v.raw);                 \
        } else {                                                                                                       \
            if (lds > 64 * 1024)                                                                                       \
                HIPCHK(c, hipFuncSetAttribute((const void *)wta3_kernel<RPV, false>,                                   \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
            hipLaunchKernelGGL((wta3_kernel<RPV, false>), grid, dim3(256), lds, st, sa, P2, v.m, v.raw);                \
        }                                                                                                              \
    } while (0)
        if (c->RP == 1)
            SGMX_WTA(1);
        else if (c->RP == 2)
            SGMX_WTA(2);
        else if (c->RP == 4)
            SGMX_WTA(4);
        else if (c->RP == 8)
            SGMX_WTA(8);
        else if (c->RP == 12)
            SGMX_WTA(12);
        else
            SGMX_WTA(16);
#undef SGMX_WTA
        HIPCHK(c, hipGetLastError());
    }
    {
        ProfScope ps(c, SGMX_K_MEDIAN, st);
        hipLaunchKernelGGL(median3_kernel, dim3((v.W + 255) / 256, v.H, n), dim3(256), 0, st, v.raw, v.W, v.H, d_disp);
        HIPCHK(c, hipGetLastError());
    }
    return SGMX_OK;
}

// The data-cost volume of pair `index` of the last launch group, pitch Dp, into d_dst
// (fast path: recomputed from the census images, which is all that path keeps).
int materialize_cost(sgmx_ctx *c, int index, uint8_t *d_dst, hipStream_t st)
{
    const size_t npix = (size_t)c->w * c->h;
    const size_t groups = npix * (c->DpF / 16);
    hipLaunchKernelGGL(hamming_volume_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                       c->d_cl + (size_t)index * npix, c->d_cr + (size_t)index * c->wp * c->h, c->wp, c->DpF, c->w, c->h,
                       c->prm.disp_range, c->DpF, d_dst, groups);
    HIPCHK(c, hipGetLastError());
    return SGMX_OK;
}

// Census transform of n pairs into the census images at cl / cr (right image: row pitch `pitch`, pixel x in
// column xoff + x).
int run_census(sgmx_ctx *c, int n, const uint8_t *dL, const uint8_t *dR, int W, int H, u32 *cl, u32 *cr, int pitch, int xoff,
               hipStream_t st, bool prof)
{
    const int m = (c->prm.window_sz - 1) / 2, w = W - 2 * m, h = H - 2 * m;
    const int was_profiling = c->profiling;
    if (!prof)
        c->profiling = 0;
    {
        ProfScope ps(c, SGMX_K_CENSUS, st);
        const dim3 g4((w + 1023) / 1024, h, 2 * n);
        if (c->prm.window_sz == 5 && W >= 8)
            hipLaunchKernelGGL(census4_kernel<5>, g4, dim3(256), 0, st, dL, dR, n, W, H, cl, cr, pitch, xoff);
        else if (c->prm.window_sz == 3 && W >= 8)
            hipLaunchKernelGGL(census4_kernel<3>, g4, dim3(256), 0, st, dL, dR, n, W, H, cl, cr, pitch, xoff);
        else
            hipLaunchKernelGGL(census_kernel, dim3((w + 255) / 256, h, 2 * n), dim3(256), 0, st, dL, dR, n, W, H,
                               c->prm.window_sz, cl, cr, pitch, xoff);
    }
    c->profiling = was_profiling;
    HIPCHK(c, hipGetLastError());
    return SGMX_OK;
}

// One launch group (n <= max_batch) of the census path on the context's scratch, everything device resident.
// Both formulations give the same bits; contexts with P2 <= 255 run the volume-free one (sgmx_kernels_v2.hpp),
// the faster one at every group size measured.
// Pairs per kernel launch on the volume-free path.  Every direction sweep reads the census images of all pairs of
// the launch again, in between its G stores.  Measured: while those images stay in the 256 MB
// Infinity Cache next to the write stream (~208 MB of them: 3 pairs at 3840 x 2160, D = 256) the reads cost
// nothing; once they come from HBM the sweeps take 20-40 % longer (8 such pairs per launch: 5.5 ms per pair and
// direction against 4.6 for 3).  And beyond 4 pairs per XCD (32) the images of the pairs an XCD works on
// overflow its L2 several times over (KITTI size: 32 pairs 5 140 frames/s, 48: 4 730, 64: 3 800).
int fast_launch_pairs(const sgmx_ctx *c, int w, int h, int wp)
{
    const size_t per = ((size_t)w * h + (size_t)wp * h) * 4;
    size_t g = ((size_t)208 << 20) / per;
    g = g < 1 ? 1 : (g > 32 ? 32 : g);
    return (int)g < c->prm.max_batch ? (int)g : c->prm.max_batch;
}

// base: index of the group's first pair within the caller's call (for sgmx_last_launch_group)
int census_group(sgmx_ctx *c, int n, const uint8_t *dL, const uint8_t *dR, uint16_t *d_disp, hipStream_t st, int base = 0)
{
    const int W = c->W, H = c->H, w = c->w, h = c->h;
    if (c->preblur) {
        // cv::GaussianBlur(img, img, Size(3,3), sigma) of the callers (dense_stereo/main.cc:83-85,
        // stereo_eval/main.cc:36-38) on both images of every pair, on the device, before anything reads them
        const size_t nb = (size_t)n * W * H;
        if (c->blur_cap < 2 * nb) {
            if (c->d_blur)
                HIPCHK(c, hipStreamSynchronize(st)); // (an earlier group of this call may still read it)
            (void)hipFree(c->d_blur);
            c->d_blur = nullptr;
            c->blur_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_blur, 2 * nb));
            c->blur_cap = 2 * nb;
        }
        const dim3 grid((W + 255) / 256, H, n);
        hipLaunchKernelGGL(gauss3_u8_kernel, grid, dim3(256), 0, st, dL, W, H, c->blur_k0, c->blur_k1, c->d_blur);
        hipLaunchKernelGGL(gauss3_u8_kernel, grid, dim3(256), 0, st, dR, W, H, c->blur_k0, c->blur_k1, c->d_blur + nb);
        HIPCHK(c, hipGetLastError());
        dL = c->d_blur;
        dR = c->d_blur + nb;
    }
    if (c->fast) {
        // launches of at most fast_launch_pairs pairs, one after the other on the same scratch
        const int cap = fast_launch_pairs(c, w, h, c->wp);
        const size_t nfull = (size_t)W * H;
        int rc = SGMX_OK;
        for (int i = 0; i < n && !rc; i += cap) {
            const int k = n - i < cap ? n - i : cap;
            rc = run_census(c, k, dL + i * nfull, dR + i * nfull, W, H, c->d_cl, c->d_cr, c->wp, c->DpF, st, true);
            if (!rc)
                rc = run_fast(c, ctx_view(c), k, d_disp + i * nfull, st, true);
            c->last_n = k;
            c->last_first = base + i;
        }
        if (!rc)
            rc = scratch_release(c, st);
        return rc;
    }
    // the materialised formulation: census, data-cost volume, sweeps over C and S, WTA
    int rc = run_census(c, n, dL, dR, W, H, c->d_cl, c->d_cr, w, 0, st, true);
    if (rc)
        return rc;
    {
        ProfScope ps(c, SGMX_K_HAMMING, st);
        const size_t groups = (size_t)n * w * h * (c->Dp / 16);
        hipLaunchKernelGGL(hamming_volume_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, c->d_cl,
                           c->d_cr, w, 0, w, h, c->prm.disp_range, c->Dp, (uint8_t *)c->d_C, groups);
        HIPCHK(c, hipGetLastError());
    }
    rc = run_aggregation(c, n, false, st, c->d_C);
    if (!rc)
        rc = run_wta_median(c, n, false, d_disp, st);
    c->last_n = n;
    c->last_first = base;
    if (!rc)
        rc = scratch_release(c, st);
    return rc;
}

// The descriptor data cost of one pair into d_C (pitch Dp): sgm.cc:76-84, or with replicate the variant of
// new_sgm/sgm_stereo.cc:158-181.
int run_descriptor_cost(sgmx_ctx *c, bool replicate, const float *d_fl, const float *d_fr, int W, int H, int K, hipStream_t st)
{
    const int D = c->prm.disp_range;
    const size_t lds = desc_lds_bytes(c->Dp);
    const int TX = desc_tile_pixels(c->Dp);
    const dim3 grid((W + TX - 1) / TX, H);
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(DESC_THREADS), lds, st, d_fl, d_fr, W, K, D, (float *)c->d_C);
        return hipSuccess;
    };
    ProfScope ps(c, SGMX_K_HAMMING, st); // data-cost class
    hipError_t le;
    if (c->Dp == 128)
        le = replicate ? go(descriptor_cost_kernel<true, 4, 4>) : go(descriptor_cost_kernel<false, 4, 4>);
    else if (c->Dp == 256)
        le = replicate ? go(descriptor_cost_kernel<true, 4, 8>) : go(descriptor_cost_kernel<false, 4, 8>);
    else if (c->Dp == 512)
        le = replicate ? go(descriptor_cost_kernel<true, 2, 16>) : go(descriptor_cost_kernel<false, 2, 16>);
    else
        return fail(c, SGMX_ERR_STATE, "descriptor cost: unexpected volume pitch");
    HIPCHK(c, le);
    HIPCHK(c, hipGetLastError());
    return SGMX_OK;
}

// SpeckleFilter(100, 2, image) of sgm_stereo.cc:413 on n integer disparity images.
int ns_speckle(sgmx_ctx *c, int n, int *d_img, hipStream_t st)
{
    const int w = c->w, h = c->h, npix = w * h;
    HIPCHK(c, hipMemsetAsync(c->d_size, 0, (size_t)npix * 4 * n, st));
    hipLaunchKernelGGL(ns_ccl_rows_kernel, dim3((h + 3) / 4, n), dim3(256), 0, st, d_img, w, h, 2, c->d_label);
    if (h > 1)
        hipLaunchKernelGGL(ns_ccl_cols_kernel, dim3((w + 255) / 256, h - 1, n), dim3(256), 0, st, d_img, w, h, 2, c->d_label);
    hipLaunchKernelGGL(ns_ccl_count_kernel, dim3((npix + 255) / 256, n), dim3(256), 0, st, c->d_label, npix, c->d_size);
    hipLaunchKernelGGL(ns_ccl_apply_kernel, dim3((npix + 255) / 256, n), dim3(256), 0, st, d_img, c->d_label, npix, c->d_size,
                       100);
    HIPCHK(c, hipGetLastError());
    return SGMX_OK;
}

template <int R> void launch_ns_wta(dim3 grid, hipStream_t st, const float *S, int w, int h, int D, int *disp, float *sub)
{
    hipLaunchKernelGGL(ns_wta_kernel<R>, grid, dim3(256), 0, st, S, w, h, D, disp, sub);
}

// recon::SGMStereo::ComputeSGM (new_sgm/sgm_stereo.cc:58-101) on n <= max_batch left data-cost volumes of
// pitch Dp, device resident.  The right-image pass runs first so that d_S holds the left sums afterwards
// (for the tap); the two passes are independent.
int ns_group(sgmx_ctx *c, int n, const float *left, int thr, uint16_t *d_disp, hipStream_t st, int base = 0)
{
    const int w = c->w, h = c->h, D = c->prm.disp_range;
    const size_t npix = (size_t)w * h;
    const dim3 rows(h, n);
    auto wta = [&](int *disp, float *sub) {
        ProfScope ps(c, SGMX_K_WTA, st);
        if (c->R == 1)
            launch_ns_wta<1>(rows, st, (const float *)c->d_S, w, h, D, disp, sub);
        else if (c->R == 2)
            launch_ns_wta<2>(rows, st, (const float *)c->d_S, w, h, D, disp, sub);
        else
            launch_ns_wta<4>(rows, st, (const float *)c->d_S, w, h, D, disp, sub);
    };
    {
        const size_t tot = (size_t)n * npix * c->Dp;
        ProfScope ps(c, SGMX_K_HAMMING, st); // data-cost class
        hipLaunchKernelGGL(ns_right_cost_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, left, w, D, c->Dp,
                           c->d_C2, tot);
        HIPCHK(c, hipGetLastError());
    }
    int rc = run_aggregation(c, n, true, st, c->d_C2, DIRS_NS);
    if (rc)
        return rc;
    wta(c->d_rd, nullptr);
    if ((rc = ns_speckle(c, n, c->d_rd, st)))
        return rc;
    if ((rc = run_aggregation(c, n, true, st, left, DIRS_NS)))
        return rc;
    wta(c->d_ld, c->d_sub);
    if ((rc = ns_speckle(c, n, c->d_ld, st)))
        return rc;
    hipLaunchKernelGGL(ns_lr_kernel, dim3((w + 255) / 256, h, n), dim3(256), 0, st, c->d_ld, c->d_rd, w, h, D, thr,
                       c->d_states);
    hipLaunchKernelGGL(ns_interpolate_kernel, dim3((h + 3) / 4, n), dim3(256), 0, st, c->d_states, c->d_sub, w, h);
    hipLaunchKernelGGL(ns_scale_kernel, dim3((unsigned)((npix * n + 255) / 256)), dim3(256), 0, st, c->d_sub, npix * n,
                       c->d_raw);
    {
        ProfScope ps(c, SGMX_K_MEDIAN, st);
        hipLaunchKernelGGL(median5_kernel, dim3((w + 255) / 256, h, n), dim3(256), 0, st, c->d_raw, w, h, d_disp);
    }
    HIPCHK(c, hipGetLastError());
    c->last_n = n;
    c->last_first = base;
    return scratch_release(c, st);
}

int ns_check(sgmx_ctx *c, int W, int H, int D, int thr)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (D != c->prm.disp_range)
        return fail(c, SGMX_ERR_INVALID_ARG, "D differs from params.disp_range");
    // SetSmoothnessCostParameters / SetConsistencyThreshold (sgm_stereo.cc:16-36) throw on these
    if (c->prm.penalty1 >= c->prm.penalty2)
        return fail(c, SGMX_ERR_INVALID_ARG, "SGMStereo: penalty1 must be smaller than penalty2");
    if (thr < 0)
        return fail(c, SGMX_ERR_INVALID_ARG, "SGMStereo: consistency threshold must not be negative");
    if (D > 256)
        return fail(c, SGMX_ERR_INVALID_ARG, "SGMStereo: disparity range above 256 overflows the 16-bit output (256 * d)");
    if (W < D)
        return fail(c, SGMX_ERR_INVALID_ARG, "SGMStereo: width smaller than the disparity range (the reference indexes out of bounds)");
    return SGMX_OK;
}

// uint16 guarantees of the census path: L_k <= 255*(k+1) + P2 on any pixel, so S <= 36*255 + 8*P2, and the
// "no disparity" sentinel 0x7FFF must stay above every L + P1 (sgmx_kernels.hpp INF16).  Checked on the
// census entry points only: the float pipelines hold no uint16 volume and take any penalty, as the reference does.
int check_census_penalties(sgmx_ctx *c)
{
    const sgmx_params &p = c->prm;
    if (36LL * 255 + 8LL * p.penalty2 > 65535 || 9LL * 255 + p.penalty2 + p.penalty1 >= 0x7FFF || 0x7FFFLL + p.penalty1 > 65535)
        return fail(c, SGMX_ERR_RANGE, "penalties too large for uint16 path costs (need 9180 + 8*P2 <= 65535)");
    return SGMX_OK;
}

int check_size(sgmx_ctx *c, int W, int H, bool census)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (census) {
        const int rc = check_census_penalties(c);
        if (rc)
            return rc;
    }
    const int m = census ? (c->prm.window_sz - 1) / 2 : 0;
    if (W - 2 * m < 1 || H - 2 * m < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "image smaller than the census window");
    const int Dp = census && c->fast ? c->DpF : c->Dp;
    // The sweeps address a pair's volume with 32-bit offsets: voffset + soffset reach the volume's size plus one
    // backward walk over it plus the wrapped diagonals' bias of w + h pixels (sgmx_kernels_v2.hpp), i.e. up to
    // (2 npix + 2 h) * Dp bytes, which has to stay below 2^32.
    if (((size_t)(W - 2 * m) * (H - 2 * m) + (size_t)(W - 2 * m) + (size_t)(H - 2 * m)) * Dp >= (size_t)1 << 31)
        return fail(c, SGMX_ERR_INVALID_ARG, "cost volume of one pair exceeds 2^31 voxels");
    if (W - 2 * m > 30000) // (the sweeps' alive masks are int16 arithmetic on x - d; the WTA's LDS tables end far below this)
        return fail(c, SGMX_ERR_INVALID_ARG, "image wider than 30000 pixels");
    // dynamic LDS of the WTA kernel this call will launch (a workgroup may use all 160 KiB of a CU)
    const int w = W - 2 * m;
    const size_t lds = census ? (c->fast ? wta3_lds_bytes(c->DpF, w) : (size_t)w * 8) : (size_t)w * 16;
    if (lds > 160 * 1024)
        return fail(c, SGMX_ERR_INVALID_ARG, "image too wide for the per-row LDS tables of the WTA kernel (" +
                                                 std::to_string(lds) + " bytes needed, 163840 available)");
    return SGMX_OK;
}

} // namespace

extern "C" {

const char *sgmx_version(void) { return "sgmx 0.1 (gfx950, HIP)"; }

const char *sgmx_last_error(const sgmx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sgmx_create(const sgmx_params *params, sgmx_ctx **out)
{
    if (!params || !out)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    sgmx_params p = *params;
    if (p.num_paths == 0)
        p.num_paths = 8;
    if (p.max_batch == 0)
        p.max_batch = 1;
    if (p.window_sz == 0)
        p.window_sz = 1;
    if (p.disp_range < 2 || p.disp_range > 512)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "disp_range must be in [2, 512]");
    if (p.window_sz < 1 || p.window_sz > 31 || (p.window_sz & 1) == 0)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "window_sz must be odd and in [1, 31]");
    if (p.num_paths != 4 && p.num_paths != 8)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "num_paths must be 4 or 8");
    if (p.penalty1 < 0 || p.penalty2 < 0 || p.max_batch < 1 || p.max_batch > 4096 || (p.flags & ~(SGMX_FLAG_MATERIALIZE | SGMX_FLAG_NO_SMALL_GROUPS)) != 0)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "negative penalty, bad max_batch or unknown flags");
    if (p.device < 0 || p.device >= MAX_DEVICES)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "device ordinal out of range");
    int ndev = 0;
    const hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev <= 0)
        return fail(nullptr, SGMX_ERR_NO_DEVICE,
                    std::string("no HIP device visible; sgmx has no CPU fallback") +
                        (de != hipSuccess ? std::string(" (hipGetDeviceCount: ") + hipGetErrorString(de) + ")" : std::string()));
    if (p.device < 0 || p.device >= ndev)
        return fail(nullptr, SGMX_ERR_INVALID_ARG, "device ordinal out of range");
    sgmx_ctx *c = new (std::nothrow) sgmx_ctx();
    if (!c)
        return fail(nullptr, SGMX_ERR_OOM, "host allocation failed");
    c->prm = p;
    c->R = p.disp_range <= 128 ? 1 : (p.disp_range <= 256 ? 2 : 4);
    c->Dp = 128 * c->R;
    c->fast = p.penalty2 <= 255 && !(p.flags & SGMX_FLAG_MATERIALIZE);
    c->RP = p.disp_range <= 32 ? 1 : (p.disp_range <= 64 ? 2 : (p.disp_range <= 128 ? 4 : (p.disp_range <= 256 ? 8 : (p.disp_range <= 384 ? 12 : 16))));
    c->DpF = 32 * c->RP;
    bool ok = hipSetDevice(p.device) == hipSuccess && hipStreamCreate(&c->own_stream) == hipSuccess &&
              hipStreamCreate(&c->copy_stream) == hipSuccess && hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) == hipSuccess && // (no implicit ordering with a caller's null stream)
              hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_scratch, hipEventDisableTiming) == hipSuccess &&
              // set-up of the entry point that allocates nothing: this device's address of the static scratch
              noalloc_scratch_for_device(p.device, &c->noalloc_scratch) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming) == hipSuccess;
    int cus = 0;
    if (ok && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p.device) == hipSuccess && cus > 0)
        c->num_cus = cus;
    if (!ok) {
        sgmx_destroy(c);
        return fail(nullptr, SGMX_ERR_HIP, "hipSetDevice / stream or event creation / hipGetSymbolAddress failed");
    }
    *out = c;
    return SGMX_OK;
}

void sgmx_destroy(sgmx_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->prm.device);
    (void)hipDeviceSynchronize();
    for (auto &r : ctx->prof) {
        if (r.owns_a)
            (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    free_scratch(ctx);
    for (void *p : ctx->ham_buf)
        if (p)
            (void)hipFree(p);
    for (int i = 0; i < 2; i++)
        for (hipEvent_t ev : {ctx->ev_in[i], ctx->ev_done[i], ctx->ev_out[i]})
            if (ev)
                (void)hipEventDestroy(ev);
    if (ctx->ev_scratch)
        (void)hipEventDestroy(ctx->ev_scratch);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->aux_stream)
        (void)hipStreamDestroy(ctx->aux_stream);
    for (hipEvent_t ev : {ctx->ev_fork, ctx->ev_join})
        if (ev)
            (void)hipEventDestroy(ev);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int sgmx_compute_census_u8_dev(sgmx_ctx *c, int n, const uint8_t *d_left, const uint8_t *d_right, int W, int H,
                               uint16_t *d_disp, void *stream)
{
    if (!c || !d_left || !d_right || !d_disp || n < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or n < 1");
    int rc = check_size(c, W, H, true);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = (hipStream_t)stream;
    rc = ensure_scratch(c, 1, W, H, st);
    if (rc)
        return rc;
    const size_t nfull = (size_t)W * H;
    for (int i = 0; i < n; i += c->prm.max_batch) {
        const int g = n - i < c->prm.max_batch ? n - i : c->prm.max_batch;
        rc = census_group(c, g, d_left + i * nfull, d_right + i * nfull, d_disp + i * nfull, st, i);
        if (rc)
            return rc;
    }
    return SGMX_OK;
}

size_t sgmx_noalloc_scratch_bytes(void) { return SGMX_NOALLOC_SCRATCH_BYTES; }

// sgmx_compute_census_u8_dev for callers that cannot allocate: same kernels, but every intermediate buffer is a
// slice of this device's static scratch array instead of the context's hipMalloc'd scratch, and no event is
// created.  No hipMalloc / hipFree / allocating routine is called on this path.
int sgmx_compute_census_u8_dev_noalloc(sgmx_ctx *c, int n, const uint8_t *d_left, const uint8_t *d_right, int W, int H,
                                       uint16_t *d_disp, void *stream)
{
    if (!c || !d_left || !d_right || !d_disp || n < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or n < 1");
    int rc = check_size(c, W, H, true);
    if (rc)
        return rc;
    if (!c->fast)
        return fail(c, SGMX_ERR_INVALID_ARG, "the no-allocation entry point serves the volume-free formulation only "
                                             "(penalty2 <= 255, no SGMX_FLAG_MATERIALIZE)");
    if (c->preblur)
        return fail(c, SGMX_ERR_INVALID_ARG, "the no-allocation entry point does not run the pre-blur of sgmx_set_preblur3 "
                                             "(its buffer is allocated on demand)");
    const int m = (c->prm.window_sz - 1) / 2, w = W - 2 * m, h = H - 2 * m, wp = w + c->DpF;
    const size_t lds = wta3_lds_bytes(c->DpF, w);
    if (lds > 64 * 1024)
        return fail(c, SGMX_ERR_INVALID_ARG, "image too wide for the no-allocation entry point (WTA tables above 64 KiB "
                                             "would need a function attribute change)");
    // Layout of one launch group of g pairs in the scratch: ONE function gives the offsets the view below uses and the
    // total the limit is checked against.  Pairs lie back to back inside a part at the sizes the kernels index with
    // (direction k's byte volumes at E + k * g * b_E, as sum_args / run_fast address them); parts start 256-byte aligned.
    const size_t npix = (size_t)w * h, nfull = (size_t)W * H;
    const size_t b_cl = npix * 4, b_cr = (size_t)wp * h * 4, b_E = npix * c->DpF, b_ring = (size_t)(2 * w + 2 * h) * c->DpF * 2,
                 b_raw = nfull * 2;
    const int np = c->prm.num_paths;
    struct Layout {
        size_t cl, cr, raw, E, BL, SB, BL2, total;
    };
    auto layout = [&](size_t g) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        Layout L;
        size_t off = 0;
        L.cl = off, off = up(off + g * b_cl);
        L.cr = off, off = up(off + g * b_cr);
        L.raw = off, off = up(off + g * b_raw);
        L.E = off, off = up(off + g * b_E * np);
        L.BL = off, off = up(off + g * b_ring);
        L.SB = off, off = up(off + g * b_ring);
        L.BL2 = off, off = up(off + g * b_ring); // second ring: two directions side by side in small launch groups
        L.total = off;
        return L;
    };
    if (layout(1).total > SGMX_NOALLOC_SCRATCH_BYTES)
        return fail(c, SGMX_ERR_INVALID_ARG,
                    "one pair needs " + std::to_string(layout(1).total) + " bytes of scratch; the no-allocation entry point has " +
                        std::to_string(SGMX_NOALLOC_SCRATCH_BYTES) + " (use sgmx_compute_census_u8_dev)");
    int g = fast_launch_pairs(c, w, h, wp);
    g = n < g ? n : g;
    while (layout((size_t)g).total > SGMX_NOALLOC_SCRATCH_BYTES)
        g--;
    const Layout lay = layout((size_t)g);
    HIPCHK(c, hipSetDevice(c->prm.device));
    unsigned char *base = c->noalloc_scratch; // hipGetSymbolAddress result of this device (sgmx_create)
    if (!base)
        return fail(c, SGMX_ERR_STATE, "static scratch address missing");
    FastView v;
    v.W = W; v.H = H; v.w = w; v.h = h; v.m = m; v.wp = wp;
    v.cap = g;
    v.bytes_E = b_E;
    v.bytes_BL = b_ring;
    v.cl = reinterpret_cast<u32 *>(base + lay.cl);
    v.cr = reinterpret_cast<u32 *>(base + lay.cr);
    v.raw = reinterpret_cast<uint16_t *>(base + lay.raw);
    v.E = base + lay.E;
    v.BL = reinterpret_cast<uint16_t *>(base + lay.BL);
    v.SB = reinterpret_cast<uint16_t *>(base + lay.SB);
    v.BL2 = reinterpret_cast<uint16_t *>(base + lay.BL2);
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; i += g) {
        const int k = n - i < g ? n - i : g;
        // The array keeps the previous call's bytes: clear what must start at zero -- the padding columns of the
        // right census image, the zero border of the disparity image (the rings are cleared in run_fast).
        HIPCHK(c, hipMemsetAsync(v.cr, 0, (size_t)k * wp * h * 4, st));
        HIPCHK(c, hipMemsetAsync(v.raw, 0, (size_t)k * nfull * 2, st));
        rc = run_census(c, k, d_left + (size_t)i * nfull, d_right + (size_t)i * nfull, W, H, v.cl, v.cr, wp, c->DpF, st, false);
        if (!rc)
            rc = run_fast(c, v, k, d_disp + (size_t)i * nfull, st, false);
        if (rc)
            return rc;
    }
    return SGMX_OK;
}

int sgmx_host_alloc(sgmx_ctx *c, size_t bytes, void **out)
{
    if (!c || !out || bytes == 0)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or zero size");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->prm.device));
    HIPCHK(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SGMX_OK;
}

int sgmx_host_free(sgmx_ctx *c, void *ptr)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (ptr)
        HIPCHK(c, hipHostFree(ptr));
    return SGMX_OK;
}

int sgmx_compute_census_u8_batch(sgmx_ctx *c, int n, const uint8_t *left, const uint8_t *right, int W, int H,
                                 uint16_t *disp)
{
    if (!c || !left || !right || !disp || n < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or n < 1");
    int rc = check_size(c, W, H, true);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    rc = ensure_scratch(c, 1, W, H, c->own_stream);
    if (rc)
        return rc;
    // Launch groups are pipelined over two staging sets: while group g computes on the compute stream,
    // the copy stream brings in group g+1 and takes out group g-1 (the scratch volumes are shared, so the
    // groups themselves run one after the other).
    hipStream_t st = c->own_stream, cs = c->copy_stream;
    const size_t nfull = (size_t)W * H;
    const int B = c->prm.max_batch, G = (n + B - 1) / B;
    uint8_t *dl[2] = {c->d_left, c->d_left2}, *dr[2] = {c->d_right, c->d_right2};
    uint16_t *dout[2] = {c->d_out, c->d_out2};
    auto count = [&](int g) { return n - g * B < B ? n - g * B : B; };
    auto copy_in = [&](int g) -> hipError_t {
        const int s = g & 1;
        hipError_t e = hipMemcpyAsync(dl[s], left + (size_t)g * B * nfull, nfull * count(g), hipMemcpyHostToDevice, cs);
        if (e == hipSuccess)
            e = hipMemcpyAsync(dr[s], right + (size_t)g * B * nfull, nfull * count(g), hipMemcpyHostToDevice, cs);
        if (e == hipSuccess)
            e = hipEventRecord(c->ev_in[s], cs);
        return e;
    };
    auto copy_out = [&](int g) -> hipError_t {
        const int s = g & 1;
        hipError_t e = hipStreamWaitEvent(cs, c->ev_done[s], 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(disp + (size_t)g * B * nfull, dout[s], nfull * 2 * count(g), hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess)
            e = hipEventRecord(c->ev_out[s], cs);
        return e;
    };
    hipError_t e = copy_in(0);
    for (int g = 0; g < G && e == hipSuccess && rc == SGMX_OK; g++) {
        const int s = g & 1;
        e = hipStreamWaitEvent(st, c->ev_in[s], 0);
        if (e == hipSuccess && g >= 2)
            e = hipStreamWaitEvent(st, c->ev_out[s], 0); // group g-2 has left this set's output buffer
        if (e != hipSuccess)
            break;
        rc = census_group(c, count(g), dl[s], dr[s], dout[s], st, g * B);
        if (rc)
            break;
        e = hipEventRecord(c->ev_done[s], st);
        if (e == hipSuccess && g + 1 < G) {
            if (g >= 1)
                e = hipStreamWaitEvent(cs, c->ev_done[s ^ 1], 0); // group g-1 has read the other set's inputs
            if (e == hipSuccess)
                e = copy_in(g + 1);
        }
        if (e == hipSuccess && g >= 1)
            e = copy_out(g - 1);
    }
    if (e == hipSuccess && rc == SGMX_OK)
        e = copy_out(G - 1);
    const hipError_t e1 = hipStreamSynchronize(st), e2 = hipStreamSynchronize(cs);
    if (rc)
        return rc;
    if (e != hipSuccess || e1 != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("batch copy / sync: ") +
                                         hipGetErrorString(e != hipSuccess ? e : (e1 != hipSuccess ? e1 : e2)));
    return SGMX_OK;
}

int sgmx_compute_census_u8(sgmx_ctx *c, const uint8_t *left, const uint8_t *right, int W, int H, int stride_left,
                           int stride_right, uint16_t *disp, int stride_disp)
{
    if (!c || !left || !right || !disp)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer");
    if (stride_left < W || stride_right < W || stride_disp < 2 * W)
        return fail(c, SGMX_ERR_INVALID_ARG, "row pitch smaller than the row");
    int rc = check_size(c, W, H, true);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = c->own_stream;
    rc = ensure_scratch(c, 1, W, H, st);
    if (rc)
        return rc;
    // The caller's images are pageable memory (cv::Mat rows): they go through a page-locked staging buffer of the
    // context with plain memcpy, so that the three transfers are real asynchronous DMAs (the runtime's own path
    // for pageable memory costs ~0.1 ms more per call; a 2-D copy from pageable memory several ms).  The left
    // image is on its way while the right one is still being staged.
    const size_t nfull = (size_t)W * H;
    if (c->h_stage_bytes < 4 * nfull) {
        if (c->h_stage)
            (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage, 4 * nfull, hipHostMallocDefault));
        c->h_stage_bytes = 4 * nfull;
    }
    unsigned char *hl = c->h_stage, *hr = hl + nfull, *ho = hr + nfull;
    auto rows_in = [&](unsigned char *dst, const uint8_t *src, int stride) {
        if (stride == W)
            std::memcpy(dst, src, nfull);
        else
            for (int y = 0; y < H; y++)
                std::memcpy(dst + (size_t)y * W, src + (size_t)y * stride, (size_t)W);
    };
    rows_in(hl, left, stride_left);
    HIPCHK(c, hipMemcpyAsync(c->d_left, hl, nfull, hipMemcpyHostToDevice, st));
    rows_in(hr, right, stride_right);
    HIPCHK(c, hipMemcpyAsync(c->d_right, hr, nfull, hipMemcpyHostToDevice, st));
    rc = census_group(c, 1, c->d_left, c->d_right, c->d_out, st);
    if (rc)
        return rc;
    HIPCHK(c, hipMemcpyAsync(ho, c->d_out, nfull * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (stride_disp == 2 * W)
        std::memcpy(disp, ho, nfull * 2);
    else
        for (int y = 0; y < H; y++)
            std::memcpy(reinterpret_cast<unsigned char *>(disp) + (size_t)y * stride_disp, ho + (size_t)y * W * 2, (size_t)W * 2);
    return SGMX_OK;
}

int sgmx_compute_cost_f32_dev(sgmx_ctx *c, int n, const float *d_cost, int W, int H, int D, uint16_t *d_disp,
                              void *stream)
{
    if (!c || !d_cost || !d_disp || n < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or n < 1");
    if (D != c->prm.disp_range)
        return fail(c, SGMX_ERR_INVALID_ARG, "D differs from params.disp_range");
    int rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = (hipStream_t)stream;
    rc = ensure_scratch(c, 2, W, H, st);
    if (rc)
        return rc;
    const size_t npix = (size_t)W * H;
    for (int i = 0; i < n; i += c->prm.max_batch) {
        const int g = n - i < c->prm.max_batch ? n - i : c->prm.max_batch;
        const float *src = d_cost + (size_t)i * npix * D;
        const void *vol = src; // D == Dp: stream the caller's volume in place
        if (D != c->Dp) {
            const size_t tot = (size_t)g * npix * c->Dp;
            hipLaunchKernelGGL(repitch_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, D,
                               (float *)c->d_C, c->Dp, (size_t)g * npix, 0.f);
            vol = c->d_C;
        }
        rc = run_aggregation(c, g, true, st, vol);
        if (rc)
            return rc;
        rc = run_wta_median(c, g, true, d_disp + i * npix, st);
        if (rc)
            return rc;
        c->last_n = g;
        c->last_first = i;
        if ((rc = scratch_release(c, st)))
            return rc;
    }
    return SGMX_OK;
}

int sgmx_compute_cost_f32(sgmx_ctx *c, const float *cost, int W, int H, int D, uint16_t *disp)
{
    if (!c || !cost || !disp)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer");
    if (D != c->prm.disp_range)
        return fail(c, SGMX_ERR_INVALID_ARG, "D differs from params.disp_range");
    int rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    const size_t npix = (size_t)W * H;
    float *d_in = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_in, npix * D * 4));
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemcpyAsync(d_in, cost, npix * D * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = ensure_scratch(c, 2, W, H, st);
        if (!rc)
            rc = sgmx_compute_cost_f32_dev(c, 1, d_in, W, H, D, c->d_out, st);
        if (!rc)
            e = hipMemcpyAsync(disp, c->d_out, npix * 2, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_in);
    if (rc)
        return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("float path copy/sync: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_compute_desc_f32(sgmx_ctx *c, const float *desc_left, const float *desc_right, int W, int H, int K,
                          uint16_t *disp)
{
    if (!c || !desc_left || !desc_right || !disp || K < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or K < 1");
    int rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    rc = ensure_scratch(c, 2, W, H, c->own_stream);
    if (rc)
        return rc;
    const size_t npix = (size_t)W * H, nb = npix * K * sizeof(float);
    float *d_f = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_f, 2 * nb));
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemcpyAsync(d_f, desc_left, nb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_f + npix * K, desc_right, nb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = run_descriptor_cost(c, false, d_f, d_f + npix * K, W, H, K, st);
        if (!rc)
            rc = run_aggregation(c, 1, true, st, c->d_C);
        if (!rc)
            rc = run_wta_median(c, 1, true, c->d_out, st);
        c->last_n = 1;
        c->last_first = 0;
        if (!rc)
            rc = scratch_release(c, st);
        if (!rc)
            e = hipMemcpyAsync(disp, c->d_out, npix * 2, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_f);
    if (rc)
        return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("descriptor path copy/sync: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_sgmstereo_cost_f32_dev(sgmx_ctx *c, int n, const float *d_cost, int W, int H, int D, int consistency_threshold,
                                uint16_t *d_disp, void *stream)
{
    if (!c || !d_cost || !d_disp || n < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or n < 1");
    int rc = ns_check(c, W, H, D, consistency_threshold);
    if (!rc)
        rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = (hipStream_t)stream;
    if ((rc = ensure_scratch(c, 3, W, H, st)))
        return rc;
    const size_t npix = (size_t)W * H;
    for (int i = 0; i < n; i += c->prm.max_batch) {
        const int g = n - i < c->prm.max_batch ? n - i : c->prm.max_batch;
        const float *src = d_cost + (size_t)i * npix * D;
        if (D != c->Dp) { // D == Dp: the caller's volume is streamed in place
            const size_t tot = (size_t)g * npix * c->Dp;
            hipLaunchKernelGGL(repitch_kernel<float>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, D,
                               (float *)c->d_C, c->Dp, (size_t)g * npix, 0.f);
            src = (const float *)c->d_C;
        }
        if ((rc = ns_group(c, g, src, consistency_threshold, d_disp + i * npix, st, i)))
            return rc;
    }
    return SGMX_OK;
}

int sgmx_sgmstereo_cost_f32(sgmx_ctx *c, const float *cost, int W, int H, int D, int consistency_threshold, uint16_t *disp)
{
    if (!c || !cost || !disp)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer");
    int rc = ns_check(c, W, H, D, consistency_threshold);
    if (!rc)
        rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    const size_t npix = (size_t)W * H;
    float *d_in = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_in, npix * D * 4));
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemcpyAsync(d_in, cost, npix * D * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = ensure_scratch(c, 3, W, H, st);
        if (!rc)
            rc = sgmx_sgmstereo_cost_f32_dev(c, 1, d_in, W, H, D, consistency_threshold, c->d_out, st);
        if (!rc)
            e = hipMemcpyAsync(disp, c->d_out, npix * 2, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_in);
    if (rc)
        return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("SGMStereo copy/sync: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_sgmstereo_desc_f32(sgmx_ctx *c, const float *desc_left, const float *desc_right, int W, int H, int K,
                            int consistency_threshold, uint16_t *disp)
{
    if (!c || !desc_left || !desc_right || !disp || K < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or K < 1");
    int rc = ns_check(c, W, H, c->prm.disp_range, consistency_threshold);
    if (!rc)
        rc = check_size(c, W, H, false);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    if ((rc = ensure_scratch(c, 3, W, H, c->own_stream)))
        return rc;
    const size_t npix = (size_t)W * H, nb = npix * K * sizeof(float);
    float *d_f = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_f, 2 * nb));
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemcpyAsync(d_f, desc_left, nb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_f + npix * K, desc_right, nb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = run_descriptor_cost(c, true, d_f, d_f + npix * K, W, H, K, st);
        if (!rc)
            rc = ns_group(c, 1, (const float *)c->d_C, consistency_threshold, c->d_out, st);
        if (!rc)
            e = hipMemcpyAsync(disp, c->d_out, npix * 2, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_f);
    if (rc)
        return rc;
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("SGMStereo descriptor path copy/sync: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_kitti_eval_u16(sgmx_ctx *c, int n, const uint16_t *disp, const uint16_t *gt_noc, const uint16_t *gt_occ, int W,
                        int H, float *errors, uint16_t *ipol)
{
    if (!c || !disp || !gt_noc || !gt_occ || !errors || n < 1 || W < 1 || H < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    if ((size_t)W * H >= (size_t)1 << 30)
        return fail(c, SGMX_ERR_INVALID_ARG, "image too large");
    HIPCHK(c, hipSetDevice(c->prm.device));
    const size_t npix = (size_t)W * H;
    const int chunk = n < 64 ? n : 64; // images per launch group
    uint16_t *d_u16 = nullptr; // disp | gt_noc | gt_occ | ipol as 16-bit
    float *d_ip = nullptr, *d_err = nullptr;
    hipError_t e = hipMalloc((void **)&d_u16, npix * 2 * 4 * chunk);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_ip, npix * 4 * chunk);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_err, sizeof(float) * 26 * chunk);
    hipStream_t st = c->own_stream;
    for (int i = 0; i < n && e == hipSuccess; i += chunk) {
        const int g = n - i < chunk ? n - i : chunk;
        const size_t gb = npix * 2 * g;
        uint16_t *d_d = d_u16, *d_n = d_u16 + npix * chunk, *d_o = d_u16 + 2 * npix * chunk, *d_w = d_u16 + 3 * npix * chunk;
        e = hipMemcpyAsync(d_d, disp + i * npix, gb, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_n, gt_noc + i * npix, gb, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_o, gt_occ + i * npix, gb, hipMemcpyHostToDevice, st);
        if (e != hipSuccess)
            break;
        hipLaunchKernelGGL(kitti_rows_kernel, dim3((H + 3) / 4, g), dim3(256), 0, st, d_d, W, H, d_ip);
        hipLaunchKernelGGL(kitti_cols_kernel, dim3((W + 255) / 256, g), dim3(256), 0, st, d_ip, W, H);
        hipLaunchKernelGGL(kitti_errors_kernel, dim3(g), dim3(256), 0, st, d_d, d_ip, d_n, d_o, W, H, d_err);
        if (ipol)
            hipLaunchKernelGGL(kitti_write_kernel, dim3((unsigned)((npix * g + 255) / 256)), dim3(256), 0, st, d_ip, npix * g, d_w);
        e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(errors + (size_t)i * 26, d_err, sizeof(float) * 26 * g, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && ipol)
            e = hipMemcpyAsync(ipol + i * npix, d_w, gb, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(d_u16);
    (void)hipFree(d_ip);
    (void)hipFree(d_err);
    if (e != hipSuccess)
        return fail(c, e == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP, std::string("KITTI evaluation: ") + hipGetErrorString(e));
    for (int i = 0; i < n; i++)
        if (errors[(size_t)i * 26] < 0.f || errors[(size_t)i * 26 + 11] < 0.f)
            return fail(c, SGMX_ERR_INVALID_ARG, "KITTI evaluation: ground truth " + std::to_string(i) + " holds no valid pixel");
    return SGMX_OK;
}

int sgmx_depth_from_disparity_u16(sgmx_ctx *c, const uint16_t *disp, int W, int H, double focal, double baseline,
                                  uint16_t *depth)
{
    if (!c || !disp || !depth || W < 1 || H < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    HIPCHK(c, hipSetDevice(c->prm.device));
    const size_t n = (size_t)W * H;
    uint16_t *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, n * 4 + 16));
    int *flag = reinterpret_cast<int *>(d + 2 * n); // behind the two images: byte offset 4 n
    hipStream_t st = c->own_stream;
    int over = 0;
    hipError_t e = hipMemcpyAsync(d, disp, n * 2, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemsetAsync(flag, 0, 4, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(depth_from_disparity_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, n, focal,
                           baseline, d + n, flag);
        e = hipMemcpyAsync(depth, d + n, n * 2, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(&over, flag, 4, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("depth image: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    if (over)
        return fail(c, SGMX_ERR_RANGE, "depth image: a depth of 65536 cm or more does not fit 16 bits (the reference throws)");
    return SGMX_OK;
}

// getGaussianKernel(3, sigma, CV_32F), then 8 fractional bits (see include/sgmx.h): centre tap k0, side taps k1
static void gauss3_taps_q8(double sigma, int *k0, int *k1)
{
    const double sx = sigma > 0 ? sigma : ((3 - 1) * 0.5 - 1) * 0.3 + 0.8;
    float kf[3];
    double sum = 0;
    for (int i = 0; i < 3; i++) {
        const double x = i - 1.0;
        kf[i] = (float)std::exp(-0.5 / (sx * sx) * x * x);
        sum += kf[i];
    }
    sum = 1. / sum;
    *k0 = (int)std::lrint((double)(float)(kf[1] * sum) * 256.0);
    *k1 = (int)std::lrint((double)(float)(kf[0] * sum) * 256.0);
}

int sgmx_set_preblur3(sgmx_ctx *c, int enabled, double sigma)
{
    if (!c || !(sigma >= 0.0) || !std::isfinite(sigma))
        return fail(c, SGMX_ERR_INVALID_ARG, "null context or bad sigma");
    c->preblur = enabled != 0;
    gauss3_taps_q8(sigma, &c->blur_k0, &c->blur_k1);
    return SGMX_OK;
}

int sgmx_gaussian_blur3_u8(sgmx_ctx *c, const uint8_t *src, int W, int H, int stride_src, double sigma, uint8_t *dst,
                           int stride_dst)
{
    if (!c || !src || !dst || W < 1 || H < 1 || stride_src < W || stride_dst < W)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size / pitch");
    HIPCHK(c, hipSetDevice(c->prm.device));
    int k0, k1;
    gauss3_taps_q8(sigma, &k0, &k1);
    uint8_t *d = nullptr;
    const size_t nb = (size_t)W * H;
    HIPCHK(c, hipMalloc((void **)&d, 2 * nb));
    hipStream_t st = c->own_stream;
    hipError_t e = stride_src == W ? hipMemcpyAsync(d, src, nb, hipMemcpyHostToDevice, st)
                                   : hipMemcpy2DAsync(d, W, src, stride_src, W, H, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gauss3_u8_kernel, dim3((W + 255) / 256, H, 1), dim3(256), 0, st, d, W, H, k0, k1, d + nb);
        e = stride_dst == W ? hipMemcpyAsync(dst, d + nb, nb, hipMemcpyDeviceToHost, st)
                            : hipMemcpy2DAsync(dst, stride_dst, d + nb, W, W, H, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("gaussian blur: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_census_u32(sgmx_ctx *c, const uint8_t *img, int W, int H, int stride, uint32_t *out)
{
    if (!c || !img || !out || stride < W)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad pitch");
    int rc = check_size(c, W, H, true);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->prm.device));
    const int m = (c->prm.window_sz - 1) / 2, w = W - 2 * m, h = H - 2 * m;
    uint8_t *d_img = nullptr;
    u32 *d_o = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_img, (size_t)W * H));
    if (hipMalloc((void **)&d_o, (size_t)w * h * 4 * 2) != hipSuccess) {
        (void)hipFree(d_img);
        return fail(c, SGMX_ERR_OOM, "hipMalloc failed");
    }
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemcpy2DAsync(d_img, W, img, stride, W, H, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(census_kernel, dim3((w + 255) / 256, h, 1), dim3(256), 0, st, d_img, d_img, 1, W, H,
                       c->prm.window_sz, d_o, d_o + (size_t)w * h, w, 0);
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d_o, (size_t)w * h * 4, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_img);
    (void)hipFree(d_o);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("census tap: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_hamming_volume_u8(sgmx_ctx *c, const uint32_t *cl, const uint32_t *cr, int w, int h, uint8_t *out)
{
    if (!c || !cl || !cr || !out || w < 1 || h < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    HIPCHK(c, hipSetDevice(c->prm.device));
    const int D = c->prm.disp_range, Dp = c->Dp;
    const size_t npix = (size_t)w * h;
    u32 *d_c = nullptr;
    uint8_t *d_v = nullptr, *d_o = nullptr;
    hipError_t e = hipMalloc((void **)&d_c, npix * 8);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_v, npix * Dp);
    if (e == hipSuccess)
        e = hipMalloc((void **)&d_o, npix * D);
    hipStream_t st = c->own_stream;
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_c, cl, npix * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_c + npix, cr, npix * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const size_t groups = npix * (Dp / 16);
        hipLaunchKernelGGL(hamming_volume_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, d_c,
                           d_c + npix, w, 0, w, h, D, Dp, d_v, groups);
        const size_t tot = npix * D;
        hipLaunchKernelGGL(repitch_kernel<uint8_t>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_v, Dp, d_o,
                           D, npix, (uint8_t)0);
        e = hipMemcpyAsync(out, d_o, tot, hipMemcpyDeviceToHost, st);
    }
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_c);
    (void)hipFree(d_v);
    (void)hipFree(d_o);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("hamming tap: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

// One of the context's grow-only Hamming buffers, at least `bytes` large (the previous call's stream work has been
// waited for: every Hamming entry point synchronises before it returns).
static hipError_t ham_buffer(sgmx_ctx *c, int slot, size_t bytes, void **out)
{
    if (c->ham_cap[slot] < bytes) {
        if (c->ham_buf[slot])
            (void)hipFree(c->ham_buf[slot]);
        c->ham_buf[slot] = nullptr;
        c->ham_cap[slot] = 0;
        const size_t want = bytes + bytes / 4 + 256; // some room, so that slowly growing inputs do not reallocate every call
        hipError_t e = hipMalloc(&c->ham_buf[slot], want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&c->ham_buf[slot], bytes);
            if (e != hipSuccess)
                return e;
            c->ham_cap[slot] = bytes;
        } else {
            c->ham_cap[slot] = want;
        }
    }
    *out = c->ham_buf[slot];
    return hipSuccess;
}

// signatures of 4, 8, 16, 32 bytes (the reference's fast paths are 4 / 8 / 16, core/hamming.c:183-213) run the
// register-tiled kernels
#ifndef SGMX_HAM_NAIVE // (1: A/B builds only -- every signature length through the generic one-thread-per-distance kernels)
#define SGMX_HAM_NAIVE 0
#endif
static int ham_words(int ncodes)
{
    if (SGMX_HAM_NAIVE)
        return 0;
    return ncodes == 4 ? 1 : ncodes == 8 ? 2 : ncodes == 16 ? 4 : ncodes == 32 ? 8 : 0;
}

int sgmx_hamming_matrix_u16(sgmx_ctx *c, uint16_t *dis, const uint8_t *a, const uint8_t *b, int na, int nb, int ncodes)
{
    if (!c || !dis || !a || !b || na < 1 || nb < 1 || ncodes < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    HIPCHK(c, hipSetDevice(c->prm.device));
    void *d_a = nullptr, *d_b = nullptr, *d_d = nullptr;
    hipError_t e = ham_buffer(c, HAM_A, (size_t)na * ncodes + 4, &d_a);
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_B, (size_t)nb * ncodes + 4, &d_b);
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_DIS, (size_t)na * nb * 2, &d_d);
    hipStream_t st = c->own_stream;
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_a, a, (size_t)na * ncodes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_b, b, (size_t)nb * ncodes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        {
            ProfScope ps(c, SGMX_K_HAMLIB, st);
            const dim3 tg((na + 256 * HAM_II - 1) / (256 * HAM_II), (nb + HAM_JT - 1) / HAM_JT);
            const u32 *wa = static_cast<const u32 *>(d_a), *wb = static_cast<const u32 *>(d_b);
            uint16_t *dd = static_cast<uint16_t *>(d_d);
            switch (ham_words(ncodes)) {
            case 1: hipLaunchKernelGGL(hamming_matrix_tiled_kernel<1>, tg, dim3(256), 0, st, dd, wa, wb, na, nb); break;
            case 2: hipLaunchKernelGGL(hamming_matrix_tiled_kernel<2>, tg, dim3(256), 0, st, dd, wa, wb, na, nb); break;
            case 4: hipLaunchKernelGGL(hamming_matrix_tiled_kernel<4>, tg, dim3(256), 0, st, dd, wa, wb, na, nb); break;
            case 8: hipLaunchKernelGGL(hamming_matrix_tiled_kernel<8>, tg, dim3(256), 0, st, dd, wa, wb, na, nb); break;
            default:
                hipLaunchKernelGGL(hamming_matrix_kernel, dim3((na + 255) / 256, nb), dim3(256), 0, st, dd,
                                   static_cast<const uint8_t *>(d_a), static_cast<const uint8_t *>(d_b), na, nb, ncodes);
            }
        }
        e = hipMemcpyAsync(dis, d_d, (size_t)na * nb * 2, hipMemcpyDeviceToHost, st);
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, e == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP,
                    std::string("hamming matrix: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

// Thresholded matching of binary signatures (core/hamming.c:300-322, 385-407, 725-765, 810-839).  cross: one
// set against itself, pairs j > i only.  idx == nullptr: count only.
// records != nullptr: the allocating forms (match_hamming_thres, crossmatch_hamming; core/hamming.c:533-577, 768-807):
// once the count is known a buffer of max(bufsize, count, 1) hammatch_t records is malloc'ed into *records (the
// reference mallocs bufsize and grows by realloc; the caller frees it either way) and filled in loop order.
// Exclusive prefix sum of the n counts at d_cnt in place, their total to d_cnt[n]; the block sums of the three-launch
// form live behind it (the HAM_CNT buffer holds ham_scan_room(n) counts).
static size_t ham_scan_room(size_t n) { return n + 2 + (n + 1023) / 1024 + 1; }
static void ham_scan(unsigned long long *d_cnt, long n, hipStream_t st)
{
    if (n <= 4096) {
        hipLaunchKernelGGL(ham_scan_kernel, dim3(1), dim3(1024), 0, st, d_cnt, (int)n, d_cnt + n);
        return;
    }
    const int nblk = (int)((n + 1023) / 1024);
    unsigned long long *bsum = d_cnt + n + 1;
    hipLaunchKernelGGL(ham_scan_blocks_kernel, dim3(nblk), dim3(1024), 0, st, d_cnt, (int)n, bsum);
    hipLaunchKernelGGL(ham_scan_kernel, dim3(1), dim3(1024), 0, st, bsum, nblk, d_cnt + n);
    hipLaunchKernelGGL(ham_scan_add_kernel, dim3(nblk), dim3(1024), 0, st, d_cnt, (int)n, bsum);
}

static int ham_match(sgmx_ctx *c, const uint8_t *bs1, const uint8_t *bs2, long n1, long n2, int ht, int ncodes, bool cross,
                     int *idx, uint16_t *hams, size_t capacity, size_t *nptr, sgmx_hammatch **records = nullptr,
                     size_t bufsize = 0)
{
    if (records)
        *records = nullptr; // argument errors leave nothing to release
    if (!c || !nptr || n1 < 0 || n2 < 0 || ncodes < 1 || n1 > 0x7fffffffL || n2 > 0x7fffffffL || (n1 > 0 && !bs1) ||
        (!cross && n2 > 0 && !bs2) || (idx && !hams))
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    *nptr = 0;
    if (records) {
        *records = static_cast<sgmx_hammatch *>(std::malloc((bufsize ? bufsize : 1) * sizeof(sgmx_hammatch)));
        if (!*records)
            return fail(c, SGMX_ERR_OOM, "host allocation failed");
    }
    if (n1 == 0 || n2 == 0)
        return SGMX_OK;
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = c->own_stream;
    void *d_a = nullptr, *d_b = nullptr, *d_cntv = nullptr, *d_idx = nullptr, *d_h = nullptr, *d_rec = nullptr;
    unsigned long long total = 0;
    hipError_t e = ham_buffer(c, HAM_A, (size_t)n1 * ncodes + 4, &d_a);
    if (e == hipSuccess) {
        if (cross)
            d_b = d_a;
        else
            e = ham_buffer(c, HAM_B, (size_t)n2 * ncodes + 4, &d_b);
    }
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_CNT, ham_scan_room((size_t)n1) * sizeof(unsigned long long), &d_cntv);
    unsigned long long *d_cnt = static_cast<unsigned long long *>(d_cntv);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_a, bs1, (size_t)n1 * ncodes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && !cross)
        e = hipMemcpyAsync(d_b, bs2, (size_t)n2 * ncodes, hipMemcpyHostToDevice, st);
    const int nw = ham_words(ncodes);
    const dim3 grid((unsigned)((n1 + 3) / 4)), tgrid((unsigned)((n1 + 4 * HAM_QW - 1) / (4 * HAM_QW)));
    const uint8_t *ba = static_cast<const uint8_t *>(d_a), *bb = static_cast<const uint8_t *>(d_b);
    const u32 *wa = static_cast<const u32 *>(d_a), *wb = static_cast<const u32 *>(d_b);
    // count (emit == false) or emit pass
    auto pass = [&](bool emit) {
        ProfScope ps(c, SGMX_K_HAMLIB, st);
        int *di = static_cast<int *>(d_idx);
        uint16_t *dh = static_cast<uint16_t *>(d_h);
#define SGMX_HAM_PASS(NW)                                                                                              \
    do {                                                                                                               \
        if (emit)                                                                                                      \
            hipLaunchKernelGGL((ham_match_tiled_kernel<NW, true>), tgrid, dim3(256), 0, st, wa, wb, (int)n1, (int)n2, ht, cross ? 1 : 0, \
                               d_cnt, total, di, dh);                                                                  \
        else                                                                                                           \
            hipLaunchKernelGGL((ham_match_tiled_kernel<NW, false>), tgrid, dim3(256), 0, st, wa, wb, (int)n1, (int)n2, ht, cross ? 1 : 0, \
                               d_cnt, 0ull, (int *)nullptr, (uint16_t *)nullptr);                                      \
    } while (0)
        switch (nw) {
        case 1: SGMX_HAM_PASS(1); break;
        case 2: SGMX_HAM_PASS(2); break;
        case 4: SGMX_HAM_PASS(4); break;
        case 8: SGMX_HAM_PASS(8); break;
        default:
            if (emit)
                hipLaunchKernelGGL(ham_emit_kernel, grid, dim3(256), 0, st, ba, bb, (int)n1, (int)n2, ncodes, ht, cross ? 1 : 0, d_cnt,
                                   total, di, dh);
            else
                hipLaunchKernelGGL(ham_count_kernel, grid, dim3(256), 0, st, ba, bb, (int)n1, (int)n2, ncodes, ht, cross ? 1 : 0, d_cnt);
        }
#undef SGMX_HAM_PASS
    };
    if (e == hipSuccess) {
        pass(false);
        ham_scan(d_cnt, n1, st);
        e = hipMemcpyAsync(&total, d_cnt + n1, sizeof(total), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st); // nothing of this call may still run when the next one regrows the buffers
        return fail(c, e == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP, std::string("hamming match: ") + hipGetErrorString(e));
    }
    *nptr = (size_t)total;
    if ((!idx && !records) || total == 0)
        return SGMX_OK;
    if (records && (size_t)total > (bufsize ? bufsize : 1)) {
        void *grown = std::realloc(*records, (size_t)total * sizeof(sgmx_hammatch));
        if (!grown)
            return fail(c, SGMX_ERR_OOM, "host allocation failed");
        *records = static_cast<sgmx_hammatch *>(grown);
    }
    if (!records && (size_t)total > capacity)
        return fail(c, SGMX_ERR_INVALID_ARG, "hamming match: more matches than the caller's capacity (count first)");
    e = ham_buffer(c, HAM_IDX, (size_t)total * 2 * sizeof(int), &d_idx);
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_H, (size_t)total * sizeof(uint16_t), &d_h);
    if (e == hipSuccess) {
        pass(true);
        if (!records)
            e = hipMemcpyAsync(idx, d_idx, (size_t)total * 2 * sizeof(int), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess && !records)
        e = hipMemcpyAsync(hams, d_h, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && records) {
        static_assert(sizeof(sgmx_hammatch) == 12, "hammatch_t is two ints and a uint16 (core/hamming.h:52-56)");
        e = ham_buffer(c, HAM_REC, (size_t)total * sizeof(sgmx_hammatch), &d_rec);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(ham_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<const int *>(d_idx),
                               static_cast<const uint16_t *>(d_h), total, static_cast<int *>(d_rec));
            e = hipMemcpyAsync(*records, d_rec, (size_t)total * sizeof(sgmx_hammatch), hipMemcpyDeviceToHost, st);
        }
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, e == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP,
                    std::string("hamming match: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_match_hamming_count(sgmx_ctx *c, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes, size_t *nptr)
{
    return ham_match(c, bs1, bs2, n1, n2, ht, ncodes, false, nullptr, nullptr, 0, nptr);
}

int sgmx_crossmatch_hamming_count(sgmx_ctx *c, const uint8_t *dbs, long n, int ht, int ncodes, size_t *nptr)
{
    return ham_match(c, dbs, dbs, n, n, ht, ncodes, true, nullptr, nullptr, 0, nptr);
}

int sgmx_match_hamming_thres_prealloc(sgmx_ctx *c, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes,
                                      int *idx, uint16_t *hams, size_t capacity, size_t *nptr)
{
    if (!idx || !hams)
        return fail(c, SGMX_ERR_INVALID_ARG, "null output pointer");
    return ham_match(c, bs1, bs2, n1, n2, ht, ncodes, false, idx, hams, capacity, nptr);
}

int sgmx_crossmatch_hamming_prealloc(sgmx_ctx *c, const uint8_t *dbs, long n, int ht, int ncodes, int *idx, uint16_t *hams,
                                     size_t capacity, size_t *nptr)
{
    if (!idx || !hams)
        return fail(c, SGMX_ERR_INVALID_ARG, "null output pointer");
    return ham_match(c, dbs, dbs, n, n, ht, ncodes, true, idx, hams, capacity, nptr);
}

int sgmx_match_hamming_thres(sgmx_ctx *c, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes,
                             size_t bufsize, sgmx_hammatch **hmptr, size_t *nptr)
{
    if (!hmptr)
        return fail(c, SGMX_ERR_INVALID_ARG, "null output pointer");
    return ham_match(c, bs1, bs2, n1, n2, ht, ncodes, false, nullptr, nullptr, 0, nptr, hmptr, bufsize);
}

int sgmx_crossmatch_hamming(sgmx_ctx *c, const uint8_t *dbs, long n, int ht, int ncodes, long bufsize, sgmx_hammatch **hmptr,
                            size_t *nptr)
{
    if (!hmptr || bufsize < 0)
        return fail(c, SGMX_ERR_INVALID_ARG, "null output pointer or negative bufsize");
    return ham_match(c, dbs, dbs, n, n, ht, ncodes, true, nullptr, nullptr, 0, nptr, hmptr, (size_t)bufsize);
}

// compute_hamming_thread (core/hamming.c:849-860) is compute_hamming with the rows of the matrix dealt to OpenMP
// threads: the same matrix.  The GPU form is parallel either way.
int sgmx_compute_hamming_thread(sgmx_ctx *c, uint16_t *dis, const uint8_t *a, const uint8_t *b, int na, int nb, int ncodes)
{
    return sgmx_hamming_matrix_u16(c, dis, a, b, na, nb, ncodes);
}

int sgmx_match_hamming_thres_nt(sgmx_ctx *c, const uint8_t *bs1, const uint8_t *bs2, int n1, int n2, int ht, int ncodes, int nt,
                                int **keys, uint16_t **ham, size_t *nmatches)
{
    (void)nt; // the reference's thread count: does not change the result there either
    if (!c || !bs1 || !bs2 || !keys || !ham || !nmatches || n1 < 1 || n2 < 1 || ncodes < 1)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer or bad size");
    *keys = nullptr;
    *ham = nullptr;
    *nmatches = 0;
    const int nblock1 = 1 + (n1 - 1) / HAMBLOCK, nblock2 = 1 + (n2 - 1) / HAMBLOCK;
    if (nblock1 != nblock2)
        return fail(c, SGMX_ERR_INVALID_ARG,
                    "match_hamming_thres_nt: the reference indexes both sets by bl / nblock1 and bl % nblock1 "
                    "(core/hamming.c:876-877), which leaves the sets unless both have the same number of 128-blocks");
    HIPCHK(c, hipSetDevice(c->prm.device));
    hipStream_t st = c->own_stream;
    const long nitems = (long)nblock1 * nblock2 * HAMBLOCK;
    const HamBlocked g{nblock1, n1, n2};
    uint8_t *d_a = nullptr, *d_b = nullptr;
    unsigned long long *d_cnt = nullptr, total = 0;
    int *d_idx = nullptr;
    uint16_t *d_h = nullptr;
    // (device buffers: the context's grow-only ones, nothing to release here)
    hipError_t e = ham_buffer(c, HAM_A, (size_t)n1 * ncodes + 4, (void **)&d_a);
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_B, (size_t)n2 * ncodes + 4, (void **)&d_b);
    if (e == hipSuccess)
        e = ham_buffer(c, HAM_CNT, ham_scan_room((size_t)nitems) * sizeof(unsigned long long), (void **)&d_cnt);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_a, bs1, (size_t)n1 * ncodes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_b, bs2, (size_t)n2 * ncodes, hipMemcpyHostToDevice, st);
    const dim3 grid((unsigned)((nitems + 3) / 4));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ham_count_blocked_kernel, grid, dim3(256), 0, st, d_a, d_b, g, nitems, ncodes, ht, d_cnt);
        ham_scan(d_cnt, nitems, st);
        e = hipMemcpyAsync(&total, d_cnt + nitems, sizeof(total), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
        e = hipStreamSynchronize(st);
    if (e == hipSuccess) {
        // malloc(0) may return NULL: keep one element so that a caller's free() and pointer checks behave
        *keys = static_cast<int *>(std::malloc(((size_t)total ? (size_t)total : 1) * 2 * sizeof(int)));
        *ham = static_cast<uint16_t *>(std::malloc(((size_t)total ? (size_t)total : 1) * sizeof(uint16_t)));
        if (!*keys || !*ham) {
            std::free(*keys);
            std::free(*ham);
            *keys = nullptr;
            *ham = nullptr;
            return fail(c, SGMX_ERR_OOM, "host allocation failed");
        }
    }
    if (e == hipSuccess && total > 0) {
        e = ham_buffer(c, HAM_IDX, (size_t)total * 2 * sizeof(int), (void **)&d_idx);
        if (e == hipSuccess)
            e = ham_buffer(c, HAM_H, (size_t)total * sizeof(uint16_t), (void **)&d_h);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(ham_emit_blocked_kernel, grid, dim3(256), 0, st, d_a, d_b, g, nitems, ncodes, ht, d_cnt, d_idx, d_h);
            e = hipMemcpyAsync(*keys, d_idx, (size_t)total * 2 * sizeof(int), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(*ham, d_h, (size_t)total * sizeof(uint16_t), hipMemcpyDeviceToHost, st);
        const hipError_t e2 = hipStreamSynchronize(st);
        if (e == hipSuccess)
            e = e2;
    }
    if (e != hipSuccess) {
        std::free(*keys);
        std::free(*ham);
        *keys = nullptr;
        *ham = nullptr;
        return fail(c, e == hipErrorOutOfMemory ? SGMX_ERR_OOM : SGMX_ERR_HIP, std::string("hamming match (blocked): ") + hipGetErrorString(e));
    }
    *nmatches = (size_t)total;
    return SGMX_OK;
}

void sgmx_free(void *ptr) { std::free(ptr); }

static int get_stages(sgmx_ctx *c, int mode, int index, void *C, void *S, uint16_t *raw)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (c->mode != mode || c->last_n < 1)
        return fail(c, SGMX_ERR_STATE, "no matching compute call precedes this tap");
    if (index < 0 || index >= c->last_n)
        return fail(c, SGMX_ERR_INVALID_ARG, "pair index outside the last launch group");
    HIPCHK(c, hipSetDevice(c->prm.device));
    HIPCHK(c, hipDeviceSynchronize());
    const int D = c->prm.disp_range, Dp = (mode == 1 && c->fast) ? c->DpF : c->Dp;
    const size_t npix = (size_t)c->w * c->h, tot = npix * D;
    const size_t esz_c = mode == 1 ? 1 : 4, esz_s = mode == 1 ? 2 : 4;
    void *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, tot * 4));
    hipStream_t st = c->own_stream;
    hipError_t e = hipSuccess;
    const unsigned nb = (unsigned)((tot + 255) / 256);
    const bool fast = mode == 1 && c->fast;
    if (C && mode == 1) {
        const uint8_t *src = (const uint8_t *)c->d_C + index * c->bytes_C;
        uint8_t *vol = nullptr;
        if (fast) { // this path keeps no volume: rebuild it from the census images
            if (hipMalloc((void **)&vol, npix * Dp) != hipSuccess || materialize_cost(c, index, vol, st) != SGMX_OK) {
                (void)hipFree(tmp);
                (void)hipFree(vol);
                return fail(c, SGMX_ERR_OOM, "stage tap: cannot rebuild the data-cost volume");
            }
            src = vol;
        }
        hipLaunchKernelGGL(repitch_kernel<uint8_t>, dim3(nb), dim3(256), 0, st, src, Dp, (uint8_t *)tmp, D, npix,
                           (uint8_t)0);
        e = hipMemcpyAsync(C, tmp, tot * esz_c, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
        else
            (void)hipStreamSynchronize(st);
        (void)hipFree(vol);
    }
    if (S && e == hipSuccess) {
        if (fast) {
            const SumArgs sa = sum_args(c, ctx_view(c));
            const size_t thr = npix * 16;
            const dim3 grid((unsigned)((thr + 255) / 256));
            if (c->RP == 1)
                hipLaunchKernelGGL(sums_tap_kernel<1>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
            else if (c->RP == 2)
                hipLaunchKernelGGL(sums_tap_kernel<2>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
            else if (c->RP == 4)
                hipLaunchKernelGGL(sums_tap_kernel<4>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
            else if (c->RP == 8)
                hipLaunchKernelGGL(sums_tap_kernel<8>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
            else if (c->RP == 12)
                hipLaunchKernelGGL(sums_tap_kernel<12>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
            else
                hipLaunchKernelGGL(sums_tap_kernel<16>, grid, dim3(256), 0, st, sa, index, (uint16_t *)tmp);
        } else if (mode == 1)
            hipLaunchKernelGGL(repitch_kernel<uint16_t>, dim3(nb), dim3(256), 0, st,
                               (const uint16_t *)((const uint8_t *)c->d_S + index * c->bytes_S), Dp, (uint16_t *)tmp, D,
                               npix, (uint16_t)0);
        else
            hipLaunchKernelGGL(repitch_kernel<float>, dim3(nb), dim3(256), 0, st,
                               (const float *)((const uint8_t *)c->d_S + index * c->bytes_S), Dp, (float *)tmp, D, npix,
                               0.f);
        e = hipMemcpyAsync(S, tmp, tot * esz_s, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
            e = hipStreamSynchronize(st);
    }
    if (raw && e == hipSuccess)
        e = hipMemcpy(raw, c->d_raw + (size_t)index * c->W * c->H, (size_t)c->W * c->H * 2, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("stage tap: ") + hipGetErrorString(e));
    return SGMX_OK;
}

int sgmx_get_stages_census(sgmx_ctx *c, int index, uint8_t *C, uint16_t *S, uint16_t *raw_disp)
{
    return get_stages(c, 1, index, C, S, raw_disp);
}

int sgmx_get_stages_f32(sgmx_ctx *c, int index, float *S, uint16_t *raw_disp)
{
    return get_stages(c, 2, index, nullptr, S, raw_disp);
}

int sgmx_get_stages_sgmstereo(sgmx_ctx *c, int index, float *S_left, int32_t *left_disp, int32_t *right_disp,
                              uint8_t *states, uint16_t *raw_disp)
{
    int rc = get_stages(c, 3, index, nullptr, S_left, raw_disp);
    if (rc)
        return rc;
    const size_t npix = (size_t)c->w * c->h;
    if (left_disp)
        HIPCHK(c, hipMemcpy(left_disp, c->d_ld + index * npix, npix * 4, hipMemcpyDeviceToHost));
    if (right_disp)
        HIPCHK(c, hipMemcpy(right_disp, c->d_rd + index * npix, npix * 4, hipMemcpyDeviceToHost));
    if (states)
        HIPCHK(c, hipMemcpy(states, c->d_states + index * npix, npix, hipMemcpyDeviceToHost));
    return SGMX_OK;
}

int sgmx_get_data_cost_wta_u8(sgmx_ctx *c, int index, uint8_t *out)
{
    if (!c || !out)
        return fail(c, SGMX_ERR_INVALID_ARG, "null pointer");
    if (c->mode != 1 || c->last_n < 1)
        return fail(c, SGMX_ERR_STATE, "no census compute call precedes this tap");
    if (index < 0 || index >= c->last_n)
        return fail(c, SGMX_ERR_INVALID_ARG, "pair index outside the last launch group");
    HIPCHK(c, hipSetDevice(c->prm.device));
    HIPCHK(c, hipDeviceSynchronize());
    uint8_t *d_o = nullptr;
    const size_t nfull = (size_t)c->W * c->H;
    HIPCHK(c, hipMalloc((void **)&d_o, nfull));
    hipStream_t st = c->own_stream;
    hipError_t e = hipMemsetAsync(d_o, 0, nfull, st);
    const uint8_t *src = (const uint8_t *)c->d_C + index * c->bytes_C;
    uint8_t *vol = nullptr;
    if (c->fast) {
        if (hipMalloc((void **)&vol, (size_t)c->w * c->h * c->DpF) != hipSuccess || materialize_cost(c, index, vol, st) != SGMX_OK) {
            (void)hipFree(d_o);
            (void)hipFree(vol);
            return fail(c, SGMX_ERR_OOM, "data-cost WTA tap: cannot rebuild the data-cost volume");
        }
        src = vol;
    }
    hipLaunchKernelGGL(data_cost_wta_kernel, dim3((c->w + 255) / 256, c->h), dim3(256), 0, st, src, c->w, c->h,
                       c->prm.disp_range, c->fast ? c->DpF : c->Dp, c->m, d_o);
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d_o, nfull, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(d_o);
    (void)hipFree(vol);
    if (e != hipSuccess || e2 != hipSuccess)
        return fail(c, SGMX_ERR_HIP, std::string("data-cost WTA tap: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SGMX_OK;
}

int sgmx_last_launch_group(const sgmx_ctx *c, int *first_pair, int *count)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (first_pair)
        *first_pair = c->last_first;
    if (count)
        *count = c->last_n;
    return SGMX_OK;
}

int sgmx_set_profiling(sgmx_ctx *c, int enabled)
{
    if (!c)
        return SGMX_ERR_INVALID_ARG;
    if (enabled < 0 || enabled > SGMX_PROFILE_SWEEPS)
        return fail(c, SGMX_ERR_INVALID_ARG, "bad profiling mode");
    c->profiling = enabled;
    return SGMX_OK;
}

// Synchronise and fold the recorded event pairs into the per-class and per-direction sums.
static int collect_profile(sgmx_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->prm.device));
    HIPCHK(c, hipDeviceSynchronize());
    for (auto &r : c->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->prof_ms[r.cls] += ms;
            c->prof_n[r.cls] += 1;
            if (r.dir >= 0 && r.dir < 16) {
                c->prof_dir_ms[r.dir] += ms;
                c->prof_dir_n[r.dir] += 1;
            }
        }
    }
    // (only now: in the chained mode a record's first event is the previous record's second one)
    for (auto &r : c->prof) {
        if (r.owns_a)
            (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    c->prof.clear();
    return SGMX_OK;
}

int sgmx_direction_stats(sgmx_ctx *c, int k, double *total_ms, long *launches)
{
    if (!c || k < 0 || k >= 16)
        return fail(c, SGMX_ERR_INVALID_ARG, "bad direction index");
    const int rc = collect_profile(c);
    if (rc)
        return rc;
    if (total_ms)
        *total_ms = c->prof_dir_ms[k];
    if (launches)
        *launches = c->prof_dir_n[k];
    c->prof_dir_ms[k] = 0;
    c->prof_dir_n[k] = 0;
    return SGMX_OK;
}

int sgmx_kernel_stats(sgmx_ctx *c, int cls, double *total_ms, long *launches)
{
    if (!c || cls < 0 || cls >= SGMX_K_COUNT)
        return fail(c, SGMX_ERR_INVALID_ARG, "bad kernel class");
    const int rc = collect_profile(c);
    if (rc)
        return rc;
    if (total_ms)
        *total_ms = c->prof_ms[cls];
    if (launches)
        *launches = c->prof_n[cls];
    c->prof_ms[cls] = 0;
    c->prof_n[cls] = 0;
    return SGMX_OK;
}

double sgmx_algorithmic_bytes_census(const sgmx_ctx *c, int W, int H, double *aggregate_only)
{
    if (!c)
        return 0.0;
    const int m = (c->prm.window_sz - 1) / 2;
    const double V = (double)(W - 2 * m) * (H - 2 * m) * c->prm.disp_range;
    const int N = c->prm.num_paths;
    if (aggregate_only)
        *aggregate_only = V * (5.0 * N - 2.0); // N reads of C, S written once then N-1 read-modify-writes
    return V * (5.0 * N + 1.0) + 10.0 * W * H;
}

} // extern "C"
