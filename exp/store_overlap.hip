// Do streaming stores overlap with VALU work on gfx950?  4096 waves (4 per SIMD), each: T steps of K slow-class
// VALU instructions followed by one 16-byte-per-lane non-temporal buffer store (1 KB per wave and step), i.e. the
// shape of a D = 256 direction sweep without its loads.  Prints ms and TB/s per K, for a streaming and a
// row-strided address pattern.   hipcc --offload-arch=gfx950 -O3 -o exp/bin/store_overlap exp/store_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int K, int AUX>
__global__ __launch_bounds__(256) void k(unsigned char *out, int T, long wave_stride, long step_stride)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + wave * wave_stride, 0, -1, 0x00020000);
    u32 a = lane, b = lane * 3 + 1, c = 7, d = 9;
    unsigned so = 0;
    for (int t = 0; t < T; t++) {
#pragma unroll
        for (int i = 0; i < K / 4; i++) {
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(b) : "v"(c));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(c) : "v"(d));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(d) : "v"(a));
        }
        u32x4 v;
        v.x = a; v.y = b; v.z = c; v.w = d;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, so, AUX);
        so += (unsigned)step_stride;
    }
}

// the same with one dword load per step from a small cache-resident table, consumed PF steps later: the wave now
// waits (s_waitcnt vmcnt) for a load that was issued before the last PF stores -- as the sweeps do
template <int K, int PF>
__global__ __launch_bounds__(256) void kl(unsigned char *out, const u32 *tab, int T, long wave_stride, long step_stride)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + wave * wave_stride, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32 *>(tab), 0, -1, 0x00020000);
    u32 a = lane, b = lane * 3 + 1, c = 7, d = 9;
    unsigned so = 0, sl = 0;
    u32 q[PF];
#pragma unroll
    for (int i = 0; i < PF; i++)
        q[i] = __builtin_amdgcn_raw_buffer_load_b32(rt, (lane >> 4) * 4, 16 * i, 0);
    for (int t = 0; t < T; t += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
            a ^= q[i];
            q[i] = __builtin_amdgcn_raw_buffer_load_b32(rt, (lane >> 4) * 4, sl, 0);
            sl = (sl + 16) & 0xFFFF;
#pragma unroll
            for (int j = 0; j < K / 4; j++) {
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(b) : "v"(c));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(c) : "v"(d));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(d) : "v"(a));
            }
            u32x4 v;
            v.x = a; v.y = b; v.z = c; v.w = d;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, so, 2);
            so += (unsigned)step_stride;
        }
    }
}
template <int K, int PF> float runl(unsigned char *buf, const u32 *tab, int T, long ws, long ss)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kl<K, PF>), dim3(1024), dim3(256), 0, 0, buf, tab, T, ws, ss);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL((kl<K, PF>), dim3(1024), dim3(256), 0, 0, buf, tab, T, ws, ss);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

// the vertical D = 256 sweep's memory side: per step four 16-byte loads per lane (windows of a census row that stays in
// cache: 4 KB per wave and step through the L1 path), consumed PF steps later, and the 1 KB store
template <int K, int PF, bool ROWS = false>
__global__ __launch_bounds__(256) void kv(unsigned char *out, const u32 *tab, int T, long wave_stride, long step_stride)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + wave * wave_stride, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32 *>(tab), 0, -1, 0x00020000);
    u32 a = lane, b = lane * 3 + 1, c = 7, d = 9;
    // ROWS: the waves of one XCD (blockIdx % 8: a "pair") read their windows from one 16 KB "census row" per step, a
    // new row every step (32 MB per pair and launch: the rows miss the XCD's 4 MB L2 on first touch, like a 4K image)
    const unsigned pairw = (unsigned)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    unsigned so = 0, sl = ROWS ? (unsigned)(blockIdx.x & 7) * (32u << 20) : (unsigned)(wave & 63) * 1024;
    const unsigned vo = (lane & 15) * 32 + (lane >> 4) * 4 + (ROWS ? pairw * 16 : 0);
    constexpr unsigned LSTEP = ROWS ? 16384 : 4096;
    u32x4 q[PF][4];
#pragma unroll
    for (int i = 0; i < PF; i++)
#pragma unroll
        for (int m = 0; m < 4; m++)
            q[i][m] = __builtin_amdgcn_raw_buffer_load_b128(rt, vo + 16 * (m & 1) + 1024 * (m >> 1), sl + LSTEP * i, 0);
    sl += LSTEP * (PF - 1);
    for (int t = 0; t < T; t += PF) {
#pragma unroll
        for (int i = 0; i < PF; i++) {
#pragma unroll
            for (int m = 0; m < 4; m++)
                a ^= q[i][m].x ^ q[i][m].y ^ q[i][m].z ^ q[i][m].w;
            sl = ROWS ? sl + LSTEP : ((sl + 4096) & 0xFFFFF); // not ROWS: a 1 MB table that stays in L2
#pragma unroll
            for (int m = 0; m < 4; m++)
                q[i][m] = __builtin_amdgcn_raw_buffer_load_b128(rt, vo + 16 * (m & 1) + 1024 * (m >> 1), sl, 0);
#pragma unroll
            for (int j = 0; j < (K - 16) / 4; j++) {
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(b) : "v"(c));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(c) : "v"(d));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(d) : "v"(a));
            }
            u32x4 v;
            v.x = a; v.y = b; v.z = c; v.w = d;
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, so, 2);
            so += (unsigned)step_stride;
        }
    }
}
template <int K, int PF, bool ROWS = false> float runv(unsigned char *buf, const u32 *tab, int T, long ws, long ss)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((kv<K, PF, ROWS>), dim3(1024), dim3(256), 0, 0, buf, tab, T, ws, ss);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL((kv<K, PF, ROWS>), dim3(1024), dim3(256), 0, 0, buf, tab, T, ws, ss);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

// the D = 128 shape: 8 bytes per lane and step (512 B per wave and step)
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
template <int K>
__global__ __launch_bounds__(256) void k64(unsigned char *out, int T, long wave_stride, long step_stride)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + wave * wave_stride, 0, -1, 0x00020000);
    u32 a = lane, b = lane * 3 + 1, c = 7, d = 9;
    unsigned so = 0;
    for (int t = 0; t < T; t++) {
#pragma unroll
        for (int i = 0; i < K / 4; i++) {
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(b));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(b) : "v"(c));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(c) : "v"(d));
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(d) : "v"(a));
        }
        u32x2 v;
        v.x = a ^ c; v.y = b ^ d;
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, lane * 8, so, 2);
        so += (unsigned)step_stride;
    }
}
template <int K> float run64(unsigned char *buf, int T, long ws, long ss)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k64<K>), dim3(1024), dim3(256), 0, 0, buf, T, ws, ss);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL((k64<K>), dim3(1024), dim3(256), 0, 0, buf, T, ws, ss);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

template <int K, int AUX> float run(unsigned char *buf, int T, long ws, long ss)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<K, AUX>), dim3(1024), dim3(256), 0, 0, buf, T, ws, ss);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL((k<K, AUX>), dim3(1024), dim3(256), 0, 0, buf, T, ws, ss);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

int main()
{
    const int T = 2048;
    const size_t bytes = (size_t)4096 * T * 1024; // 8.6 GB
    unsigned char *buf;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMemset(buf, 0, bytes));
    printf("4096 waves x %d steps x 1 KB = %.2f GB per launch\n", T, bytes / 1e9);
    // streaming: a wave owns T consecutive KB;  strided: step t of all waves forms one 4 MB "row" (like a vertical sweep)
    const long ws_seq = (long)T * 1024, ss_seq = 1024, ws_row = 1024, ss_row = (long)4096 * 1024;
#define ROW(K)                                                                                                        \
    {                                                                                                                  \
        const float a = run<K, 2>(buf, T, ws_seq, ss_seq), b = run<K, 2>(buf, T, ws_row, ss_row),                     \
                    c = run<K, 0>(buf, T, ws_row, ss_row);                                                             \
        printf("K = %3d VALU per step: streaming nt %7.3f ms %5.2f TB/s | rows nt %7.3f ms %5.2f TB/s | rows plain %7.3f ms %5.2f TB/s\n", \
               K, a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9);                                          \
    }
    ROW(0) ROW(16) ROW(32) ROW(64) ROW(88) ROW(128) ROW(192)
    u32 *tab;
    CHK(hipMalloc(&tab, 1 << 17));
    CHK(hipMemset(tab, 1, 1 << 17));
#define ROWL(K)                                                                                                       \
    {                                                                                                                  \
        const float a = runl<K, 2>(buf, tab, T, ws_row, ss_row), b = runl<K, 8>(buf, tab, T, ws_row, ss_row),         \
                    c = runl<K, 16>(buf, tab, T, ws_row, ss_row);                                                      \
        printf("K = %3d + a load per step used PF steps later, rows nt: PF 2 %7.3f ms %5.2f TB/s | PF 8 %7.3f ms %5.2f TB/s | PF 16 %7.3f ms %5.2f TB/s\n", \
               K, a, bytes / a / 1e9, b, bytes / b / 1e9, c, bytes / c / 1e9);                                          \
    }
    ROWL(0) ROWL(32) ROWL(64) ROWL(88) ROWL(128)
    {
        u32 *big;
        const size_t bigb = ((size_t)8 * 32 << 20) + ((size_t)(2048 + 8) * 16384);
        CHK(hipMalloc(&big, bigb));
        CHK(hipMemset(big, 1, bigb));
        const float a = runv<88, 2>(buf, big, T, ws_row, ss_row), b = runv<88, 4>(buf, big, T, ws_row, ss_row);
        printf("K =  88 (16 of them the xors of the loaded words) + 4 x 16-byte loads per lane and step (4 KB per wave, L2 resident), rows nt: "
               "PF 2 %7.3f ms %5.2f TB/s | PF 4 %7.3f ms %5.2f TB/s\n", a, bytes / a / 1e9, b, bytes / b / 1e9);
        const float c = runv<88, 2, true>(buf, big, T, ws_row, ss_row), d = runv<88, 4, true>(buf, big, T, ws_row, ss_row);
        printf("    the same, the windows from a new 16 KB row per step and XCD (first touch misses L2): "
               "PF 2 %7.3f ms %5.2f TB/s | PF 4 %7.3f ms %5.2f TB/s\n", c, bytes / c / 1e9, d, bytes / d / 1e9);
    }
    {
        const int T2 = 4096;
        const double by = (double)4096 * T2 * 512;
#define ROW64(K)                                                                                                      \
    {                                                                                                                  \
        const float a = run64<K>(buf, T2, (long)T2 * 512, 512), b = run64<K>(buf, T2, 512, (long)4096 * 512);          \
        printf("K = %3d VALU per step, 512 B per wave and step: streaming nt %7.3f ms %5.2f TB/s | rows nt %7.3f ms %5.2f TB/s\n", K, a, \
               by / a / 1e9, b, by / b / 1e9);                                                                         \
    }
        ROW64(0) ROW64(24) ROW64(40) ROW64(48) ROW64(56) ROW64(64) ROW64(96)
    }
    return 0;
}
