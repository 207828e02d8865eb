// HBM read ceiling as a plain streaming kernel sees it: each wave reads contiguous 1 KB pieces (16 bytes per lane),
// UNROLL loads in flight, xor-reduces them and writes one dword per wave at the end.
//   hipcc --offload-arch=gfx950 -O3 -o exp/bin/read_ceiling exp/read_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4 *__restrict__ in, size_t n16, u32 *__restrict__ out)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    u32 acc = 0;
    for (size_t i = tid; i + (UNROLL - 1) * nth < n16; i += UNROLL * nth) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            v[u] = NT ? __builtin_nontemporal_load(in + i + u * nth) : in[i + u * nth];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u)
        out[tid] = acc;
}

template <int UNROLL, bool NT> void run(const u32x4 *buf, size_t bytes, u32 *out, int blocks)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out);
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, out);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("blocks %5d unroll %d %s: %7.3f ms  %5.2f TB/s\n", blocks, UNROLL, NT ? "nt   " : "plain", ms, bytes / ms / 1e9);
}

int main()
{
    const size_t bytes = (size_t)15 << 30;
    u32x4 *buf;
    u32 *out;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMalloc(&out, 64 << 20));
    CHK(hipMemset(buf, 1, bytes));
    for (int blocks : {2048, 4096, 8192, 16384}) {
        run<4, false>(buf, bytes, out, blocks);
        run<4, true>(buf, bytes, out, blocks);
        run<8, true>(buf, bytes, out, blocks);
    }
    return 0;
}
