// Checks on gfx950: (1) buffer_load_dwordx4 ... lds lands lane L's 16 bytes at LDS base + 16 L; the dword form at base + 4 L
// (lanes masked off by EXEC leave their slots alone); (2) ds_read_b128 from a 4-byte aligned LDS address returns the
// right 16 bytes.      hipcc --offload-arch=gfx950 -O3 -o exp/bin/lds_direct_check exp/lds_direct_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void k(const u32 *in, u32 *out)
{
    __shared__ __attribute__((aligned(16))) u32 buf[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64)
        buf[i] = 0xDEAD0000u + i;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32 *>(in), 0, -1, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, &buf[0], 16, lane * 16, 0, 0, 0);        // words 0..255 <- in[0..255]
    if (lane < 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, &buf[256], 4, lane * 4, 4096, 0, 0); // words 256..259 <- in[1024..1027]
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 512; i += 64)
        out[i] = buf[i];
    // unaligned 16-byte LDS read: lane L reads words L+1 .. L+4
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((u32)(size_t)&buf[lane + 1]) : "memory");
    out[512 + 4 * lane] = v.x; out[513 + 4 * lane] = v.y; out[514 + 4 * lane] = v.z; out[515 + 4 * lane] = v.w;
}
int main()
{
    u32 *in, *out, h[2048], o[1024];
    for (int i = 0; i < 2048; i++) h[i] = 1000 + i;
    CHK(hipMalloc(&in, sizeof h)); CHK(hipMalloc(&out, sizeof o));
    CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
    CHK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; i++) bad += o[i] != 1000u + i;
    for (int i = 0; i < 4; i++) bad += o[256 + i] != 2024u + i;
    for (int i = 260; i < 512; i++) bad += o[i] != 0xDEAD0000u + i;
    printf("lds-direct layout: %s (first words %u %u %u %u ... %u %u | %u %u %u %u %x)\n", bad ? "UNEXPECTED" : "as expected", o[0], o[1], o[2],
           o[3], o[254], o[255], o[256], o[257], o[258], o[259], o[260]);
    int bad2 = 0;
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 4; j++)
            bad2 += o[512 + 4 * l + j] != 1000u + l + 1 + j;
    printf("unaligned ds_read_b128: %s (lane 1: %u %u %u %u)\n", bad2 ? "WRONG" : "right", o[516], o[517], o[518], o[519]);
    return bad || bad2;
}
