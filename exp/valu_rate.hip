// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU opcodes the sweep uses.
// Each kernel runs N independent dependency chains per lane; 8 waves per SIMD (grid fills the chip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int OP> __global__ __launch_bounds__(256) void k(u32 *out, u32 seed)
{
    u32 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    u32 b = seed ^ 0x5a5a5a5a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = a[i] ^ b;
            else if (OP == 1) a[i] = __builtin_popcount(a[i]) + b;
            else if (OP == 2) a[i] = __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, b)));
            else if (OP == 3) a[i] = __builtin_bit_cast(u32, (us2)(__builtin_bit_cast(us2, a[i]) + __builtin_bit_cast(us2, b)));
            else if (OP == 4) a[i] = min(a[i], (u32)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)a[i], 0x121, 0xF, 0xF, false));
            else if (OP == 5) a[i] = (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x121, 0xF, 0xF, true);
            else if (OP == 6) a[i] = (a[i] << 16) | b;
            else if (OP == 7) a[i] = __builtin_amdgcn_perm(a[i], b, 0x06040200u);
            else if (OP == 8) a[i] = __builtin_amdgcn_alignbit(a[i], b, 16);
            else if (OP == 9) a[i] = __builtin_bit_cast(u32, __builtin_fmaf(__builtin_bit_cast(float, a[i]), 1.0001f, 0.5f));
            else if (OP == 10) a[i] = a[i] + b;
            else if (OP == 11) a[i] = min(a[i], b);
            else if (OP == 12) a[i] = (a[i] > b) ? a[i] : (b ^ a[i]);  // cmp + cndmask(+xor)
        }
        if (OP == 6 || OP == 7 || OP == 8) b += 0x10001u;
    }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, u32 *d)
{
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)blocks * 4 / 1024 * ITERS * 8; // wave-instructions per SIMD
    printf("%-22s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main()
{
    u32 *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<1>("v_bcnt_u32_b32", d); run<2>("v_pk_min_u16", d); run<3>("v_pk_add_u16", d);
    run<4>("v_min_u32_dpp row_ror", d); run<5>("v_mov_b32_dpp row_ror", d); run<6>("v_lshl_or_b32", d); run<7>("v_perm_b32", d);
    run<8>("v_alignbit_b32", d); run<9>("v_fma_f32", d); run<10>("v_add_u32", d); run<11>("v_min_u32", d); run<12>("cmp+cndmask+xor", d);
    return 0;
}
