// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU opcodes the sweeps
// (aggregate3_kernel) are made of, at exactly 1, 2, 4 and 8 waves per SIMD.
//
// Residency is pinned, not hoped for: every workgroup asks for so much LDS that only the intended
// number fits a CU (160 KiB: one workgroup of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD;
// 80 KiB: two workgroups of 1024 threads = 8 waves per SIMD), and the grid is exactly that many
// workgroups per CU, so all waves run side by side from start to end.  Each wave runs 8 independent
// dependency chains per lane (all operands in VGPRs, every instruction through asm volatile so that
// nothing is folded away).  Two readings per cell:
//   stamp = max over waves of (s_memtime delta) / instructions issued on that wave's SIMD meanwhile
//   wall  = kernel duration (HIP events) x in-kernel clock / instructions per SIMD   (includes the
//           launch and the ramp: an upper bound that the stamps should approach)
// The clock is s_memtime / s_memrealtime (100 MHz), so nothing here assumes a frequency.
//   hipcc --offload-arch=gfx950 -O3 -o exp/bin/valu_rate exp/valu_rate.hip && exp/bin/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define ITERS 8192
#define CHK(x)                                                                                                         \
    do {                                                                                                               \
        hipError_t e_ = (x);                                                                                           \
        if (e_ != hipSuccess) {                                                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                                                             \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

template <int OP> __global__ __launch_bounds__(1024) void k(u32 *out, u64 *stamps, u32 seed)
{
    extern __shared__ u32 lds[]; // only there to pin the residency
    u32 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        a[i] = seed * (i + 1) + threadIdx.x;
    u32 b = seed ^ 0x5a5a5a5a ^ threadIdx.x;
    if (seed == 0xdeadbeef)
        lds[threadIdx.x] = b;
    const u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 2) asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 3) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 4) asm volatile("v_min_u32_dpp %0, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[i]));
            else if (OP == 6) asm volatile("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 7) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(b), "s"(0x06040200u));
            else if (OP == 8) asm volatile("v_alignbit_b32 %0, %1, %1, 16" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 9) asm volatile("v_fma_f32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 10) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 11) asm volatile("v_min_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 12) asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 13) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 15) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 16) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 17) asm volatile("v_or_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 18) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 19) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 20) asm volatile("v_max_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 21) asm volatile("v_min_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 22) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 24) asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 25) asm volatile("v_pk_minimum3_f16 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 26) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 27) asm volatile("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 28) asm volatile("v_min3_u32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 29) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 30) asm volatile("v_and_or_b32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 31) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 32) asm volatile("v_add_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 33) asm volatile("v_min_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 34) asm volatile("v_bitop3_b32 %0, %1, %2, %2 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 35) asm volatile("v_add3_u32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 36) asm volatile("v_xor_b32_dpp %0, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 37) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 38) asm volatile("v_sad_u8 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 39) asm volatile("v_mad_u32_u24 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 40) asm volatile("v_xnor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b));
            else if (OP == 41) asm volatile("v_not_b32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            else if (OP == 42) asm volatile("v_pk_max_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(a[i]) : "v"(a[i]));
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        s ^= a[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[2 * wv] = t1 - t0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

static u32 *d_out;
static u64 *d_st;
static int g_cus = 256;

template <int OP> int run(const char *name)
{
    printf("%-24s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int threads = wps >= 4 ? 1024 : 256 * wps;
        const int per_cu = wps == 8 ? 2 : 1;
        const int blocks = g_cus * per_cu;
        const size_t lds = wps == 8 ? 80 * 1024 : 160 * 1024;
        CHK(hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, d_out, d_st, 12345u);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, d_out, d_st, 12345u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const size_t nw = (size_t)blocks * threads / 64;
        std::vector<u64> st(2 * nw);
        CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, ghz;
        for (size_t i = 0; i < nw; i++) {
            cyc.push_back((double)st[2 * i]);
            ghz.push_back((double)st[2 * i] / ((double)st[2 * i + 1] * 10.0)); // 100 MHz ticks -> ns
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const double c = cyc[nw - 1], g = ghz[nw / 2]; // the slowest wave: the arbiter favours older waves, so the
                                                       // waves of a SIMD end at different times
        const double per_simd = (double)ITERS * 8 * wps; // wave-instructions a SIMD issues in the run
        printf(" | %4.2f %4.2f %4.2fGHz", c / per_simd, ms * 1e6 * g / per_simd, g);
        CHK(hipEventDestroy(e0));
        CHK(hipEventDestroy(e1));
    }
    printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    CHK(hipMalloc(&d_out, (size_t)g_cus * 2 * 1024 * 4));
    CHK(hipMalloc(&d_st, (size_t)g_cus * 2 * 16 * 2 * 8));
    printf("%s, %d CUs.  Cycles per wave64 VALU instruction per SIMD: stamp wall clock, at 1 | 2 | 4 | 8 waves per SIMD\n",
           p.gcnArchName, g_cus);
    int rc = 0;
    rc |= run<0>("v_xor_b32");
    rc |= run<40>("v_xnor_b32");
    rc |= run<41>("v_not_b32");
    rc |= run<10>("v_add_u32");
    rc |= run<11>("v_min_u32");
    rc |= run<14>("v_mov_b32");
    rc |= run<15>("v_add_f32");
    rc |= run<9>("v_fma_f32");
    rc |= run<1>("v_bcnt_u32_b32");
    rc |= run<2>("v_pk_min_u16");
    rc |= run<3>("v_pk_add_u16");
    rc |= run<12>("v_pk_sub_u16");
    rc |= run<13>("v_pk_max_u16");
    rc |= run<42>("v_pk_max_u16 op_sel swap");
    rc |= run<4>("v_min_u32_dpp row_ror:1");
    rc |= run<5>("v_mov_b32_dpp row_shr:1");
    rc |= run<6>("v_lshl_add_u32");
    rc |= run<7>("v_perm_b32");
    rc |= run<8>("v_alignbit_b32");
    printf("-- other opcodes, for the classes\n");
    rc |= run<16>("v_and_b32");
    rc |= run<17>("v_or_b32");
    rc |= run<18>("v_sub_u32");
    rc |= run<19>("v_lshlrev_b32");
    rc |= run<20>("v_max_u32");
    rc |= run<21>("v_min_f32");
    rc |= run<22>("v_mul_f32");
    rc |= run<23>("v_fmac_f32");
    rc |= run<32>("v_add_u16");
    rc |= run<33>("v_min_u16");
    rc |= run<29>("v_cndmask_b32");
    rc |= run<28>("v_min3_u32");
    rc |= run<30>("v_and_or_b32");
    rc |= run<35>("v_add3_u32");
    rc |= run<34>("v_bitop3_b32");
    rc |= run<39>("v_mad_u32_u24");
    rc |= run<38>("v_sad_u8");
    rc |= run<36>("v_xor_b32_dpp row_ror:1");
    rc |= run<24>("v_pk_min_f16");
    rc |= run<31>("v_pk_add_f16");
    rc |= run<37>("v_pk_mul_lo_u16");
    rc |= run<27>("v_pk_sub_u16 clamp");
    rc |= run<25>("v_pk_minimum3_f16");
    rc |= run<26>("v_pk_maximum3_f16");
    return rc;
}
