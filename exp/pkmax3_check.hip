// Does v_pk_maximum3_f16 / v_pk_minimum3_f16 (gfx950) act as an exact 3-input packed uint16 max / min on bit
// patterns below 0x7C00 (positive finite halves, denormals included: patterns 0..0x3FF)?  The sweeps would use
// it on path-cost terms <= 255, all of them f16 denormals, so a flush-to-zero anywhere would show up here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
__global__ void k(const u32 *a, const u32 *b, const u32 *c, u32 *mx, u32 *mn, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 x, y;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(x) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(y) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    mx[i] = x;
    mn[i] = y;
}
int main()
{
    const int n = 1 << 22;
    std::vector<u32> a(n), b(n), c(n), mx(n), mn(n);
    u32 s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int i = 0; i < n; i++) {
        // thirds: tiny (<= 255), denormal range (< 0x400), anything below 0x7C00
        const u32 lim = i % 3 == 0 ? 256 : (i % 3 == 1 ? 0x400 : 0x7C00);
        auto h = [&]() { return rnd() % lim; };
        a[i] = h() | (h() << 16); b[i] = h() | (h() << 16); c[i] = h() | (h() << 16);
    }
    u32 *da, *db, *dc, *dx, *dn;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dx, n * 4); hipMalloc(&dn, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dx, dn, n);
    hipMemcpy(mx.data(), dx, n * 4, hipMemcpyDeviceToHost); hipMemcpy(mn.data(), dn, n * 4, hipMemcpyDeviceToHost);
    long bad_max = 0, bad_min = 0;
    for (int i = 0; i < n; i++) {
        auto m3 = [&](bool mxx, int sh) {
            u32 x = (a[i] >> sh) & 0xFFFF, y = (b[i] >> sh) & 0xFFFF, z = (c[i] >> sh) & 0xFFFF;
            u32 r = mxx ? (x > y ? x : y) : (x < y ? x : y);
            return mxx ? (r > z ? r : z) : (r < z ? r : z);
        };
        const u32 emx = m3(true, 0) | (m3(true, 16) << 16), emn = m3(false, 0) | (m3(false, 16) << 16);
        if (mx[i] != emx) { if (bad_max++ < 5) printf("max3 %08x %08x %08x -> %08x want %08x\n", a[i], b[i], c[i], mx[i], emx); }
        if (mn[i] != emn) { if (bad_min++ < 5) printf("min3 %08x %08x %08x -> %08x want %08x\n", a[i], b[i], c[i], mn[i], emn); }
    }
    printf("v_pk_maximum3_f16 as packed umax3 on patterns < 0x7C00: %ld mismatches of %d\n", bad_max, n);
    printf("v_pk_minimum3_f16 as packed umin3 on patterns < 0x7C00: %ld mismatches of %d\n", bad_min, n);
    return (bad_max || bad_min) ? 1 : 0;
}
