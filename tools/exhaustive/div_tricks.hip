// Exhaustive (all 2^32 bit patterns) check of the exact-division identities used by the per-point loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exhaustive/div_tricks.hip -o /tmp/div_tricks && /tmp/div_tricks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#include "../../rgbd_odometry_amd/csrc/dvo_device_math.h"     // the product's own functions are what is tested
using dvo::exact_rcp; using dvo::exact_div_z1; using dvo::exact_div_zz1; using dvo::rcp_in_proven_range;
#define fast_rcp exact_rcp
#define div_z1 exact_div_z1
#define div_zz1 exact_div_zz1
__device__ __forceinline__ bool same(float a, float b) {       // bit equality, any NaN == any NaN
    return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
}

__global__ void check(unsigned long long *cnt) {
    // cnt: 0 rcp mismatches (normal, |x| in [2^-126,2^126]), 1 rcp mismatches (all), 2 zn not in set, 3 div_z1 mismatch, 4 div_zz1 mismatch,
    //      5 zn==1 count, 6 zn==1-2^-24 count, 7 zz != expected
    const float z1 = __uint_as_float(0x3f7fffffu), zz1 = __uint_as_float(0x3f7ffffeu);
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float q = 1.0f / x;
        const float f = fast_rcp(x);
        const bool inrange = rcp_in_proven_range(x);
        if (!same(q, f)) { c[1]++; if (inrange) c[0]++; }
        if (inrange) {
            const float zn = x * q;
            if (zn == 1.0f) c[5]++; else if (zn == z1) c[6]++; else c[2]++;
            if (!same(z1 * z1, zz1)) c[7]++;
        }
        if (!same(x / z1, div_z1(x))) c[3]++;
        if (!same(x / zz1, div_zz1(x))) { if (c[4] < 4 && (i & 0x7fffffffull) < 0x7f800000ull) printf("zz1 mismatch x=%08x want=%08x got=%08x\n", (uint32_t)i, __float_as_uint(x / zz1), __float_as_uint(div_zz1(x))); c[4]++; }
    }
    for (int k = 0; k < 8; k++) if (c[k]) atomicAdd(&cnt[k], c[k]);
}

int main() {
    unsigned long long *d, h[8];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const bool ok = (h[0] == 0) && (h[2] == 0) && (h[3] == 0) && (h[4] == 0) && (h[7] == 0);
    printf("%s\n", ok ? "ALL IDENTITIES HOLD" : "IDENTITY VIOLATED");
    printf("rcp+newton != 1/x : %llu in normal range, %llu over all bit patterns\n", h[0], h[1]);
    printf("x*fl(1/x): ==1: %llu, ==1-2^-24: %llu, other: %llu   (z1*z1 != 1-2^-23: %llu)\n", h[5], h[6], h[2], h[7]);
    printf("x/(1-2^-24) trick mismatches: %llu\n", h[3]);
    printf("x/(1-2^-23) trick mismatches: %llu\n", h[4]);
    return ok ? 0 : 1;
}
