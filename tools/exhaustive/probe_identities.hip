// Exhaustive probes of CANDIDATE exact identities for the round-2 per-point loop (prototype; the ones that hold move
// into dvo_device_math.h and are then pinned by div_tricks.hip).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exhaustive/probe_identities.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ bool same(float a, float b) {
    return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
}
__device__ __forceinline__ double rcp64(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ float weight_ref(float r) { return (float)(6.0 / (6.0 + (double)(r * r) / .25)); }

__device__ __forceinline__ float w_v1(float r) {   // rcp_f64 + 2 Newton + quotient correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = rcp64(D);
    y = fma(fma(-D, y, 1.0), y, y);
    y = fma(fma(-D, y, 1.0), y, y);
    const double q = 6.0 * y;
    return (float)fma(fma(-D, q, 6.0), y, q);
}
__device__ __forceinline__ float w_v2(float r) {   // rcp_f64 + 1 Newton + quotient correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = rcp64(D);
    y = fma(fma(-D, y, 1.0), y, y);
    const double q = 6.0 * y;
    return (float)fma(fma(-D, q, 6.0), y, q);
}
__device__ __forceinline__ float w_v3(float r) {   // f32 rcp seed + 2 Newton, no correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = (double)__builtin_amdgcn_rcpf((float)D);
    y = fma(fma(-D, y, 1.0), y, y);
    y = fma(fma(-D, y, 1.0), y, y);
    return (float)(6.0 * y);
}
__device__ __forceinline__ float w_v4(float r) {   // f32 rcp seed + 1 Newton + quotient correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = (double)__builtin_amdgcn_rcpf((float)D);
    y = fma(fma(-D, y, 1.0), y, y);
    const double q = 6.0 * y;
    return (float)fma(fma(-D, q, 6.0), y, q);
}
__device__ __forceinline__ float w_v5(float r) {   // rcp_f64 + 1 Newton, no correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = rcp64(D);
    y = fma(fma(-D, y, 1.0), y, y);
    return (float)(6.0 * y);
}
__device__ __forceinline__ float w_v6(float r) {   // rcp_f64 + 2 Newton, no correction
    const double D = fma(4.0, (double)(r * r), 6.0);
    double y = rcp64(D);
    y = fma(fma(-D, y, 1.0), y, y);
    y = fma(fma(-D, y, 1.0), y, y);
    return (float)(6.0 * y);
}
__device__ __forceinline__ int cvt_flr(float u) {
    int i;
    asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(u));
    return i;
}

__global__ void check(unsigned long long *cnt) {
    const float zz1 = __uint_as_float(0x3f7ffffeu), z1 = __uint_as_float(0x3f7fffffu);
    const float C23 = __uint_as_float(0x34000001u);      // 2^-23 + 2^-46
    const float C24 = __uint_as_float(0x33800001u);      // 2^-24 + 2^-47 ... (1+2^-23)*2^-24
    unsigned long long c[16] = {0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const bool finite = ((uint32_t)i & 0x7f800000u) != 0x7f800000u;
        if (!same(x / zz1, __builtin_fmaf(x, C23, x))) { c[0]++; if (finite) { c[1]++; if (c[1] < 3) printf("zz1 fma mismatch x=%08x want=%08x got=%08x\n", (uint32_t)i, __float_as_uint(x / zz1), __float_as_uint(__builtin_fmaf(x, C23, x))); } }
        if (!same(x / z1, __builtin_fmaf(x, C24, x))) { c[2]++; if (finite) { c[3]++; if (c[3] < 3) printf("z1 fma mismatch x=%08x want=%08x got=%08x\n", (uint32_t)i, __float_as_uint(x / z1), __float_as_uint(__builtin_fmaf(x, C24, x))); } }
        if (fabsf(x) <= 1048576.0f) {      /* |r| <= 2^20 (distance-transform values are 0..255) */
            const float wr = weight_ref(x);
            if (!same(wr, w_v1(x))) c[4]++;
            if (!same(wr, w_v2(x))) c[5]++;
            if (!same(wr, w_v3(x))) c[6]++;
            if (!same(wr, w_v4(x))) c[7]++;
            if (!same(wr, w_v5(x))) c[8]++;
            if (!same(wr, w_v6(x))) c[9]++;
        }
        if (x == x) {
            const int fl = cvt_flr(x);
            const float Cs[4] = {1.0f, 30.0f, 640.0f, 4096.0f};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool ref = (x >= 0.0f) && (x < Cs[k]);
                const bool alt = (unsigned)fl < (unsigned)(int)Cs[k];
                if (ref != alt) { c[10]++; if (c[10] < 3) printf("flr mismatch x=%08x C=%g fl=%d\n", (uint32_t)i, Cs[k], fl); }
                if (ref && fl != (int)x) c[11]++;
            }
        }
        if (i < 65536) {
            const float d = (float)(uint32_t)i;
            const float q0 = d * 0.001f;
            const float r = __builtin_fmaf(-q0, 1000.0f, d);
            const float q1 = __builtin_fmaf(r, 0.001f, q0);
            if (!same(d / 1000.0f, q1)) c[12]++;
            if (!same(d / 1000.0f, q0)) c[13]++;
        }
    }
    for (int k = 0; k < 16; k++) if (c[k]) atomicAdd(&cnt[k], c[k]);
}

int main() {
    unsigned long long *d, h[16];
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("x/(1-2^-23) == fma(x, 2^-23+2^-46, x): mismatches %llu (finite x: %llu)\n", h[0], h[1]);
    printf("x/(1-2^-24) == fma(x, (1+2^-23)2^-24, x): mismatches %llu (finite x: %llu)\n", h[2], h[3]);
    printf("weight variants mismatches: v1 %llu  v2 %llu  v3 %llu  v4 %llu  v5 %llu  v6 %llu\n", h[4], h[5], h[6], h[7], h[8], h[9]);
    printf("floor-visibility mismatches %llu, floor != trunc where visible %llu\n", h[10], h[11]);
    printf("u16/1000.0f: corrected mismatches %llu, plain product mismatches %llu\n", h[12], h[13]);
    return 0;
}
