// Calibration of rocprofv3 FETCH_SIZE / TCC_EA0_RDREQ for the access patterns of the alignment kernel
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
//   kernels read a 2 GiB buffer (>> 256 MiB Infinity Cache, every byte touched at most once):
//   stream16      : every lane reads consecutive float4                      -> 2 GiB useful
//   gather16_s64  : one float4 per 64-byte sector  (lanes scattered)         -> 512 MiB useful, 2 GiB of sectors
//   gather16_s128 : one float4 per 128-byte line                             -> 256 MiB useful
//   gather16_rand : one float4 at a pseudo-random 16-byte slot, 1/8 of slots -> 256 MiB useful
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void stream16(const float4 *p, size_t n, float *out) {
    float s = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.f) *out = s;
}
__global__ void gather16_stride(const float4 *p, size_t n_slots, int stride_slots, float *out) {
    // lane l of wave w reads slot perm(i)*stride : scatter lanes so that a wave instruction touches 64 different lines
    float s = 0; const size_t n = n_slots / stride_slots;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = (i * 2654435761ull) % n;      // bijection for odd multiplier when n is a power of two
        float4 v = p[j * stride_slots]; s += v.x + v.w;
    }
    if (s == 12345.f) *out = s;
}
int main() {
    const size_t bytes = 2ull << 30, n = bytes / 16;
    float4 *p; float *o; hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
    hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, p, n, o);
    hipLaunchKernelGGL(gather16_stride, dim3(2048), dim3(256), 0, 0, p, n, 4, o);   // one per 64 B
    hipLaunchKernelGGL(gather16_stride, dim3(2048), dim3(256), 0, 0, p, n, 8, o);   // one per 128 B
    hipLaunchKernelGGL(gather16_stride, dim3(2048), dim3(256), 0, 0, p, n, 1, o);   // every slot, scattered order: 2 GiB useful
    hipDeviceSynchronize();
    // Infinity-Cache-resident variants: a 96 MiB window (>> 32 MiB of L2, < 256 MiB MALL), swept 4 times back to back
    const size_t nw = (96ull << 20) / 16;
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(gather16_stride, dim3(2048), dim3(256), 0, 0, p, nw, 4, o);   // 64-B sectors
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, p, nw, o);
    hipDeviceSynchronize();
    printf("done\n"); return 0;
}
