// round 6: does the Infinity Cache serve single-line gathers faster than HBM does?
// Random ONE-line gathers with the packed kernel's own load (12 bytes: up / centre / down of a pixel's column in a 128-byte
// line of the compact now form) over footprints of 32 MB ... 1 GB, swept repeatedly so that footprints below 256 MiB are
// cache-resident after the first sweep; eight independent loads in flight per lane, 8 x 256 threads per CU like the kernel's
// two workgroups of 256.  Output: G lines/s per footprint (every load touches another 128-byte line).
//   hipcc -O3 --offload-arch=gfx950 -o ic_gather ic_gather.hip ; ./ic_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct u3 { unsigned x, y, z; };
template <int K>
__global__ void __launch_bounds__(256) gather12(const unsigned char *__restrict__ base, unsigned n_lines, int iters, unsigned seed, unsigned *out) {
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        u3 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            s = s * 1664525u + 1013904223u;
            const unsigned line = (unsigned)(((unsigned long long)(s >> 4) * n_lines) >> 28);      // uniform in [0, n_lines)
            const unsigned off = ((s >> 1) & 7u) * 12u;                                            // a 12-byte piece inside the line
            v[k] = *reinterpret_cast<const u3 *>(base + (size_t)line * 128u + off);
        }
#pragma unroll
        for (int k = 0; k < K; k++) acc += v[k].x ^ v[k].y ^ v[k].z;
    }
    if (acc == 0x12345678u) *out = acc;
}
template <int K>
static double rate(const unsigned char *p, unsigned *o, size_t mb, int wgs, hipEvent_t e0, hipEvent_t e1) {
    const unsigned n_lines = (unsigned)((mb << 20) / 128);
    const int iters = 512 / K;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL(gather12<K>, dim3(wgs), dim3(256), 0, 0, p, n_lines, iters, 17u * w, o);      // warm: the footprint into the caches
    (void)hipDeviceSynchronize();
    const int reps = 10;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(gather12<K>, dim3(wgs), dim3(256), 0, 0, p, n_lines, iters, 1000u + r, o);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * 256 * iters * K * reps / (ms * 1e-3) / 1e9;
}
int main() {
    const size_t max_bytes = 1ull << 30;
    unsigned char *p; unsigned *o;
    if (hipMalloc(&p, max_bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(p, 1, max_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("G lines/s of random single-line 12-byte gathers (every load another 128-byte line); rows: footprint, columns: loads in flight per lane\n");
    for (int wpc : {2, 4, 8}) {
        printf("-- %d workgroups of 256 threads per CU (%d waves per CU)\n footprint_MB      1      2      4      8\n", wpc, 4 * wpc);
        for (size_t mb : {32, 64, 128, 192, 288, 1024}) {
            const int wgs = 256 * wpc;
            printf("%8zu      %6.1f %6.1f %6.1f %6.1f\n", mb, rate<1>(p, o, mb, wgs, e0, e1), rate<2>(p, o, mb, wgs, e0, e1), rate<4>(p, o, mb, wgs, e0, e1), rate<8>(p, o, mb, wgs, e0, e1));
        }
    }
    return 0;
}
