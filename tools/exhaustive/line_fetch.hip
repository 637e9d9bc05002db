// What does an L2 miss fetch on gfx950: the 64-byte half of the 128-byte line that was asked for, or the whole line?
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate in your own access pattern".)
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum (tools/experiments/r04_line_fetch.sh).
//   A  halves_two_pass : one workgroup per XCD-sized region (1 MiB each, 8 workgroups).  Pass 1 reads 4 bytes of the FIRST half of
//      every line (plain loads); pass 2 reads 4 bytes of the SECOND half with L1-bypassing loads.  RDREQ == lines: a miss fetches the
//      whole line.  RDREQ == 2 x lines: it fetches the half.
//   B  gather_pairs    : big buffer, every line touched once; lanes 2k / 2k+1 of an instruction read the two halves of ONE random line.
//   C  gather_single   : the same lines, but an instruction touches only one half per line; the other halves by a later instruction.
//   D  gather12        : 12-byte loads (the kernel's dwordx3 gather), one per 64-byte half, both halves in one instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ unsigned load_sc1(const unsigned *p) {
    unsigned r;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__global__ void halves_two_pass(const unsigned *buf, size_t region_words, unsigned *out, int second_pass) {
    const unsigned *p = buf + (size_t)blockIdx.x * region_words;
    const size_t lines = region_words / 32;
    unsigned s = 0;
    for (size_t i = threadIdx.x; i < lines; i += blockDim.x) s += p[i * 32];               // first half of every line
    __syncthreads();
    if (second_pass)
        for (size_t i = threadIdx.x; i < lines; i += blockDim.x) s += load_sc1(p + i * 32 + 16);   // second half, L1 bypassed
    if (s == 0x12345u) *out = s;
}
__global__ void gather_pairs(const unsigned *buf, size_t lines, unsigned *out, int mode) {
    // mode 0: lane pair (2k, 2k+1) -> halves 0 / 1 of line perm(i/2);  mode 1: all lanes half 0 of line perm(i), then (second loop) half 1
    unsigned s = 0;
    const size_t n = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) {
        for (size_t i = t; i < 2 * lines; i += n) {
            const size_t j = ((i >> 1) * 2654435761ull) % lines;
            s += buf[j * 32 + (i & 1) * 16];
        }
    } else {
        for (int h = 0; h < 2; h++)
            for (size_t i = t; i < lines; i += n) {
                const size_t j = (i * 2654435761ull) % lines;
                s += buf[j * 32 + h * 16];
            }
    }
    if (s == 0x12345u) *out = s;
}
struct __attribute__((packed, aligned(4))) U3 { unsigned a, b, c; };
__global__ void gather12(const unsigned *buf, size_t lines, unsigned *out) {
    unsigned s = 0;
    const size_t n = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t; i < 2 * lines; i += n) {
        const size_t j = ((i >> 1) * 2654435761ull) % lines;
        const U3 v = *reinterpret_cast<const U3 *>(buf + j * 32 + (i & 1) * 16 + ((i >> 1) & 7) + 1);
        s += v.a + v.b + v.c;
    }
    if (s == 0x12345u) *out = s;
}
int main() {
    const size_t bytes = 2ull << 30;
    unsigned *p, *o; hipMalloc(&p, bytes); hipMalloc(&o, 4); hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    // A: 8 workgroups x 1 MiB; regions 16 MiB apart so that nothing is shared; first without, then with the second pass
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(halves_two_pass, dim3(8), dim3(1024), 0, 0, p + rep * (64u << 20), (size_t)(1u << 20) / 4, o, 0);
        hipLaunchKernelGGL(halves_two_pass, dim3(8), dim3(1024), 0, 0, p + (32u << 20) + rep * (64u << 20), (size_t)(1u << 20) / 4, o, 1);
    }
    hipDeviceSynchronize();
    const size_t lines = (1ull << 30) / 128;         // 1 GiB of lines, power of two
    hipLaunchKernelGGL(gather_pairs, dim3(2048), dim3(256), 0, 0, p, lines, o, 0);
    hipLaunchKernelGGL(gather_pairs, dim3(2048), dim3(256), 0, 0, p + (1ull << 28), lines, o, 1);
    hipLaunchKernelGGL(gather12, dim3(2048), dim3(256), 0, 0, p, lines, o);
    hipDeviceSynchronize();
    printf("lines per region (A): %zu x 8 regions; lines (B, C, D): %zu\n", (size_t)(1u << 20) / 128, lines);
    return 0;
}
