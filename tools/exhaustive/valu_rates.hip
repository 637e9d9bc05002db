/*
 * valu_rates.hip -- issue cost (shader cycles per wave-instruction) of the vector instructions the fused alignment
 * kernel is built from, measured with s_memtime around long runs of independent instructions, at 1 and 2 waves per SIMD.
 * Decides whether packed-f32 (v_pk_*_f32) formulations of the per-point math pay (round 2).
 *
 *   hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

/* each BODY is one instruction on registers v[8..31] / s[20..27]; written so that consecutive copies are independent
 * enough (dest rotates via separate asm statements is not possible in a macro, so every test uses 4 variants) */
#define KERNEL(name, I0, I1, I2, I3)                                                                 \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, int iters, float seed) {   \
        asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n" \
                     "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n" \
                     "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n" \
                     "v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n" \
                     "v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n" \
                     "s_mov_b32 s20, 0x3f800000\n s_mov_b32 s21, 0x3f800000\n s_mov_b32 s22, 0x3f800000\n s_mov_b32 s23, 0x3f800000\n" \
                     :: "v"(seed) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
                        "v20","v21","v22","v23","v24","v25","v26","v27","s20","s21","s22","s23");  \
        unsigned long long t0, t1;                                                                   \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                \
        for (int i = 0; i < iters; i++) {                                                            \
            asm volatile(REP8(I0 "\n" I1 "\n" I2 "\n" I3 "\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15", \
                         "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","vcc","s24","s25","s26","s27", "memory"); \
        }                                                                                            \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0; \
    }

KERNEL(k_mul_f32, "v_mul_f32 v8, v16, v17", "v_mul_f32 v9, v18, v19", "v_mul_f32 v10, v20, v21", "v_mul_f32 v11, v22, v23")
KERNEL(k_mul_f32_s, "v_mul_f32 v8, s20, v17", "v_mul_f32 v9, s21, v19", "v_mul_f32 v10, s22, v21", "v_mul_f32 v11, s23, v23")
KERNEL(k_fma_f32, "v_fma_f32 v8, v16, v17, v18", "v_fma_f32 v9, v18, v19, v20", "v_fma_f32 v10, v20, v21, v22", "v_fma_f32 v11, v22, v23, v24")
KERNEL(k_pk_mul, "v_pk_mul_f32 v[8:9], v[16:17], v[18:19]", "v_pk_mul_f32 v[10:11], v[18:19], v[20:21]", "v_pk_mul_f32 v[12:13], v[20:21], v[22:23]", "v_pk_mul_f32 v[14:15], v[22:23], v[24:25]")
KERNEL(k_pk_mul_s, "v_pk_mul_f32 v[8:9], s[20:21], v[18:19] op_sel_hi:[0,1]", "v_pk_mul_f32 v[10:11], s[22:23], v[20:21] op_sel_hi:[0,1]", "v_pk_mul_f32 v[12:13], s[20:21], v[22:23] op_sel_hi:[0,1]", "v_pk_mul_f32 v[14:15], s[22:23], v[24:25] op_sel_hi:[0,1]")
KERNEL(k_pk_add, "v_pk_add_f32 v[8:9], v[16:17], v[18:19]", "v_pk_add_f32 v[10:11], v[18:19], v[20:21]", "v_pk_add_f32 v[12:13], v[20:21], v[22:23]", "v_pk_add_f32 v[14:15], v[22:23], v[24:25]")
KERNEL(k_pk_fma, "v_pk_fma_f32 v[8:9], v[16:17], v[18:19], v[20:21]", "v_pk_fma_f32 v[10:11], v[18:19], v[20:21], v[22:23]", "v_pk_fma_f32 v[12:13], v[20:21], v[22:23], v[24:25]", "v_pk_fma_f32 v[14:15], v[22:23], v[24:25], v[26:27]")
KERNEL(k_pk_mul_dep, "v_pk_mul_f32 v[8:9], v[8:9], v[18:19]", "v_pk_mul_f32 v[10:11], v[10:11], v[20:21]", "v_pk_mul_f32 v[12:13], v[12:13], v[22:23]", "v_pk_mul_f32 v[14:15], v[14:15], v[24:25]")
KERNEL(k_mul_dep, "v_mul_f32 v8, v8, v18", "v_mul_f32 v9, v9, v20", "v_mul_f32 v10, v10, v22", "v_mul_f32 v11, v11, v24")
KERNEL(k_fma_f64, "v_fma_f64 v[8:9], v[16:17], v[18:19], v[20:21]", "v_fma_f64 v[10:11], v[18:19], v[20:21], v[22:23]", "v_fma_f64 v[12:13], v[20:21], v[22:23], v[24:25]", "v_fma_f64 v[14:15], v[22:23], v[24:25], v[26:27]")
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 v[8:9], v16", "v_cvt_f64_f32 v[10:11], v17", "v_cvt_f64_f32 v[12:13], v18", "v_cvt_f64_f32 v[14:15], v19")
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 v8, v[16:17]", "v_cvt_f32_f64 v9, v[18:19]", "v_cvt_f32_f64 v10, v[20:21]", "v_cvt_f32_f64 v11, v[22:23]")
KERNEL(k_rcp_f32, "v_rcp_f32 v8, v16", "v_rcp_f32 v9, v17", "v_rcp_f32 v10, v18", "v_rcp_f32 v11, v19")
KERNEL(k_rcp_f64, "v_rcp_f64 v[8:9], v[16:17]", "v_rcp_f64 v[10:11], v[18:19]", "v_rcp_f64 v[12:13], v[20:21]", "v_rcp_f64 v[14:15], v[22:23]")
KERNEL(k_cndmask, "v_cndmask_b32 v8, v16, v17, vcc", "v_cndmask_b32 v9, v18, v19, vcc", "v_cndmask_b32 v10, v20, v21, vcc", "v_cndmask_b32 v11, v22, v23, vcc")
KERNEL(k_cvt_flr, "v_cvt_flr_i32_f32 v8, v16", "v_cvt_flr_i32_f32 v9, v17", "v_cvt_flr_i32_f32 v10, v18", "v_cvt_flr_i32_f32 v11, v19")
KERNEL(k_cvt_i32, "v_cvt_i32_f32 v8, v16", "v_cvt_i32_f32 v9, v17", "v_cvt_i32_f32 v10, v18", "v_cvt_i32_f32 v11, v19")
KERNEL(k_mad_u24, "v_mad_u32_u24 v8, v16, v17, v18", "v_mad_u32_u24 v9, v18, v19, v20", "v_mad_u32_u24 v10, v20, v21, v22", "v_mad_u32_u24 v11, v22, v23, v24")
KERNEL(k_mul_lo, "v_mul_lo_u32 v8, v16, v17", "v_mul_lo_u32 v9, v18, v19", "v_mul_lo_u32 v10, v20, v21", "v_mul_lo_u32 v11, v22, v23")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 v[8:9], v[16:17], 4, v[18:19]", "v_lshl_add_u64 v[10:11], v[18:19], 4, v[20:21]", "v_lshl_add_u64 v[12:13], v[20:21], 4, v[22:23]", "v_lshl_add_u64 v[14:15], v[22:23], 4, v[24:25]")
KERNEL(k_cmp_e64, "v_cmp_lt_f32 s[24:25], v16, v17", "v_cmp_lt_f32 s[26:27], v18, v19", "v_cmp_lt_f32 s[24:25], v20, v21", "v_cmp_lt_f32 s[26:27], v22, v23")
KERNEL(k_cvt_sdwa, "v_cvt_f32_u32_sdwa v8, v16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_cvt_f32_u32_sdwa v9, v17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0", "v_cvt_f32_u32_sdwa v10, v18 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_cvt_f32_u32_sdwa v11, v19 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0")
KERNEL(k_and_or, "v_and_or_b32 v8, v16, v17, v18", "v_and_or_b32 v9, v18, v19, v20", "v_and_or_b32 v10, v20, v21, v22", "v_and_or_b32 v11, v22, v23, v24")
KERNEL(k_mix, "v_pk_mul_f32 v[8:9], v[16:17], v[18:19]", "v_cndmask_b32 v10, v20, v21, vcc", "v_pk_add_f32 v[12:13], v[20:21], v[22:23]", "v_fma_f64 v[14:15], v[22:23], v[24:25], v[26:27]")

struct T { const char *name; void (*k)(unsigned long long *, int, float); };

int main() {
    T tests[] = {
        {"v_mul_f32 (vv)", k_mul_f32}, {"v_mul_f32 (sv)", k_mul_f32_s}, {"v_fma_f32", k_fma_f32},
        {"v_pk_mul_f32 (vv)", k_pk_mul}, {"v_pk_mul_f32 (s-splat,v)", k_pk_mul_s}, {"v_pk_add_f32", k_pk_add},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32 dependent x4", k_pk_mul_dep}, {"v_mul_f32 dependent x4", k_mul_dep},
        {"v_fma_f64", k_fma_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_rcp_f32", k_rcp_f32}, {"v_rcp_f64", k_rcp_f64}, {"v_cndmask_b32", k_cndmask},
        {"v_cvt_flr_i32_f32", k_cvt_flr}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_mad_u32_u24", k_mad_u24},
        {"v_mul_lo_u32", k_mul_lo}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cmp_lt_f32 e64", k_cmp_e64},
        {"v_cvt_f32_u32 sdwa", k_cvt_sdwa}, {"v_and_or_b32", k_and_or}, {"mix pk_mul/cndmask/pk_add/fma_f64", k_mix},
    };
    unsigned long long *d;
    hipMalloc(&d, 4096 * sizeof(unsigned long long));
    const int iters = 2000;
    printf("%-36s %10s %10s %10s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    for (auto &t : tests) {
        printf("%-36s", t.name);
        for (int threads : {256, 512, 1024}) {
            /* one workgroup per CU: 256 threads = 1 wave per SIMD, 512 = 2, 1024 = 4 */
            hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * threads / 64);
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double cyc = (double)h[h.size() / 2] / (iters * 32.0);
            /* per-wave cycles per instruction; SIMD issue cost = that / waves per SIMD */
            printf(" %6.2f/%-4.2f", cyc, cyc / (threads / 256));
        }
        printf("\n");
    }
    printf("(columns: cycles per instruction seen by one wave / SIMD cycles per wave-instruction)\n");
    return 0;
}
