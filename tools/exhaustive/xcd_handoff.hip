/*
 * xcd_handoff.hip -- what a 16-byte tagged record costs between two workgroups, by store flavour and placement (gfx950).
 *
 * The team exchange of the fused kernel (rgbd_odometry_amd/csrc/dvo_fused.hip: team_exchange) publishes every member's sums as
 * {value, tag} records with `global_store_dwordx4 ... sc1` and polls them with `global_load_dwordx4 ... sc1`.  sc1 stores drop the
 * line from the XCD's L2 (MI355X_MICROARCH.md, inter-workgroup visibility), so even members that share an XCD -- which is how the
 * teams of up to 32 workgroups are placed -- read each other's records at the memory-side latency.  This tool measures the
 * alternative for members KNOWN to share an XCD (HW_REG_XCC_ID read at run time): plain stores (the line stays in that XCD's L2,
 * which all its CUs share) + sc1 loads (bypass the reader's L1, served by the L2).
 *
 * Ping-pong: workgroup A writes record i, B polls it and answers with record i, A polls that: N round trips, s_memtime around them.
 * Every hop carries value + 1, so a stale or torn read shows up as a wrong value or as a bounded-spin timeout.
 *   pairs (b, b + 8)  same XCD under round-robin dispatch; pairs (b, b + 1) different XCDs.
 *   load: the other workgroups of the launch stream a buffer from HBM meanwhile.
 *
 * build: hipcc -O3 --offload-arch=gfx950 -o xcd_handoff xcd_handoff.hip ; run: ./xcd_handoff
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_rec(v4u *p, v4u r, int flavour) {
    if (flavour == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory");
    else if (flavour == 1) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(r) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ v4u load_rec(const v4u *p) {
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

struct Result { unsigned long long ticks; unsigned xcc_a, xcc_b; unsigned errors, timeouts; unsigned long long last; };

/* blocks [0, 2*npairs): ping-pong pairs; the rest: background readers */
__global__ void __launch_bounds__(64) pingpong(v4u *recs, Result *res, int npairs, int n_pp /* ping-pong blocks */, int stride /* partner = b + stride */,
                                               int n_iter, int flavour, const float4 *bg, size_t bg_n, float *sink) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (b >= n_pp) {      /* background load: stream bg_n float4 per block */
        float acc = 0.f;
        const float4 *p = bg + (size_t)(b - n_pp) * bg_n;
        for (size_t i = lane; i < bg_n; i += 64) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
        if (acc == 12345.678f) sink[0] = acc;
        return;
    }
    /* pair index and role: with stride 8 blocks 0..7 are A of pairs 0..7 and 8..15 their partners; with stride 1 even blocks are A */
    int pair, role;
    if (stride == 8) { pair = b & 7; role = b >> 3; }
    else { pair = b >> 1; role = b & 1; }
    if (pair >= npairs) return;
    v4u *slotAB = recs + (size_t)pair * 16;          /* own 128-byte line each */
    v4u *slotBA = recs + (size_t)pair * 16 + 8;
    Result &R = res[pair];
    const unsigned my_xcc = xcc_id();
    if (lane == 0) { if (role == 0) R.xcc_a = my_xcc; else R.xcc_b = my_xcc; }
    if (lane != 0) return;
    unsigned errors = 0, timeouts = 0;
    unsigned long long value = 0;
    unsigned long long t0 = 0;
    for (int i = 0; i < n_iter + 16; i++) {
        if (i == 16 && role == 0) t0 = __builtin_amdgcn_s_memtime();
        const unsigned tag = (unsigned)i + 1u;
        if (role == 0) {
            value += 1;
            v4u r; r.x = (unsigned)value; r.y = tag; r.z = (unsigned)(value >> 32); r.w = tag;
            store_rec(slotAB, r, flavour);
            int spins = 0;
            v4u g;
            for (;;) {
                g = load_rec(slotBA);
                if (g.y == tag && g.w == tag) break;
                if (++spins > 2000000) { timeouts++; break; }
            }
            const unsigned long long got = ((unsigned long long)g.z << 32) | g.x;
            if (got != value + 1) errors++;
            value = got;
        } else {
            int spins = 0;
            v4u g;
            for (;;) {
                g = load_rec(slotAB);
                if (g.y == tag && g.w == tag) break;
                if (++spins > 2000000) { timeouts++; break; }
            }
            const unsigned long long got = ((unsigned long long)g.z << 32) | g.x;
            if (got != value + 1) errors++;
            value = got + 1;
            v4u r; r.x = (unsigned)value; r.y = tag; r.z = (unsigned)(value >> 32); r.w = tag;
            store_rec(slotBA, r, flavour);
        }
        if (timeouts > 3) break;
    }
    if (role == 0) { R.ticks = __builtin_amdgcn_s_memtime() - t0; R.last = value; }
    atomicAdd(&R.errors, errors);
    atomicAdd(&R.timeouts, timeouts);
}

int main() {
    const int n_iter = 2000;
    v4u *recs; Result *res; float4 *bg; float *sink;
    const int max_pairs = 16;
    hipMalloc(&recs, sizeof(v4u) * 16 * max_pairs);
    hipMalloc(&res, sizeof(Result) * max_pairs);
    const size_t bg_n = 1 << 18;            /* 4 MB per background block */
    const int bg_blocks = 240;
    hipMalloc(&bg, sizeof(float4) * bg_n * bg_blocks);
    hipMemset(bg, 0, sizeof(float4) * bg_n * bg_blocks);
    hipMalloc(&sink, 4);
    const char *fl[3] = {"sc1 store", "plain store", "sc0 store"};
    for (int load = 0; load < 2; load++)
        for (int stride : {8, 1})
            for (int flavour = 0; flavour < 3; flavour++)
                for (int npairs : {1, 8}) {
                    hipMemset(recs, 0, sizeof(v4u) * 16 * max_pairs);
                    hipMemset(res, 0, sizeof(Result) * max_pairs);
                    /* with stride 8 the pairs need blocks b and b + 8: launch 16 ping-pong blocks, pairs >= npairs return at once */
                    const int pp_blocks = (stride == 8) ? 16 : 2 * npairs;
                    hipLaunchKernelGGL(pingpong, dim3(pp_blocks + (load ? bg_blocks : 0)), dim3(64), 0, 0, recs, res, npairs, pp_blocks, stride, n_iter,
                                       flavour, bg, bg_n, sink);
                    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                    std::vector<Result> h(max_pairs);
                    hipMemcpy(h.data(), res, sizeof(Result) * max_pairs, hipMemcpyDeviceToHost);
                    double sum = 0; unsigned err = 0, to = 0; int same = 0;
                    for (int p = 0; p < npairs; p++) { sum += (double)h[p].ticks / n_iter; err += h[p].errors; to += h[p].timeouts; same += h[p].xcc_a == h[p].xcc_b; }
                    printf("%-6s partner b+%d  %-12s pairs %d: %8.0f ticks per round trip (2 hops)  same-XCD pairs %d/%d  errors %u timeouts %u  xcc %u/%u\n",
                           load ? "loaded" : "idle", stride, fl[flavour], npairs, sum / npairs, same, npairs, err, to, h[0].xcc_a, h[0].xcc_b);
                }
    return 0;
}
