/*
 * loopback_rccl.hip -- TEST INFRASTRUCTURE ONLY (tests/test_gpu_tiled_ranks.py).
 *
 * An in-process stand-in for the one RCCL call the tiled path makes (ncclAllReduce of <= 64 doubles, ncclSum), between HOST THREADS
 * that each drive one context on the SAME GPU.  The test boxes have one GPU and RCCL refuses two ranks on one device, so this is the
 * only way to run rank > 0 of dvo_align_pyramid_tiled -- its shard ranges, the update from sums that are not this rank's own -- on the
 * hardware.  dvo_tiled_attach(ctx, comm, rank, world, "<this library>") resolves ncclAllReduce from the library it is given
 * (rgbd_odometry_amd/csrc/dvo_capi_tiled.cpp), exactly as it does for librccl.so.1.
 *
 * One call: copy this rank's buffer into its slot (on the caller's stream), record an event, meet the other ranks at a host barrier,
 * make the stream wait for their events, add the slots in rank order -- the same bits on every rank, like a ring all-reduce gives.
 * Slots and events are double-buffered by the parity of the call number; a capturing stream is refused (the events of the other
 * ranks cannot be captured), which sends the product to its direct-submission path.
 */
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace {
constexpr int kMaxCount = 64;
struct Shared {
    int world = 0;
    double *slots = nullptr;                 /* [2][world][kMaxCount] */
    std::vector<hipEvent_t> ev;              /* [2][world] */
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;
};
struct Comm { Shared *sh; int rank; unsigned long long seq; long calls; };

__global__ void loopback_sum_kernel(const double *__restrict__ slots, int world, int count, double *__restrict__ out) {
    const int i = threadIdx.x;
    if (i >= count) return;
    double s = slots[i];
    for (int r = 1; r < world; r++) s += slots[r * kMaxCount + i];
    out[i] = s;
}

bool meet(Shared *sh) {                      /* host barrier of the `world` calling threads; false after 30 s */
    std::unique_lock<std::mutex> lock(sh->m);
    if (sh->broken) return false;
    const unsigned long long gen = sh->generation;
    if (++sh->arrived == sh->world) { sh->arrived = 0; sh->generation++; sh->cv.notify_all(); return true; }
    if (!sh->cv.wait_for(lock, std::chrono::seconds(30), [&] { return sh->generation != gen || sh->broken; })) { sh->broken = true; sh->cv.notify_all(); return false; }
    return !sh->broken;
}
}  // namespace

extern "C" {

/* comms_out[0..world): one handle per rank (what the test passes to dvo_tiled_attach as the communicator) */
int loopback_create(int world, void **comms_out) {
    if (world < 1 || world > 16) return 1;
    Shared *sh = new Shared;
    sh->world = world;
    if (hipMalloc(&sh->slots, sizeof(double) * 2 * world * kMaxCount) != hipSuccess) { delete sh; return 2; }
    sh->ev.resize(2 * world);
    for (auto &e : sh->ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 2;
    for (int r = 0; r < world; r++) comms_out[r] = new Comm{sh, r, 0ull, 0l};
    return 0;
}
void loopback_destroy(void **comms, int world) {
    if (!comms || !comms[0]) return;
    Shared *sh = static_cast<Comm *>(comms[0])->sh;
    (void)hipDeviceSynchronize();
    for (auto &e : sh->ev) (void)hipEventDestroy(e);
    (void)hipFree(sh->slots);
    for (int r = 0; r < world; r++) delete static_cast<Comm *>(comms[r]);
    delete sh;
}
long loopback_calls(void *comm) { return static_cast<Comm *>(comm)->calls; }

int ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, hipStream_t stream) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c || count > (size_t)kMaxCount || datatype != 8 /* ncclDouble */ || op != 0 /* ncclSum */) return 4;   /* ncclInvalidArgument */
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 1; }
    Shared *sh = c->sh;
    const int par = (int)(c->seq & 1ull);
    double *slot0 = sh->slots + (size_t)par * sh->world * kMaxCount;
    if (hipMemcpyAsync(slot0 + (size_t)c->rank * kMaxCount, sendbuff, sizeof(double) * count, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    if (hipEventRecord(sh->ev[(size_t)par * sh->world + c->rank], stream) != hipSuccess) return 1;
    if (!meet(sh)) return 6;                  /* ncclRemoteError: a rank did not arrive */
    for (int r = 0; r < sh->world; r++)
        if (r != c->rank && hipStreamWaitEvent(stream, sh->ev[(size_t)par * sh->world + r], 0) != hipSuccess) return 1;
    hipLaunchKernelGGL(loopback_sum_kernel, dim3(1), dim3(kMaxCount), 0, stream, slot0, sh->world, (int)count, static_cast<double *>(recvbuff));
    if (hipGetLastError() != hipSuccess) return 1;
    c->seq++; c->calls++;
    return 0;
}
const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 4 ? "invalid argument" : rc == 6 ? "a rank did not arrive at the loopback all-reduce" : "HIP error / capturing stream"; }

}  // extern "C"
