// Pure-HIP reproducer for the host-heap abort seen when a hipGraph captured across side streams is destroyed and eager
// multi-stream work follows ("graph lifetime").  No torch, no libhmvae.
//   hipcc --offload-arch=gfx950 -O2 -o graph_destroy_repro graph_destroy_repro.hip
//   MALLOC_CHECK_=3 ./graph_destroy_repro <mode> [iters]
// mode 0: clean fork/join (every joined side stream was forked inside the capture)
// mode 1: the engine's round-1 pattern -- the join also records + waits events on side streams that were NEVER forked
//         in this capture (an eager record on a non-capturing stream, waited on by the capturing stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void bump(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
constexpr int NS = 16, NFORK = 12, N = 1 << 16;
static hipStream_t s0, side[NS];
static hipEvent_t fork_ev[NS], join_ev[NS];
static float* buf[NS + 1];
static int step(int mode) {
    bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
    for (int i = 1; i <= NFORK; ++i) {
        CK(hipEventRecord(fork_ev[i], s0));
        CK(hipStreamWaitEvent(side[i], fork_ev[i], 0));
        bump<<<N / 256, 256, 0, side[i]>>>(buf[i], N);
        bump<<<N / 256, 256, 0, side[i]>>>(buf[i], N);
        bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
    }
    for (int i = 0; i < NS; ++i) {
        if (mode == 0 && (i == 0 || i > NFORK)) continue;
        CK(hipEventRecord(join_ev[i], side[i]));
        CK(hipStreamWaitEvent(s0, join_ev[i], 0));
    }
    bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
    return 0;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 200;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    for (int i = 0; i < NS; ++i) {
        CK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&fork_ev[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming));
    }
    for (int i = 0; i <= NS; ++i) { CK(hipMalloc(&buf[i], N * sizeof(float))); CK(hipMemset(buf[i], 0, N * sizeof(float))); }
    for (int it = 0; it < iters; ++it) {
        if (step(mode)) return 2;                          // eager multi-stream warm-up
        CK(hipDeviceSynchronize());
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        if (step(mode)) return 2;
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
        CK(hipDeviceSynchronize());
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        if (it % 20 == 0) { printf("iter %d ok\n", it); fflush(stdout); }
    }
    float h = 0; CK(hipMemcpy(&h, buf[NS], 4, hipMemcpyDeviceToHost));
    printf("mode %d: %d capture/destroy rounds clean, counter %.0f\n", mode, iters, h);
    return 0;
}
