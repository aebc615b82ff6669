// Pure-HIP reproducer for the host-side crash seen when a hipGraph captured across side streams is destroyed and eager
// multi-stream work follows ("graph lifetime").  No torch, no libhmvae.
//   hipcc --offload-arch=gfx950 -O2 -o graph_destroy_repro graph_destroy_repro.hip
//   MALLOC_CHECK_=3 ./graph_destroy_repro <joinall> <rotate> <lazy> <tempwait> <torchlike> [iters]
// joinall  1: the join records + waits events on all 16 side streams, forked in this step or not
// rotate   1: the fork -> side-stream assignment keeps rotating across steps (so a capture forks other streams than
//             the eager steps around it); 0: the same streams every step
// lazy     1: events are created on first use (possibly inside a capture), like torch.cuda.Event
// tempwait 1: the warm-up is bracketed by torch's Stream.wait_stream pattern (temporary event: record, wait, destroy)
// torchlike 1: the graph is instantiated with hipGraphInstantiateFlagAutoFreeOnLaunch, the hipGraph destroyed right after
//             instantiation (only the exec lives on) and the exec launched on the NULL stream, as at::cuda::CUDAGraph does
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void bump(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
constexpr int NS = 16, NFORK = 5, N = 1 << 16;
static hipStream_t side[NS];
static hipEvent_t fork_ev[NS], join_ev[NS];
static float* buf[NS + 1];
static int turn = 0, joinall, rotate_, lazy_;
static hipEvent_t get(hipEvent_t* e) { if (!*e) CK(hipEventCreateWithFlags(e, hipEventDisableTiming)); return *e; }
static void wait_stream(hipStream_t waiter, hipStream_t waited) {          // torch's Stream.wait_stream
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipEventRecord(e, waited)); CK(hipStreamWaitEvent(waiter, e, 0)); CK(hipEventDestroy(e));
}
static void step(hipStream_t s0) {
    bool forked[NS] = {};
    if (!rotate_) turn = 0;
    bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
    for (int k = 0; k < NFORK; ++k) {
        turn = (turn + 1) % NS;
        CK(hipEventRecord(get(&fork_ev[turn]), s0));
        CK(hipStreamWaitEvent(side[turn], fork_ev[turn], 0));
        forked[turn] = true;
        bump<<<N / 256, 256, 0, side[turn]>>>(buf[turn], N);
        bump<<<N / 256, 256, 0, side[turn]>>>(buf[turn], N);
        bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
    }
    for (int i = 0; i < NS; ++i) {
        if (!joinall && !forked[i]) continue;
        CK(hipEventRecord(get(&join_ev[i]), side[i]));
        CK(hipStreamWaitEvent(s0, join_ev[i], 0));
    }
    bump<<<N / 256, 256, 0, s0>>>(buf[NS], N);
}
int main(int argc, char** argv) {
    joinall = argc > 1 ? atoi(argv[1]) : 0; rotate_ = argc > 2 ? atoi(argv[2]) : 1; lazy_ = argc > 3 ? atoi(argv[3]) : 1;
    const int tempwait = argc > 4 ? atoi(argv[4]) : 1, torchlike = argc > 5 ? atoi(argv[5]) : 1, iters = argc > 6 ? atoi(argv[6]) : 200;
    int rtv = 0; CK(hipRuntimeGetVersion(&rtv)); printf("HIP runtime version %d\n", rtv);
    hipStream_t cur = nullptr, warm, cap;
    if (!torchlike) CK(hipStreamCreateWithFlags(&cur, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&warm, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    for (int i = 0; i < NS; ++i) {
        CK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
        if (!lazy_) { get(&fork_ev[i]); get(&join_ev[i]); }
    }
    for (int i = 0; i <= NS; ++i) { CK(hipMalloc(&buf[i], N * sizeof(float))); CK(hipMemset(buf[i], 0, N * sizeof(float))); }
    for (int it = 0; it < iters; ++it) {
        if (tempwait) wait_stream(warm, cur);
        for (int r = 0; r < 3; ++r) step(warm);            // eager multi-stream warm-up
        if (tempwait) wait_stream(cur, warm);
        CK(hipDeviceSynchronize());
        hipGraph_t g; hipGraphExec_t ge;
        if (tempwait) wait_stream(cap, cur);
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal));
        step(cap);
        CK(hipStreamEndCapture(cap, &g));
        if (tempwait) wait_stream(cur, cap);
        if (torchlike) {
            CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
            CK(hipGraphDestroy(g));
        } else {
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, cur));
        CK(hipDeviceSynchronize());
        CK(hipGraphExecDestroy(ge));
        if (!torchlike) CK(hipGraphDestroy(g));
        if (it % 20 == 0) { printf("iter %d ok\n", it); fflush(stdout); }
    }
    float h = 0; CK(hipMemcpy(&h, buf[NS], 4, hipMemcpyDeviceToHost));
    printf("joinall %d rotate %d lazy %d tempwait %d torchlike %d: %d capture/destroy rounds clean, counter %.0f\n", joinall,
           rotate_, lazy_, tempwait, torchlike, iters, h);
    return 0;
}
