"""Does an EXTERNAL event recorded in the middle of a captured hipGraph release a waiter on another stream at that point of
the replay (not before, not only at the end)?  Basis for all-reducing gradient slices beside the rest of a one-graph step."""
import torch

dev = torch.device("cuda:0")
x = torch.ones(1 << 24, device=dev)
y = torch.zeros(1, device=dev)
val = torch.zeros(1, device=dev)
snap = torch.zeros(1, device=dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
E = ctypes.c_void_p()
assert hip.hipEventCreateWithFlags(ctypes.byref(E), 2) == 0            # hipEventDisableTiming


def record_external(stream):            # torch refuses external events on ROCm: go to the runtime directly
    rc = hip.hipEventRecordWithFlags(E, ctypes.c_void_p(stream.cuda_stream), 1)          # hipEventRecordExternal
    assert rc == 0, "hipEventRecordWithFlags rc=%d" % rc


def wait_external(stream):
    rc = hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), E, 0)
    assert rc == 0, "hipStreamWaitEvent rc=%d" % rc


side = torch.cuda.Stream()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.mul_(1.0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        x.mul_(1.0)
    y.copy_(val)
    record_external(torch.cuda.current_stream())
    for _ in range(300):
        x.mul_(1.0)
ok = True
for it in range(6):
    val.fill_(float(it + 1))
    t0, t_snap, t_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record()
    g.replay()
    with torch.cuda.stream(side):
        wait_external(side)
        snap.copy_(y)
        t_snap.record()
    t_end.record()
    torch.cuda.synchronize()
    a, b = t0.elapsed_time(t_snap), t0.elapsed_time(t_end)
    print("it %d: snapshot %.0f (want %d)  snapshot at %.3f ms, graph end at %.3f ms" % (it, float(snap), it + 1, a, b))
    ok = ok and float(snap) == it + 1 and a < 0.5 * b
print("EXTERNAL_EVENT_OK" if ok else "EXTERNAL_EVENT_BROKEN")
