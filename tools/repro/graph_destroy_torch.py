"""Torch-only reproducer (no libhmvae_hip.so) for the host-side crash after a hipGraph captured across side streams is
destroyed: the engine's stream / event pattern with torch.cuda.graph, torch.cuda.Stream and persistent torch.cuda.Events.

    MALLOC_CHECK_=3 python tools/repro/graph_destroy_torch.py <joinall> <rotate> <precreate> <keep> [iters]
joinall   1: the join records + waits events on all 16 side streams, forked in the step or not
rotate    1: the fork -> side-stream assignment keeps rotating across steps; 0: the same streams every step
precreate 1: every event is recorded once eagerly before the first capture (torch creates the hipEvent on first record)
keep      1: captured graphs are kept alive (parked) instead of destroyed
"""
import sys

import torch

joinall, rotate, precreate, keep = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 0), (2, 1), (3, 0), (4, 0)))
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
import ctypes
_v = ctypes.c_int(0)
ctypes.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(ctypes.byref(_v))           # the copy already mapped into this process
print("HIP runtime version", _v.value, "torch.version.hip", torch.version.hip, flush=True)
with open("/proc/self/maps") as _f:
    print(sorted(set(ln.split()[-1] for ln in _f if "libamdhip64" in ln)), flush=True)
NS, NFORK = 16, 5
side = [torch.cuda.Stream(device=dev) for _ in range(NS)]
events = {}
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(NS + 1)]
turn = 0
parked = []


def ev(key):
    if key not in events:
        events[key] = torch.cuda.Event()
    return events[key]


def wait(waiter, waited, key):
    e = ev(key)
    e.record(waited)
    waiter.wait_event(e)


def step():
    global turn
    main = torch.cuda.current_stream()
    if not rotate:
        turn = 0
    bufs[NS].add_(1.0)
    forked = []
    for _ in range(NFORK):
        turn = (turn + 1) % NS
        wait(side[turn], main, ("fork", turn))
        forked.append(turn)
        with torch.cuda.stream(side[turn]):
            bufs[turn].add_(1.0)
            bufs[turn].mul_(0.5)
        bufs[NS].add_(1.0)
    for i in (range(NS) if joinall else forked):
        wait(main, side[i], ("join", i))
    bufs[NS].add_(1.0)


if precreate:
    for i in range(NS):
        ev(("fork", i)).record()
        ev(("join", i)).record()
    torch.cuda.synchronize()
for it in range(iters):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    if keep:
        parked.append(g)
    del g
    if it % 10 == 0:
        print("iter", it, "ok", flush=True)
print("joinall %d rotate %d precreate %d keep %d: %d rounds clean, counter %.0f" % (joinall, rotate, precreate, keep, iters,
                                                                                  float(bufs[NS][0])))
