"""Torch-only reproducer (no libhmvae_hip.so) for the host-heap abort after a hipGraph captured across side streams is
destroyed: the engine's stream / event pattern with torch.cuda.graph, torch.cuda.Stream and persistent torch.cuda.Events.

    MALLOC_CHECK_=3 python tools/repro/graph_destroy_torch.py <mode> [iters]
mode 0: the join waits only for side streams forked in the step; mode 1: the join records + waits events on all 16 side
streams, forked or not (round-1 engine behaviour).
"""
import sys

import torch

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
NS, NFORK = 16, 5
side = [torch.cuda.Stream(device=dev) for _ in range(NS)]
events = {}
bufs = [torch.zeros(1 << 16, device=dev) for _ in range(NS + 1)]
turn = 0


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
    bufs[NS].add_(1.0)
    forked = []
    for _ in range(NFORK):
        turn = (turn + 1) % NS                      # like the trajectory engine of round 1: keeps rotating across steps
        wait(side[turn], main, ("fork", turn))
        forked.append(turn)
        with torch.cuda.stream(side[turn]):
            bufs[turn].add_(1.0)
            bufs[turn].mul_(0.5)
        bufs[NS].add_(1.0)
    for i in (range(NS) if mode == 1 else forked):
        wait(main, side[i], ("join", i))
    bufs[NS].add_(1.0)


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
    del g
    if it % 10 == 0:
        print("iter", it, "ok", flush=True)
print("mode %d: %d capture/destroy rounds clean, counter %.0f" % (mode, iters, float(bufs[NS][0])))
