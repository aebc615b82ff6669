"""Dev tool (GPU box): the one-launch latent bridge alone (HIP events), deep level of the len64 config at B = 256."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd import ops
dev = torch.device("cuda:0")
B, k, n_in, d, n_out = 256, 7, 384, 24, 384
bp = 256
x = torch.randn(k * n_in, bp, device=dev)
w1, b1 = torch.randn(2 * d, n_in, device=dev) / 20, torch.randn(2 * d, device=dev)
w2, b2 = torch.randn(n_out, d, device=dev) / 5, torch.randn(n_out, device=dev)
eps = torch.randn(B * k, d, device=dev)
dist, z, y = torch.empty(k * 2 * d, bp, device=dev), torch.empty(k * d, bp, device=dev), torch.empty(k * n_out, bp, device=dev)
part = torch.empty(k * (bp // 16), device=dev)
dy, dd, dx = torch.randn_like(y), torch.empty_like(dist), torch.empty_like(x)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(p.elapsed_time(q) for p, q in evs)[len(evs) // 2] * 1e3
f = lambda: ops.latent_bridge_fwd(x, w1, b1, eps, w2, b2, dist, z, y, part, k, n_in, d, n_out, n_in, n_out, 0, B, bp, True)
g = lambda: ops.latent_bridge_bwd(dy, w2, dist, eps, w1, dd, dx, k, n_in, d, n_out, n_in, n_out, 0, B, bp, 1e-3, act=x)
h = lambda: ops.latent_bridge_bwd(dy, w2, dist, eps, None, dd, None, k, n_in, d, n_out, n_in, n_out, 0, B, bp, 1e-3)
print("fwd %.1f us, bwd (with dx, leaky) %.1f us, bwd (ddist only) %.1f us (event pair around one launch, launch overhead included)" % (timeit(f), timeit(g), timeit(h)))
e = lambda: None
print("empty pair %.1f us" % timeit(e))
