import torch, numpy as np, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hm_vae_amd import ops
dev = torch.device("cuda:0")
B, T = 4096, 64
n = B * T
bp = ops.pad_batch(B)
off = torch.from_numpy(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hm_vae_amd/assets/skeleton_offsets.npy"))).to(dev)
dec = torch.randn(144 * T, bp, device=dev)
c6 = torch.empty(B, T, 24, 6, device=dev); R = torch.empty(B, T, 24, 3, 3, device=dev); P = torch.empty(B, T, 24, 3, device=dev)
def timed(fn, iters=30):
    for _ in range(3): fn()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sum(x.elapsed_time(y) for x, y in evs) / len(evs) * 1e3
t = timed(lambda: ops.decode_tail_fwd(dec, off, T, B, bp, None, R, P)); print("groups", os.environ.get("HMVAE_TF_GROUPS"), "tail<false> %.1f us %.2f TB/s" % (t, 1728 * n / t / 1e6))
t = timed(lambda: ops.decode_tail_fwd(dec, off, T, B, bp, c6, R, P)); print("groups", os.environ.get("HMVAE_TF_GROUPS"), "tail<true>  %.1f us %.2f TB/s" % (t, 2304 * n / t / 1e6))
