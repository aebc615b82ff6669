"""Dev tool (GPU box): FK backward kernels at the training size (16 384 frames) and config 4's (262 144) for library
variants built with different occupancy targets (HMVAE_LIB=...).  One process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from hm_vae_amd import ops
    dev = torch.device("cuda:0")
    off = torch.from_numpy(np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "skeleton_offsets.npy"))).to(dev)
    par = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
    def timed(fn, it=12):
        for _ in range(3): fn()
        ev = []
        for _ in range(it):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        return sorted(x.elapsed_time(y) for x, y in ev)[it // 2] * 1e3
    out = []
    for B in (256, 4096):
        T = 64; n = B * T; bp = ops.pad_batch(B)
        r6 = torch.randn(n, 24, 6, device=dev); dR = torch.randn(n, 24, 9, device=dev); dP = torch.randn(n, 24, 3, device=dev)
        d6 = torch.empty(n, 24, 6, device=dev)
        t1 = timed(lambda: ops.fk_bwd(r6, 6, off, 0, par, n, dR, dP, d6))
        dec, g6 = torch.randn(144 * T, bp, device=dev), torch.randn(144 * T, bp, device=dev)
        gR, gP = torch.randn(216 * T, bp, device=dev), torch.randn(72 * T, bp, device=dev)
        dd = torch.empty(144 * T, bp, device=dev); part = torch.zeros(ops.decode_tail_loss_blocks(T, B, bp), 3, device=dev)
        t2 = timed(lambda: ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 1.0 / n, 1.0 / n, 1.0 / n, dd, part))
        out.append("N=%6d: fk_bwd<6> %7.1f us  tail_loss %7.1f us" % (n, t1, t2))
    print(os.path.basename(os.environ.get("HMVAE_LIB", "default")), " | ".join(out), flush=True)
else:
    for occ in (1, 2, 3, 4):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, HMVAE_LIB=os.path.join(ROOT, "hm_vae_amd", "lib", "libhmvae_occ%d.so" % occ)))
