"""Dev tool (GPU box): v2 (autotuned) vs v3 window kernel per conv layer and side at B=256."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]
eng.pack_weights()
dev = eng.device
def timeit(fn, reps=20):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
L = eng.p.L
names = ["enc%d" % i for i in range(L)] + ["dec%d" % i for i in range(L)]
convs = eng.enc_conv + eng.dec_conv
tot2 = tot3 = 0.0
for name, op in zip(names, convs):
    cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); y = torch.empty(cout * op.t_out + 400000, bp, device=dev)
    dy = torch.randn(y.shape[0], bp, device=dev); dx = torch.empty_like(x)
    for sname, side, s, d in (("fwd", op.fwd, x, y), ("bwd", op.bwd, dy, dx)):
        if side is None: continue
        g = side.g
        issued = 2.0 * g.mfma_cost() * kt * len(side.pairs_host) * bp
        side.run(s, d, bp)
        t2 = timeit(lambda: side.run_v2(s, d, bp))
        line = "%s %s: v2 %s %6.1f us %5.1f TF |" % (name, sname, side.choose(bp), t2, issued / t2 / 1e6)
        best = None
        for chunk, ts in ((8, 1), (4, 2), (2, 4), (1, 8)):
            pl = side.window_plan(chunk, ts)
            if pl is None: continue
            t3 = timeit(lambda: side.run_v3(s, d, bp, None, 0, 0.2, chunk, ts))
            line += " v3(%d,%d,un%d,w%d) %6.1f" % (chunk, ts, pl["un"], pl["win"], t3)
            if best is None or t3 < best: best = t3
        line += " | best v3 %6.1f us %5.1f TF  (%.2fx)" % (best, issued / best / 1e6, t2 / best)
        tot2 += t2; tot3 += min(t2, best)
        print(line, flush=True)
print("total v2 %.1f us, with v3 where faster %.1f us" % (tot2, tot3))
