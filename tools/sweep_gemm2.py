"""Dev tool (GPU box): (cw, ksplit, xcd mode) sweep of the v2 gather-GEMM per layer side."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256; w = eng.workspace(B); bp = w["bp"]; eng.pack_weights(); dev = eng.device
def timeit(fn, reps=12):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
tot_auto = tot_best = 0
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in + 400000, bp, device=dev); dy = torch.randn(cout * op.t_out + 400000, bp, device=dev)
    for dirn, side, src, nrows in (("fwd", op.fwd, x, cout * op.t_out + 400000), ("bwd", op.bwd, dy, cin * op.t_in + 400000)):
        if side is None: continue
        out = torch.zeros(nrows, bp, device=dev)
        res = []
        for xm in (0, 1):
            ops.XCD_MODE = xm
            for cw in (1, 2, 4):
                for ks in (1, 2, 4, 8):
                    side.tune = (cw, ks)
                    res.append((timeit(lambda: side.run_v2(src, out, bp)), cw, ks, xm))
        ops.XCD_MODE = None
        side.tune = None
        side.autotune(src, out, bp, None, 0, 0.2)
        t_auto = timeit(lambda: side.run(src, out, bp))
        res.sort()
        tot_auto += t_auto; tot_best += res[0][0]
        print("%s %s: auto %s xcd%d %.1f us | best %.1f (cw%d ks%d xcd%d) | top: %s" % (name, dirn, side._tuned.get(bp), side.xcd_mode(), t_auto, res[0][0], res[0][1], res[0][2], res[0][3],
              " ".join("cw%dks%dx%d:%.0f" % (r[1], r[2], r[3], r[0]) for r in res[:5])), flush=True)
print("sum auto %.0f best %.0f" % (tot_auto, tot_best))
