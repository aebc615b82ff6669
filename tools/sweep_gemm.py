"""Dev tool (GPU box): per-layer sweep of the v2 gather-GEMM's (np, ksplit) against v1, with a correctness check."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]
eng.pack_weights()
dev = eng.device
def timeit(fn, reps=15):
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
best_total = 0.0; v1_total = 0.0
for name, op in zip(names, convs):
    cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); dy = torch.randn(cout * op.t_out + 400000, bp, device=dev)
    useful = 2.0 * op.useful_macs_per_col * bp
    for dirn, side, src, nrows in (("fwd", op.fwd, x, cout * op.t_out + 400000), ("bwd", op.bwd, dy, cin * op.t_in)):
        if side is None: continue
        ref = torch.zeros(nrows, bp, device=dev); out = torch.zeros(nrows, bp, device=dev)
        t1 = timeit(lambda: side.run_v1(src, ref, bp)); v1_total += t1
        mt = side.g.mt
        res = []
        for np_ in {1: (1, 2, 4), 2: (1, 2), 3: (1,), 4: (1,)}[mt]:
            for ks in (2, 4, 8):
                side.tune = (np_, ks)
                out.zero_()
                side.run_v2(src, out, bp)
                err = float((out - ref).abs().max() / ref.abs().max())
                t = timeit(lambda: side.run_v2(src, out, bp))
                res.append((t, np_, ks, err))
        side.tune = None
        auto = side.choose(bp)
        res.sort()
        best_total += res[0][0]
        t_auto = [r[0] for r in res if (r[1], r[2]) == tuple(auto)][0]
        print("%s %s mt%d groups %3d pos %2d | v1 %7.1f us | best v2 %7.1f us (cw%d ks%d) %5.1f TF eff | auto (cw%d ks%d) %7.1f us | maxerr %.1e | all: %s" % (
            name, dirn, mt, side.g.n_groups, side.n_pos, t1, res[0][0], res[0][1], res[0][2], useful / res[0][0] / 1e6, auto[0], auto[1], t_auto,
            max(r[3] for r in res), " ".join("cw%dks%d:%.0f" % (r[1], r[2], r[0]) for r in res)))
print("sum v1 %.0f us, sum best v2 %.0f us" % (v1_total, best_total))
