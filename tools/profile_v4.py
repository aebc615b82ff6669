"""Dev tool (GPU box): the bf16-plane gather-GEMM (gemm_v4.hip) against the v2 / v3 pick, per layer side at B=256
(HIP events): every launch shape (rows per group, columns per unit, K split) with its time and issued rate."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HMVAE_V4", "0")          # the incumbent is tuned without v4; the shapes are swept by hand below
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
ops.USE_V4 = "auto"
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
ops.USE_V4 = "auto"
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]
eng.pack_weights()
dev = eng.device
def timeit(fn, reps=12):
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
plist = {id(op): wt for op, wt in eng._pack_list()}
tot2 = tot4 = tot41 = 0.0
for name, op in zip(names, convs):
    if only and name not in only:
        continue
    cout, cin, kt = op.w_shape
    rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 400000
    rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 400000
    x = torch.randn(rows_x, bp, device=dev); y = torch.empty(rows_y, bp, device=dev)
    for sname, side, src, dst in (("fwd", op.fwd, x, y), ("bwd", op.bwd, y, x)):
        if side is None:
            continue
        side._last_dense = plist[id(op)]
        ops.USE_V4 = "0"
        side.run(src, dst, bp)
        ops.USE_V4 = "auto"
        t2 = timeit(lambda: side._run_incumbent(src, dst, bp, None, 0, 0.2))
        n_items = sum(len(l) for l in side.pairs_host)
        issued2 = 2.0 * side.g.mfma_cost() * n_items * bp
        line = "%s %s: v2 %s %6.1f us %5.1f TF issued (groups %d, items %d)" % (name, sname, side._tuned.get(bp), t2, issued2 / t2 / 1e6, side.g.n_groups, n_items)
        print(line, flush=True)
        best = best1 = None
        for mtw in (1, 2):
            sd = side.v4_side(mtw)
            if sd is None:
                print("   v4 mtw %d: unsupported" % mtw); continue
            ops.MultiPack(dev).run([(sd, plist[id(op)])])
            g4 = sd.g
            issued4 = 2.0 * (g4.g_nc4 * 4).sum() * 16 * g4.mt * n_items * bp      # as if every tile of a group were live
            live = 2.0 * sum(int(nc4) * 4 * 16 * int(nt) for nc4, nt in zip(g4.g_nc4, sd.g_ntiles.cpu().numpy())) * n_items * bp
            for nc in (128, 64):
                if bp % nc or (mtw, nc) not in sd.V4_SHAPES: continue
                res = []
                for sp in sd.v4_splits(bp, nc):
                    t4 = timeit(lambda: sd.run_v4(src, dst, bp, None, 0, 0.2, nc, sp), reps=8)
                    res.append((sp, t4))
                    if best is None or t4 < best[0]:
                        best = (t4, mtw, nc, sp)
                    if best1 is None or t4 < best1:
                        best1 = t4
                print("   v4 mtw %d nc %3d groups %3d wgs/split %4d live-tile flops %5.2f G: %s" % (
                    mtw, nc, g4.n_groups, g4.n_groups * sd.n_pos * (bp // nc), live / 1e9,
                    "  ".join("S%d %.1f" % r for r in res)), flush=True)
        if best is None:
            continue
        print("   best v4 %s: %.1f us vs v2 %.1f us  (x%.2f)" % (best[1:], best[0], t2, t2 / best[0]), flush=True)
        tot2 += t2; tot4 += min(t2, best[0]); tot41 += min(t2, best1)
print("sum v2 %.1f us, sum min(v2, v4) %.1f us" % (tot2, tot4))
