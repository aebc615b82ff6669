"""Dev tool (GPU box, library built with HMVAE_STAMP=1): where a workgroup of the v4 gather-GEMM spends its life.
Stamps (s_memrealtime: one 100 MHz clock for the whole chip, 10 ns ticks -> printed in us): 0 start, 1 tables in LDS, 2 ring filled (loop entry), 3 loop end,
4 part stored + drained, 5 counter returned, 6 end.  usage: python tools/stamp_v4.py layer side mtw nc split"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
name, sname, mtw, nc, sp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
B = 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]
eng.pack_weights()
dev = eng.device
L = eng.p.L
names = ["enc%d" % i for i in range(L)] + ["dec%d" % i for i in range(L)]
op = dict(zip(names, eng.enc_conv + eng.dec_conv))[name]
plist = {id(o): wt for o, wt in eng._pack_list()}
cout, cin, kt = op.w_shape
rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 400000
rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 400000
x = torch.randn(rows_x, bp, device=dev); y = torch.empty(rows_y, bp, device=dev)
side, src, dst = (op.fwd, x, y) if sname == "fwd" else (op.bwd, y, x)
sd = side.v4_side(mtw)
ops.MultiPack(dev).run([(sd, plist[id(op)])])
g = sd.g
n_wg = g.n_groups * side.n_pos * (bp // nc) * sp
nblk = (n_wg + 7) // 8 * 8
for _ in range(3):
    sd.run_v4(src, dst, bp, None, 0, 0.2, nc, sp)
torch.cuda.synchronize()
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
ops.DEBUG_BUF = buf
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); sd.run_v4(src, dst, bp, None, 0, 0.2, nc, sp); b.record()
torch.cuda.synchronize()
ops.DEBUG_BUF = None
t = buf.cpu().numpy().reshape(nblk, 8).astype(np.float64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
tick = 0.01          # us per s_memtime tick (100 MHz)
print("%s %s mtw %d nc %d split %d: %d workgroups, kernel %.1f us (event), first start -> last end %.1f us" % (
    name, sname, mtw, nc, sp, len(t), a.elapsed_time(b) * 1e3, (t[:, :7].max() - t0) * tick))
d = np.diff(t[:, :7], axis=1) * tick
lab = ["tables", "ring fill", "K loop", "tile->part", "atomic", "reduce+store"]
for i, l in enumerate(lab):
    col = d[:, i][t[:, i + 1] > 0]
    if len(col):
        print("  %-12s mean %6.2f  p50 %6.2f  p90 %6.2f us  (n %d)" % (l, col.mean(), np.median(col), np.percentile(col, 90), len(col)))
end = np.where(t[:, 6] > 0, t[:, 6], t[:, 5])
life = (end - t[:, 0]) * tick
st = np.sort(t[:, 0] - t0) * tick
print("  workgroup life mean %.2f p90 %.2f us; starts after the first: p10 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f us; last end %.1f us" % (
    life.mean(), np.percentile(life, 90), *[np.percentile(st, q) for q in (10, 50, 75, 90)], st.max(), (end.max() - t0) * tick))
# how many workgroups are alive at once (sampled every 2 us)
for ts in np.arange(0, (end.max() - t0) * tick, 4.0):
    alive = int(((t[:, 0] - t0) * tick <= ts).sum() - ((end - t0) * tick <= ts).sum())
    print("    t=%5.1f us: %4d workgroups alive" % (ts, alive))
