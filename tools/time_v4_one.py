"""Dev tool (GPU box): time ONE v4 launch shape of one layer side (HIP events, median of 20).
usage: [HMVAE_LIB=variant.so] python tools/time_v4_one.py layer side mtw nc split"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
name, sname, mtw, nc, sp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(256); bp = w["bp"]
eng.pack_weights()
dev = eng.device
L = eng.p.L
op = dict(zip(["enc%d" % i for i in range(L)] + ["dec%d" % i for i in range(L)], eng.enc_conv + eng.dec_conv))[name]
plist = {id(o): wt for o, wt in eng._pack_list()}
cout, cin, kt = op.w_shape
rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 400000
rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 400000
x = torch.randn(rows_x, bp, device=dev); y = torch.empty(rows_y, bp, device=dev)
side, src, dst = (op.fwd, x, y) if sname == "fwd" else (op.bwd, y, x)
sd = side.v4_side(mtw)
ops.MultiPack(dev).run([(sd, plist[id(op)])])
for _ in range(3):
    sd.run_v4(src, dst, bp, None, 0, 0.2, nc, sp)
evs = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sd.run_v4(src, dst, bp, None, 0, 0.2, nc, sp); b.record(); evs.append((a, b))
torch.cuda.synchronize()
t = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[10] * 1e3
print("%s: %s %s mtw %d nc %d split %d: %.1f us" % (os.path.basename(os.environ.get("HMVAE_LIB", "production")), name, sname, mtw, nc, sp, t))
