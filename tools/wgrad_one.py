"""Dev tool: launch one layer's weight gradient a few times in one configuration (for rocprofv3 kernel traces / --pmc passes).
usage: wgrad_one.py LAYER MODE SPLIT [fused|slab|both] [reps]     e.g. wgrad_one.py dec0 4 1 both 10
MODE "pq": the position-pair launch (csrc/wgrad_pq.hip) instead of hmvae_wgrad_gemm."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd import ops
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
which, mode, split = sys.argv[1], sys.argv[2], int(sys.argv[3])
how = sys.argv[4] if len(sys.argv) > 4 else "both"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ops.AUTOTUNE = False
pq = mode == "pq"
if pq:
    ops.WGRAD_PQ = "all"
else:
    ops.WGRAD_LDS = mode
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256
w = eng.workspace(B); bp = w["bp"]; dev = eng.device
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
op = dict(zip(names, eng.enc_conv + eng.dec_conv))[which]
cout, cin, kt = op.w_shape
rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 1000
rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 1000
x = torch.randn(rows_x, bp, device=dev); dy = torch.randn(rows_y, bp, device=dev)
dw = torch.zeros(cout, cin, kt, device=dev); slab = torch.empty(op.slab_floats(bp), device=dev)
op._split_tuned[bp] = split
for _ in range(reps if pq else 0):
    op.backward_weight(dy, None, dw, bp, None, src=x)
for _ in range(0 if pq else reps):
    if how in ("slab", "both"):
        op.wgrad_gemm(dy, x, bp, slab)
    if how in ("fused", "both"):
        op.backward_weight(dy, x, dw, bp, slab)
torch.cuda.synchronize()
print("done", which, "mode", op.wg_mode(bp), "split", split, "fusable", op.fusable(bp))
