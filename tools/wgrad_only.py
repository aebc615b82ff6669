"""Dev tool: launch one layer's weight-gradient GEMM a few times (for rocprofv3 --pmc passes).  usage: wgrad_only.py dec0 [bf]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd import ops
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
which = sys.argv[1] if len(sys.argv) > 1 else "dec0"
if len(sys.argv) > 2 and sys.argv[2] == "bf":
    ops.WGRAD_BF16X3 = "1"
ops.WGRAD_LDS = {"lds": "1", "bf3": "2"}.get(sys.argv[2] if len(sys.argv) > 2 else "", "0")
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256
w = eng.workspace(B); bp = w["bp"]; dev = eng.device
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
op = dict(zip(names, eng.enc_conv + eng.dec_conv))[which]
rows_x = int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8
rows_y = int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8
x = torch.randn(rows_x, bp, device=dev); dy = torch.randn(rows_y, bp, device=dev)
slab = torch.empty(op.slab_floats(bp), device=dev)
op._split_tuned[bp] = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for _ in range(10):
    op.wgrad_gemm(dy, x, bp, slab)
torch.cuda.synchronize()
print("done", which, op.n_tiles, op.wg.mt)
