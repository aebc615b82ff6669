"""Dev tool: launch one layer's gather-GEMM (forward or backward-data) a few times (for rocprofv3 --pmc passes).
usage: gather_only.py dec3 f|b"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
which = sys.argv[1] if len(sys.argv) > 1 else "dec3"
side = sys.argv[2] if len(sys.argv) > 2 else "f"
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256
w = eng.workspace(B); bp = w["bp"]; dev = eng.device
eng.pack_weights(side=False)
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
op = dict(zip(names, eng.enc_conv + eng.dec_conv))[which]
sd = op.fwd if side == "f" else op.bwd
rows_src = int(sd.chan_rows.max().item()) + 70 * max(sd.src_ts, 1) + 8
rows_dst = int(sd.dst_rows.max().item()) + 70 * max(sd.dst_ts, 1) + 8
src = torch.randn(rows_src, bp, device=dev); dst = torch.zeros(rows_dst, bp, device=dev)
for _ in range(12):
    sd.run(src, dst, bp)
torch.cuda.synchronize()
print("done", which, side, sd._tuned.get(bp), "groups", sd.g.n_groups, "mt", sd.g.mt, "n_pos", sd.n_pos)
