"""Dev tool (GPU box): run ONE MFMA launch config repeatedly (for rocprofv3 --pmc).
usage: mb_layer.py LAYER(enc0..dec3) DIR(fwd|bwd|wg) VERSION(v1|v2) [np ks] [reps]"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
layer, dirn, ver = sys.argv[1], sys.argv[2], sys.argv[3]
np_, ks = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (None, None)
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
B = 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]; eng.pack_weights()
convs = dict(zip(["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)], eng.enc_conv + eng.dec_conv))
op = convs[layer]; dev = eng.device
cout, cin, kt = op.w_shape
x = torch.randn(cin * op.t_in, bp, device=dev); dy = torch.randn(cout * op.t_out + 400000, bp, device=dev)
if dirn == "wg":
    dw = torch.zeros(cout, cin, kt, device=dev); slab = torch.empty(op.slab_floats(bp), device=dev)
    fn = lambda: op.backward_weight(dy, x, dw, bp, slab)
else:
    side, src, nrows = (op.fwd, x, cout * op.t_out + 400000) if dirn == "fwd" else (op.bwd, dy, cin * op.t_in)
    out = torch.zeros(nrows, bp, device=dev)
    if np_ is not None: side.tune = (np_, ks)
    fn = (lambda: side.run_v1(src, out, bp)) if ver == "v1" else (lambda: side.run_v2(src, out, bp))
    if ver == "v3":       # np ks are (chunk, ts) here
        side.tune = None
        fn = lambda: side.run_v3(src, out, bp, None, 0, 0.2, np_, ks)
for _ in range(reps): fn()
torch.cuda.synchronize()
print("done", layer, dirn, ver, np_, ks)
