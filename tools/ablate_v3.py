"""Dev tool (GPU box): ablations of the v3 window kernel (diagnostic HMVAE_ABLATE build; results are garbage, only the time matters)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["HMVAE_ABLATE"] = "1"
from hm_vae_amd import build as B
B.build(force=True, verbose=False)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); Bn = 256; w = eng.workspace(Bn); bp = w["bp"]; eng.pack_weights(); dev = eng.device
convs = dict(zip(["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)], eng.enc_conv + eng.dec_conv))
def timeit(fn, reps=15):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
for spec in sys.argv[1:] or ["dec3:8:1", "dec3:4:2", "dec0:2:4", "dec1:4:2"]:
    layer, chunk, ts = spec.split(":"); chunk, ts = int(chunk), int(ts)
    op = convs[layer]; cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); out = torch.zeros(cout * op.t_out + 400000, bp, device=dev)
    g = op.fwd.g
    issued = 2.0 * g.mfma_cost() * kt * len(op.fwd.pairs_host) * bp
    res = []
    for abl in (0, 1, 2, 4, 12, 6, 14, 3, 7, 15):
        res.append((abl, timeit(lambda: op.fwd.run_v3(x, out, bp, None, abl << 8, 0.2, chunk, ts))))
    print("%s v3(%d,%d): ideal %.0f us |" % (layer, chunk, ts, issued / 157.3e6), " ".join("abl%d=%.0f" % r for r in res), flush=True)
del os.environ["HMVAE_ABLATE"]
B.build(force=True, verbose=False)
