"""Dev tool (GPU box): in-kernel s_memtime accounting of the v3 window kernel (diagnostic build, never shipped)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["HMVAE_STAMP"] = "1"
from hm_vae_amd import build as B
B.build(force=True, verbose=False)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); Bn = 256; w = eng.workspace(Bn); bp = w["bp"]; eng.pack_weights(); dev = eng.device
convs = dict(zip(["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)], eng.enc_conv + eng.dec_conv))
for spec in sys.argv[1:] or ["dec3:8:1", "dec3:4:2", "dec0:2:4"]:
    layer, chunk, ts = spec.split(":"); chunk, ts = int(chunk), int(ts)
    op = convs[layer]; cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); out = torch.zeros(cout * op.t_out + 400000, bp, device=dev)
    for _ in range(3): op.fwd.run_v3(x, out, bp, None, 0, 0.2, chunk, ts)
    pl = op.fwd.window_plan(chunk, ts); g = op.fwd.g
    nblk = ((g.n_groups * pl["n_chunks"] * (bp // 64) + 7) // 8) * 8
    buf = torch.zeros(nblk * 8 * 8, dtype=torch.int64, device=dev)
    ops.DEBUG_BUF = buf
    op.fwd.run_v3(x, out, bp, None, 0, 0.2, chunk, ts); torch.cuda.synchronize()
    ops.DEBUG_BUF = None
    s = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min(); span = (s[:, 0] + s[:, 6]).max() - t0
    med = lambda c: np.median(s[:, c])
    print("%s v3(%d,%d) un%d win%d: waves %d, kernel span %.0f cyc | per wave median: prologue %.0f load %.0f compute %.0f store %.0f barrier %.0f life %.0f | ideal MFMA cyc/wave %.0f | start p50 %.0f p90 %.0f max %.0f" % (
        layer, chunk, ts, pl["un"], pl["win"], len(s), span, med(1), med(2), med(3), med(4), med(5), med(6), med(7) * 4 * 32,
        np.percentile(s[:, 0] - t0, 50), np.percentile(s[:, 0] - t0, 90), (s[:, 0] - t0).max()))
del os.environ["HMVAE_STAMP"]
B.build(force=True, verbose=False)
