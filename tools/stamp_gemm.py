"""Dev tool (GPU box): in-kernel s_memtime stamps of the gather-GEMM (diagnostic build, never the shipped one).
Builds a stamped library into a temp dir copy? No: rebuilds in place with HMVAE_STAMP=1, runs, then rebuilds clean."""
import sys, os, subprocess
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
for layer in sys.argv[1:] or ["enc0", "dec3", "dec0"]:
    op = convs[layer]; cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); out = torch.zeros(cout * op.t_out + 400000, bp, device=dev)
    for _ in range(3): op.fwd.run_v2(x, out, bp)
    g = op.fwd.g; cw, ks = op.fwd.choose(bp)[:2]
    nblk = ((g.n_groups * op.fwd.n_pos * (bp // (64 * cw)) + 7) // 8) * 8
    buf = torch.zeros(nblk * ks * 4, dtype=torch.int64, device=dev)
    ops.DEBUG_BUF = buf
    op.fwd.run_v2(x, out, bp); torch.cuda.synchronize()
    ops.DEBUG_BUF = None
    s = buf.cpu().numpy().reshape(-1, 4).astype(np.float64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    pro, loop, epi, life = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
    span = s[:, 3].max() - t0
    print("%s fwd cw%d ks%d: waves %d | kernel span %.0f cyc | per-wave median: prologue %.0f, loop %.0f, epilogue %.0f, life %.0f | start spread p50 %.0f p90 %.0f | items/wave ~%.0f" % (
        layer, cw, ks, len(s), span, np.median(pro), np.median(loop), np.median(epi), np.median(life),
        np.percentile(s[:, 0] - t0, 50), np.percentile(s[:, 0] - t0, 90), float(g.g_nc4.mean()) / ks * 15))
    # MFMA cycles per wave in loop: items * 4 MFMAs * 32 cycles * mt*cw
    print("   ideal MFMA cycles per wave in loop ~%.0f" % (float(g.g_nc4.mean()) / ks * 15 * 4 * 32 * g.mt * cw))
del os.environ["HMVAE_STAMP"]
B.build(force=True, verbose=False)
