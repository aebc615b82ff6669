"""Dev tool (GPU box): weight-gradient GEMM with every operand row aliased to row 0 (all loads hit L1/L2) vs normal:
tells memory latency apart from issue limits."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256; w = eng.workspace(B); bp = w["bp"]; dev = eng.device
def timeit(fn, reps=20):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in + 400000, bp, device=dev); dy = torch.randn(cout * op.t_out + 400000, bp, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    t0 = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab))
    keep = (op.wg_chan_rows.clone(), op.dz_rows.clone())
    op.wg_chan_rows.zero_(); op.dz_rows.clamp_(max=0)
    t1 = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab))
    op.wg_chan_rows.copy_(keep[0]); op.dz_rows.copy_(keep[1])
    useful = 2.0 * op.useful_macs_per_col * bp
    print("%s wgrad gemm only: normal %.1f us (%.1f TF) | all rows aliased %.1f us (%.1f TF)" % (name, t0, useful / t0 / 1e6, t1, useful / t1 / 1e6))
