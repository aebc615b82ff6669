"""Dev tool (GPU box): the wide weight-gradient tiles (wg_mode 4) against mode 2 / 3 per layer at B = 256: GEMM alone and
GEMM + reduce (or fused epilogue), every split candidate."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]; dev = eng.device
ops.AUTOTUNE = False
def timeit(fn, reps=12):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 1000
    rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 1000
    x = torch.randn(rows_x, bp, device=dev); dy = torch.randn(rows_y, bp, device=dev)
    dw = torch.zeros(cout, cin, kt, device=dev); slab = torch.empty(op.slab_floats(bp), device=dev)
    useful = 2.0 * op.useful_macs_per_col * bp
    wd = op.wide()
    print("%s: std tiles %d mt %d | wide %s" % (name, op.n_tiles, op.wg.mt, None if wd is None else
          "tiles %d mt %d units %d" % (wd.n_tiles, wd.g.mt, op.pair_tab(wd)[1])), flush=True)
    for md in ("2", "3", "4"):
        if (md == "3" and not op.pair_eligible()) or (md == "4" and wd is None):
            continue
        ops.WGRAD_LDS = md
        best = None
        for sp in (op.wide_split_candidates(bp) if md == "4" else op.split_candidates(bp)[:8]):
            op._split_tuned[bp] = sp
            t_g = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab))
            t_w = timeit(lambda: op.backward_weight(dy, x, dw, bp, slab))
            print("   mode %s split %2d: gemm %7.1f us   gemm+reduce/fused %7.1f us  (%5.1f TF eff)" % (md, sp, t_g, t_w, useful / t_w / 1e6), flush=True)
