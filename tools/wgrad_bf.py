"""Dev tool (GPU box): weight-gradient GEMM per conv layer, fp32 MFMA vs the bf16x3-plane variant (incl. the operand
split), same split; and how far the two results are apart."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["HMVAE_WGRAD_BF16X3"] = "0"; os.environ["HMVAE_WGRAD_LDS"] = "auto"
from hm_vae_amd import ops, _lib
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
import ctypes as C
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = eng.workspace(B); bp = w["bp"]; dev = eng.device
def timeit(fn, reps=20):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
tot = [0.0, 0.0]
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    rows_x = int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8
    rows_y = int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8
    x = torch.randn(rows_x, bp, device=dev); dy = torch.randn(rows_y, bp, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    dw = [torch.zeros(cout, cin, kt, device=dev) for _ in range(2)]
    res = []
    for bf in (False, True):          # second column: the LDS-staged kernel (mode 1), fp32
        best = None
        for c in op.split_candidates(bp) + (["p"] if op._k_chunks(bp) >= 8 else []):
            op._split_tuned[bp] = c; op._bf_tuned[bp] = False; op._mode_tuned[bp] = int(bf)
            def once():
                op.wgrad_gemm(dy, x, bp, slab)
                r = op.reduce_desc(slab, dw[int(bf)], bp, False)
                _lib.check(_lib.load().hmvae_wgrad_reduce(C.byref(r), _lib.stream()), "reduce")
            t = timeit(once, 8)
            if best is None or t < best[0]: best = (t, c)
        op._split_tuned[bp] = best[1]
        t = timeit(once)
        once(); torch.cuda.synchronize()
        res.append((t, best[1]))
    diff = float((dw[0] - dw[1]).abs().max() / dw[0].abs().max())
    useful = 2.0 * op.useful_macs_per_col * bp
    print("%s mt%d direct %.1f us (split %s, %.1f TF) | lds %.1f us (split %s, %.1f TF) | max rel diff %.2e"
          % (name, op.wg.mt, res[0][0], res[0][1], useful / res[0][0] / 1e6, res[1][0], res[1][1], useful / res[1][0] / 1e6, diff))
    tot[0] += res[0][0]; tot[1] += res[1][0]
print("sum direct %.1f us, lds %.1f us" % tuple(tot))
