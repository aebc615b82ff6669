"""Dev tool (GPU box): the position-pair weight gradient (csrc/wgrad_pq.hip) against the direct launch (+ reduce, + the up/unpool
pass it needs) per convolution at B = 256, alone-timed with HIP events."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = Trainer(cfg); tr.cuda()
eng = tr.model.engine(); w = eng.workspace(B); bp = w["bp"]
eng.pack_weights()
dev = eng.device
def timeit(fn, reps=20):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
L = eng.p.L
names = ["enc%d" % i for i in range(L)] + ["dec%d" % i for i in range(L)]
convs = eng.enc_conv + eng.dec_conv
ops.WGRAD_PQ = "all"
for name, op in zip(names, convs):
    t = op.pq()
    if t is None:
        print("%s: no position-pair form" % name)
        continue
    cout, cin, kt = op.w_shape
    rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8) + 400000
    rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8) + 400000
    x = torch.randn(rows_x, bp, device=dev); dy = torch.randn(rows_y, bp, device=dev)
    dw = torch.zeros(cout, cin, kt, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    ops.WGRAD_PQ = "0"
    t_d = timeit(lambda: op.backward_weight(dy, x, dw, bp, slab))
    ops.WGRAD_PQ = "all"
    t_p = timeit(lambda: op.backward_weight(dy, None, dw, bp, None, src=x))
    useful = 2.0 * op.useful_macs_per_col * bp
    issued = 6 * 2.0 * t.n_tiles * 128 * 128 * bp
    print("%s: T %d S %d tiles %4d  direct (mode %d split %s) %6.1f us | pairs %6.1f us  %5.1f TF eff  %6.1f TF bf16 issued (%.2f of peak)"
          % (name, t.T, t.S, t.n_tiles, op.wg_mode(bp), op._split_tuned.get(bp), t_d, t_p, useful / t_p / 1e6, issued / t_p / 1e6,
             issued / t_p / 1e6 / 2516.6))
