"""Dev tool (GPU box): per-layer timing of every MFMA launch of the engine at B=256 (HIP events, 20 reps)."""
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
tot = {"fwd": 0, "bwd": 0, "wg": 0}
print("B=%d bp=%d" % (B, bp))
for name, op in zip(names, convs):
    cout, cin, kt = op.w_shape
    # exact-size operands (mode 3 splits the whole tensor per call); the gather sides may address rows of a wider buffer
    rows_x = max(cin * op.t_in, int(op.wg_chan_rows.max().item()) + op.geom["t_src"] * op.x_ts + 8)
    rows_y = max(cout * op.t_out, int(op.dz_rows.max().item()) + op.geom["t_dst"] * op.y_ts + 8)
    if op.folded:
        rows_x, rows_y = rows_x + 400000, rows_y + 400000
    x = torch.randn(rows_x, bp, device=dev); y = torch.empty(rows_y, bp, device=dev)
    dy = torch.randn(y.shape[0], bp, device=dev); dx = torch.empty_like(x); dw = torch.zeros(cout, cin, kt, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    useful = 2.0 * op.useful_macs_per_col * bp
    gf, gb = op.fwd.g, (op.bwd.g if op.bwd else None)
    t_f = timeit(lambda: op.forward(x, y, bp))
    issued_f = 2.0 * gf.mfma_cost() * sum(len(l) for l in op.fwd.pairs_host) * bp
    line = "%s fwd: tune %s groups %3d pos %2d  %7.1f us %5.1f TF (eff) %5.1f TF issued " % (name, op.fwd.choose(bp), gf.n_groups, op.fwd.n_pos, t_f, useful/t_f/1e6, issued_f/t_f/1e6)
    tot["fwd"] += t_f
    if op.bwd:
        t_b = timeit(lambda: op.backward_data(dy, dx, bp)); tot["bwd"] += t_b
        issued_b = 2.0 * gb.mfma_cost() * sum(len(l) for l in op.bwd.pairs_host) * bp
        line += "| bwd: tune %s groups %3d pos %2d %7.1f us %5.1f TF %5.1f TF issued " % (op.bwd.choose(bp), gb.n_groups, op.bwd.n_pos, t_b, useful/t_b/1e6, issued_b/t_b/1e6)
    t_w = timeit(lambda: op.backward_weight(dy, x, dw, bp, slab)); tot["wg"] += t_w
    t_g = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab))
    issued_w = op.n_tiles * op.wg.mt * 16 * 64 * 2.0 * op.geom["t_dst"] * bp
    line += "| wgrad: tiles %4d mt %d split %2s mode %d %7.1f us (gemm %6.1f) %5.1f TF eff, gemm alone %5.1f TF issued" % (
        op.n_tiles, op.wg.mt, op._split_tuned.get(bp), op.wg_mode(bp), t_w, t_g, useful/t_w/1e6, issued_w/t_g/1e6)
    print(line)
    if os.environ.get("WG_SWEEP"):
        for md in (0, 1, 2, 3):
            if (md == 1 and not op.lds_eligible()) or (md == 2 and not op.bf3_eligible()) or (md == 3 and not op.pair_eligible()):
                continue
            for sp in op.split_candidates(bp):
                op._mode_tuned[bp], op._split_tuned[bp] = md, sp
                t_g = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab), reps=8)
                t_w = timeit(lambda: op.backward_weight(dy, x, dw, bp, slab), reps=8)
                print("      mode %d split %2d: gemm %7.1f us  gemm+reduce %7.1f us  %5.1f TF issued" % (md, sp, t_g, t_w, issued_w/t_g/1e6))
print("totals us:", tot, "sum", sum(tot.values()))
