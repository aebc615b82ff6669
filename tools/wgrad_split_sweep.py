"""Dev tool (GPU box): weight-gradient GEMM time vs split (fixed cost per workgroup vs MFMA work)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256; w = eng.workspace(B); bp = w["bp"]; dev = eng.device
def timeit(fn, reps=15):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in + 400000, bp, device=dev); dy = torch.randn(cout * op.t_out + 400000, bp, device=dev)
    per = op.n_tiles * op.wg.mt * 16 * 64
    kc = op._k_chunks(bp)
    line = "%s tiles %d mt %d k_chunks %d:" % (name, op.n_tiles, op.wg.mt, kc)
    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
        if sp > kc // 2: break
        slab = torch.empty(sp * per, device=dev)
        op._split_tuned[bp] = sp
        t = timeit(lambda: op.wgrad_gemm(dy, x, bp, slab))
        line += "  s%d %.0f" % (sp, t)
    print(line, flush=True)
