"""Dev tool (GPU box): time the slab -> dense weight-gradient reduction and the bias row sums per layer."""
import sys, os, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
from hm_vae_amd import ops, _lib
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256; w = eng.workspace(B); bp = w["bp"]; dev = eng.device
def timeit(fn, reps=20):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
tot = 0
for name, op in zip(names, eng.enc_conv + eng.dec_conv):
    cout, cin, kt = op.w_shape
    slab = torch.randn(op.slab_floats(bp), device=dev); dw = torch.zeros(cout, cin, kt, device=dev)
    r = op.reduce_desc(slab, dw, bp)
    t = timeit(lambda: _lib.check(_lib.load().hmvae_wgrad_reduce(C.byref(r), _lib.stream()), "r"))
    mb = (slab.numel() + dw.numel()) * 4 / 1e6
    tot += t
    print("%s reduce: tiles %d mt %d split %d  %.1f MB  %.1f us  %.0f GB/s" % (name, op.n_tiles_t if op.wg_transposed else op.n_tiles, op.reduce_rows() // 16, op.n_split(bp), mb, t, mb / t * 1e3))
print("total", tot)
