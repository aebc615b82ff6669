import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256; w = eng.workspace(B); bp = w["bp"]; eng.pack_weights(); dev = eng.device
def timeit(fn, reps=15):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
convs = dict(zip(["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)], eng.enc_conv + eng.dec_conv))
for layer in ("dec3", "enc0", "dec2"):
    op = convs[layer]; cout, cin, kt = op.w_shape
    x = torch.randn(cin * op.t_in, bp, device=dev); out = torch.zeros(cout * op.t_out + 400000, bp, device=dev)
    res = []
    for abl in (0, 1, 2, 3):
        res.append(timeit(lambda: op.fwd.run_v2(x, out, bp, None, abl << 8)))
    print(layer, "fwd: full %.0f us | no B loads %.0f | no A loads %.0f | no MFMA %.0f" % tuple(res))
