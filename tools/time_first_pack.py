"""Dev tool: how long the step's first launches take (first pack = encoder layer 0's layout, mid pack, aos_to_soa)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd import ops
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256
w = eng.workspace(B); bp = w["bp"]
def timeit(fn, reps=30):
    for _ in range(3): fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize(); return sorted(x.elapsed_time(y) for x, y in evs)[len(evs)//2] * 1e3
lst = eng._pack_list()
mp = ops.MultiPack(eng.device)
enc_ops = set(id(o) for o in eng.enc_conv + eng.enc_lin)
first = [(op.fwd, wt) for op, wt in lst if id(op) in enc_ops]
print("first pack, all encoder fwd layouts: %.1f us" % timeit(lambda: mp.run(first)))
for i in range(4):
    one = [(eng.enc_conv[i].fwd, dict(lst_ := [(id(o), wt) for o, wt in lst])[id(eng.enc_conv[i])])]
    mp1 = ops.MultiPack(eng.device)
    print("  enc%d fwd only: %.1f us  (vt=%d, packed %.2f MB)" % (i, timeit(lambda: mp1.run(one)), eng.enc_conv[i].fwd.g.kt, eng.enc_conv[i].fwd.g.packed_size * 4 / 1e6))
late = [(op.fwd, wt) for op, wt in lst if id(op) not in enc_ops] + [(op.bwd, wt) for op, wt in lst if op.bwd is not None]
mp2 = ops.MultiPack(eng.device)
print("late pack: %.1f us" % timeit(lambda: mp2.run(late)))
r6 = torch.randn(B, 64, 144, device=eng.device)
print("aos_to_soa X0: %.1f us" % timeit(lambda: ops.aos_to_soa(r6, w["X0"], B, 64, 144, bp)))
