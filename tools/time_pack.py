import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd.utils_common import get_config
cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); eng.workspace(256)
for _ in range(3): eng.pack_weights(side=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): eng.pack_weights(side=False)
b.record(); torch.cuda.synchronize()
print("pack_weights: %.1f us per step" % (a.elapsed_time(b) / 20 * 1e3))
# per side
from hm_vae_amd import ops
names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
P = dict(tr.model.named_parameters())
lst = eng._pack_list()
for (op, wt) in lst:
    for sn, side in (("f", op.fwd), ("b", op.bwd)):
        if side is None: continue
        mp = ops.MultiPack(eng.device)
        for _ in range(3): mp.run([(side, wt)])
        a.record()
        for _ in range(10): mp.run([(side, wt)])
        b.record(); torch.cuda.synchronize()
        g = side.g
        print("  w%s %s: kt %d groups %d packed %.1f MB blocks %d cb %d: %.1f us  (%.0f GB/s)" % (tuple(wt.shape), sn, g.kt, g.n_groups, g.packed_size * 4 / 1e6, side.pack_blocks(), side.pack_cb(), a.elapsed_time(b) / 10 * 1e3, g.packed_size * 4 / 1e3 / (a.elapsed_time(b) / 10 * 1e3)))
