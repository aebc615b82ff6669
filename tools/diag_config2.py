"""Diagnostic: the benched config-2 step (B = 256) against the float64 and fp32 oracle, every tensor, with the fp32
oracle's own distance from float64 as the yardstick.  python tools/diag_config2.py [p1|p2]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import hmvae_oracle as O
from helpers import load_cfg, offsets, rel_err, scaled_err
import test_fullsize_gpu as TF

phase = sys.argv[1] if len(sys.argv) > 1 else "p2"
dev = torch.device("cuda:0")
cfg = load_cfg("len64")
B, T = 256, 64
plan, tr, seeded = TF._seeded_trainer(cfg, dev)
batch = O.synthetic_batch(B, T, seed=11)
eps = O.level_eps(plan, B, seed=12)
it = 0 if phase == "p1" else cfg["iteration_interval"]
tr.model.eps_override = eps
out = tr.model(batch, cfg, it)
eng = tr.model.engine()
w = eng.workspace(B)
L = plan.L
dec = TF._bct(w["D%d" % (L - 1)], B, 144, T).numpy()
ddec = TF._bct(w["dD%d" % (L - 1)], B, 144, T).numpy()
branch = TF._device_branches(eng, w, B)
mine = {n: (None if p.grad is None else p.grad.detach().cpu().numpy()) for n, p in tr.model.named_parameters() if p.requires_grad}
res = {}
for dtype in (torch.float64, torch.float32):
    orc = O.OracleVAE(cfg, seeded, offsets=offsets(), dtype=dtype)
    orc.branch = branch
    ref = orc.train_forward(batch, it, eps)
    print(dtype, "ambiguous", orc.n_ambiguous, "losses", [float(ref[k].detach()) for k in ("total", "kl", "rec_6d", "rec_rot", "rec_pose")])
    res[dtype] = dict(dec=ref["dec_out"].detach().numpy().astype(np.float64), ddec=ref["dec_out"].grad.numpy().astype(np.float64),
                      grads={n: (None if p.grad is None else p.grad.numpy().astype(np.float64)) for n, p in orc.trainable()})
print("device losses", [float(x) for x in out[:5]])
r64, r32 = res[torch.float64], res[torch.float32]
print("%-40s %10s %10s | %10s %10s | %10s %10s" % ("tensor", "dev/64 rel", "scaled", "dev/32 rel", "scaled", "32/64 rel", "scaled"))
def row(name, a, b64, b32):
    print("%-40s %10.2e %10.2e | %10.2e %10.2e | %10.2e %10.2e" % (name, rel_err(a, b64), scaled_err(a, b64, 1e-4), rel_err(a, b32),
          scaled_err(a, b32, 1e-4), rel_err(b32, b64), scaled_err(b32, b64, 1e-4)))
row("dec_out", dec, r64["dec"], r32["dec"])
row("d_dec_out", ddec, r64["ddec"], r32["ddec"])
for n in r64["grads"]:
    if r64["grads"][n] is None:
        continue
    row(n, mine[n], r64["grads"][n], r32["grads"][n])
# d_dec_out by lift conditioning
c6 = torch.from_numpy(r64["dec"]).permute(0, 2, 1).reshape(B, T, 24, 6)
sin = TF._lift_conditioning(c6).numpy()                       # (B, T, 24)
def per_joint(x):
    return np.abs(torch.from_numpy(x).permute(0, 2, 1).reshape(B, T, 24, 6).numpy()).max(axis=-1)
e_dev = per_joint(ddec - r64["ddec"])
e_32 = per_joint(r32["ddec"] - r64["ddec"])
mag = per_joint(r64["ddec"])
print("max |ddec|", mag.max(), "rms", np.sqrt((r64["ddec"] ** 2).mean()))
for lo, hi in ((0, 1e-3), (1e-3, 1e-2), (1e-2, 0.05), (0.05, 0.2), (0.2, 1.01)):
    m = (sin >= lo) & (sin < hi)
    if m.any():
        print("sin in [%g, %g): n=%d  max|ref| %.3e  max err dev %.3e  max err fp32-oracle %.3e  max err*sin^2 dev %.3e" %
              (lo, hi, m.sum(), mag[m].max(), e_dev[m].max(), e_32[m].max(), (e_dev[m] * sin[m] ** 2).max()))
# candidate per-joint bounds
a = c6[..., 0:3].double(); b = c6[..., 3:6].double()
na = a.norm(dim=-1); nb = b.norm(dim=-1)
x = a / na[..., None]
bperp = torch.linalg.cross(x, b).norm(dim=-1)
cnum = (torch.maximum(na, nb) / torch.minimum(na, bperp)).numpy()
rms = np.sqrt((r64["ddec"] ** 2).mean())
for K in (8, 32, 128):
    for P in (1, 2):
        bound = (mag + rms) * (1e-4 + K * 1.19e-7 * cnum ** P)
        print("K=%d P=%d: dev max ratio %.3f   fp32-oracle max ratio %.3f" % (K, P, (e_dev / bound).max(), (e_32 / bound).max()))
idx = np.argsort((e_dev / ((mag + rms) * 1e-4)).reshape(-1))[::-1][:12]
for i in idx:
    print("err %.2e  e32 %.2e  ref %.2e  sin %.3f  |a| %.3f |b| %.3f bperp %.3f c %.1f" % (e_dev.reshape(-1)[i], e_32.reshape(-1)[i], mag.reshape(-1)[i],
          sin.reshape(-1)[i], na.reshape(-1)[i], nb.reshape(-1)[i], bperp.reshape(-1)[i], cnum.reshape(-1)[i]))
