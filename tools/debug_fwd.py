import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import hmvae_oracle as O
from helpers import load_cfg, load_model_golden, golden_batch, golden_eps, offsets, rel_err
from hm_vae_amd.trainer_motion_vae import Trainer
from hm_vae_amd import ops
tag = "len64"
g, meta = load_model_golden(tag)
cfg = load_cfg(tag); plan = O.Plan(cfg); seeded = O.name_seeded_params(plan)
tr = Trainer(cfg); sd = tr.model.state_dict(); sd.update({k: v.clone() for k, v in seeded.items()}); tr.model.load_state_dict(sd); tr.cuda()
batch, eps = golden_batch(g), golden_eps(g)
tr.model.eps_override = eps
out = tr.model(batch, cfg, 0)
eng = tr.model.engine(); B = 2; w = eng.workspace(B)
def bct(t, Cn, T):
    o = torch.empty(B, Cn*T, device=t.device); ops.soa_to_aos(t, o, B, 1, Cn*T, t.shape[1]); return o.reshape(B, Cn, T).cpu().numpy()
def run_oracle(dtype):
    o = O.OracleVAE(cfg, seeded, offsets=offsets())
    o.params = {k: v.detach().to(dtype).requires_grad_(True) for k, v in o.params.items()}
    o.masks = {k: v.to(dtype) for k, v in o.masks.items()}; o.pool_w = [x.to(dtype) for x in o.pool_w]; o.unpool_w = [x.to(dtype) for x in o.unpool_w]; o.offsets = o.offsets.to(dtype)
    x = batch[0].to(dtype).transpose(1, 2)
    lat, zs, acts = o.encode(x)
    return o, acts, zs
o64, a64, z64 = run_oracle(torch.float64)
o32, a32, z32 = run_oracle(torch.float32)
for i in range(4):
    Cn, T = a64[i].shape[1], a64[i].shape[2]
    print("A%d: hip-vs-f64 %.2e  ref32-vs-f64 %.2e" % (i, rel_err(bct(w["A%d" % i], Cn, T), a64[i].detach().numpy()), rel_err(a32[i].detach().numpy(), a64[i].detach().numpy())))
for i in (0, 3):
    k, d2 = z64[i].shape[1], z64[i].shape[2]
    print("ZD%d: hip-vs-f64 %.2e  ref32-vs-f64 %.2e" % (i, rel_err(bct(w["ZD%d" % i], k*d2, 1).reshape(B,k,d2), z64[i].detach().numpy()), rel_err(z32[i].detach().numpy(), z64[i].detach().numpy())))
# decoder from the SAME z (golden z list) in each precision
zl = [torch.from_numpy(g["p1|z%d" % i]) for i in range(4)]
d64 = o64.decode_feats([z.double() for z in zl]).detach().numpy()
d32 = o32.decode_feats(zl).detach().numpy()
c6, R, pos = tr.model._decode(zl)[:3]
mine = c6.cpu().numpy().reshape(B, 64, 144).transpose(0, 2, 1)
print("dec_out from fixed z: hip-vs-f64 %.2e ref32-vs-f64 %.2e" % (rel_err(mine, d64), rel_err(d32, d64)))
# ---- backward-side: d_dec_out and the last-layer bias gradient ----
def full(dtype):
    o = O.OracleVAE(cfg, seeded, offsets=offsets())
    o.params = {k: v.detach().to(dtype).requires_grad_(True) for k, v in o.params.items()}
    o.masks = {k: v.to(dtype) for k, v in o.masks.items()}; o.pool_w = [x.to(dtype) for x in o.pool_w]; o.unpool_w = [x.to(dtype) for x in o.unpool_w]; o.offsets = o.offsets.to(dtype)
    orig = torch.Tensor.float
    if dtype == torch.float64:
        torch.Tensor.float = lambda self: self.double()
    try:
        r = o.train_forward(tuple(t.to(dtype) for t in batch), 0, [e.to(dtype) for e in eps])
    finally:
        torch.Tensor.float = orig
    return o, r
o64, r64 = full(torch.float64); o32, r32 = full(torch.float32)
dd = bct(w["dD3"], 144, 64)
print("d_dec_out: hip-vs-f64 %.2e ref32-vs-f64 %.2e  max|d| %.3e" % (rel_err(dd, r64["dec_out"].grad.numpy()), rel_err(r32["dec_out"].grad.numpy(), r64["dec_out"].grad.numpy()), np.abs(dd).max()))
bias64 = o64.params["dec.layers.3.2.bias"].grad.numpy()
mine_b = dict(tr.model.named_parameters())["dec.layers.3.2.bias"].grad.cpu().numpy()
from_my_dd = dd.astype(np.float64).sum(axis=(0, 2))
print("bias3: hip %.2e | f64-sum of hip ddec %.2e | ref32 %.2e   max|b| %.3e" % (rel_err(mine_b, bias64), rel_err(from_my_dd, bias64), rel_err(o32.params["dec.layers.3.2.bias"].grad.numpy(), bias64), np.abs(bias64).max()))
dec_h = bct(w["D3"], 144, 64)
print("dec_out(train): hip-vs-f64 %.2e ref32 %.2e" % (rel_err(dec_h, r64["dec_out"].detach().numpy()), rel_err(r32["dec_out"].detach().numpy(), r64["dec_out"].detach().numpy())))
