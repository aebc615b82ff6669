"""The oracle's restatement of the test-time optimisation losses tied to the pieces the reference's
golden vectors pin: with an all-ones mask and the latents of a training forward pass, its reconstruction terms ARE the
training losses (seq_two_hier_sa_vae.py:717-735 reduces to l2_criterion), and a zero mask removes them."""
import numpy as np
import torch

from oracle import hmvae_oracle as O
from helpers import load_cfg, load_model_golden, golden_batch, golden_eps, offsets, rel_err


def test_refine_forward_reduces_to_training_losses():
    cfg = load_cfg("len8")
    g, _ = load_model_golden("len8")
    plan = O.Plan(cfg)
    orc = O.OracleVAE(cfg, O.name_seeded_params(plan), offsets=offsets())
    data = golden_batch(g)
    out = orc.train_forward(data, cfg["iteration_interval"], golden_eps(g), backward=False)
    z0, zd = out["z_list"][0].detach(), out["z_list"][-1].detach()
    B, T = data[0].shape[0], data[0].shape[1]
    target = data[1].reshape(B, T, 24, 3, 3)
    hp = dict(cfg, reg_w=0.5)
    orc.hp = hp
    ones = torch.ones(B, T, 24)
    o = orc.refine_forward(z0, zd, target, ones, backward=False)
    for k in ("rec_rot", "rec_pose"):
        assert rel_err(float(o[k]), float(out[k])) < 1e-6, k
    # the 6D target of the routine is rebuilt from R's first two columns (:1279-1280): equal to the data's 6D input
    assert rel_err(float(o["rec_6d"]), float(out["rec_6d"])) < 1e-5
    reg = float((z0 ** 2).mean() + (zd ** 2).mean())
    want = cfg["rec_6d_w"] * float(o["rec_6d"]) + cfg["rec_rot_w"] * float(o["rec_rot"]) + \
        cfg["rec_pose_w"] * float(o["rec_pose"]) + 0.5 * reg
    assert abs(float(o["total"]) - want) < 1e-6 * abs(want)
    zero = orc.refine_forward(z0, zd, target, torch.zeros(B, T, 24), backward=False)
    assert float(zero["rec_6d"]) == 0.0 and float(zero["rec_rot"]) == 0.0 and float(zero["rec_pose"]) == 0.0
    # half the frames observed: the mean still divides by the full element count
    half = ones.clone()
    half[:, ::2] = 0
    h = orc.refine_forward(z0, zd, target, half, backward=False)
    c6 = o["c6"]
    g6 = torch.stack((target[..., 0], target[..., 1]), dim=-2).reshape(B, T, 24, 6)
    manual = (((c6 - g6) ** 2) * half[..., None]).sum() / c6.numel()
    assert abs(float(h["rec_6d"]) - float(manual)) < 1e-7


def test_refine_forward_gradients_finite_difference():
    cfg = load_cfg("len8")
    plan = O.Plan(cfg)
    seeded = O.name_seeded_params(plan)
    orc = O.OracleVAE(dict(cfg, reg_w=0.1, reg_w_decoder=7.0), seeded, offsets=offsets())
    B, T = 2, cfg["train_seq_len"]
    gen = torch.Generator().manual_seed(0)
    target = O.rot6d_to_matrix(torch.randn(B * T * 24, 6, generator=gen)).reshape(B, T, 24, 3, 3)
    mask = (torch.rand(B, T, 24, generator=gen) < 0.5).float()
    z0 = torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=gen).double()
    zd = torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=gen).double()
    name = "dec.latent_dec_layers.0.bias"
    anchor = {name: seeded[name] + 0.1}
    z0g, zdg = z0.float().requires_grad_(True), zd.float().requires_grad_(True)
    orc.refine_forward(z0g, zdg, target, mask, anchor=anchor)
    ga = float(zdg.grad[1, 2, 3])
    gb = float(orc.params[name].grad[5])
    h = 1e-2
    def total(dz, db):
        z = zd.float().clone()
        z[1, 2, 3] += dz
        with torch.no_grad():
            orc.params[name][5] += db
        v = float(orc.refine_forward(z0.float(), z, target, mask, anchor=anchor, backward=False)["total"])
        with torch.no_grad():
            orc.params[name][5] -= db
        return v
    fa = (total(h, 0) - total(-h, 0)) / (2 * h)
    fb = (total(0, h) - total(0, -h)) / (2 * h)
    assert abs(fa - ga) < 0.05 * max(abs(ga), 1e-3), (fa, ga)
    assert abs(fb - gb) < 0.05 * max(abs(gb), 1e-3), (fb, gb)
