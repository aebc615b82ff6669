"""Full-size runs (BASELINE.json configs 2 and 4) checked through size-independent properties: the CPU oracle does not
finish these in seconds, the properties do not need it."""
import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import load_cfg, offsets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from hm_vae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


EPS32 = float(np.finfo(np.float32).eps)        # 1.19e-7


def _lift_conditioning(c6):
    """sin of the angle between the two 6D vectors of every joint: the reference's Gram-Schmidt (my_tools.py:19-39)
    computes z = normalize(x x b) in fp32, so its own relative error in z -- and with it every orthonormality / parity
    bound on R -- scales like eps / sin(angle(a, b)).  Evaluated in float64 from the very inputs the kernel read."""
    a, b = c6[..., 0:3].double(), c6[..., 3:6].double()
    x = a / a.norm(dim=-1, keepdim=True)
    return torch.linalg.cross(x, b).norm(dim=-1) / b.norm(dim=-1)


def test_decode_tail_config4_properties(dev):
    """Config 4: 4096 sequences x 64 frames = 262 144 frames through decode_tail_fwd (6D -> R -> FK).

    Seeded on the HOST (the same 37.7 M normals on every box).  What is asserted is what the reference guarantees:
    (1) parity with the oracle (my_tools.py:19-39, fk_layer.py:63-93) on the first 64 sequences and on the 4 096
    worst-conditioned joints of the whole batch, with the bound 2e-6 + 8 eps / sin(angle(a, b)) per joint -- for a
    nearly parallel to b the reference's own x x b cancels, which the kernel reproduces but cannot undo;
    (2) orthonormality and det = +1 of EVERY rotation with the same conditioning-scaled bound;
    (3) rigid bones, bounded by the measured orthonormality error of the ancestors' rotations."""
    from hm_vae_amd import ops
    B, T = 4096, 64
    bp = ops.pad_batch(B)
    off = torch.from_numpy(offsets()).to(dev)
    gen = torch.Generator().manual_seed(20260401)
    dec = torch.randn(144 * T, bp, generator=gen).to(dev)
    c6 = torch.empty(B, T, 24, 6, device=dev)
    R = torch.empty(B, T, 24, 3, 3, device=dev)
    P = torch.empty(B, T, 24, 3, device=dev)
    ops.decode_tail_fwd(dec, off, T, B, bp, c6, R, P)
    # the 6D output is the decoder output, transposed: exact
    back = torch.empty(144 * T, bp, device=dev)
    ops.aos_to_soa(c6.reshape(B, T, 144), back, B, T, 144, bp)
    assert torch.equal(back[:, :B], dec[:, :B])
    par = O.SMPL24_PARENTS
    sin = _lift_conditioning(c6).reshape(-1)                     # (B*T*24,)
    bound = (2e-6 + 8.0 * EPS32 / sin)                           # per joint
    Rf = R.reshape(-1, 3, 3)
    # (1a) oracle parity, first 64 sequences (98 304 joints), R from 6D and positions from the HIP rotations
    n0 = 64 * T
    R_o = O.rot6d_to_matrix(c6.reshape(-1, 24, 6)[:n0].cpu())
    dR = (Rf[:n0 * 24].cpu() - R_o.reshape(-1, 3, 3)).abs().amax(dim=(1, 2)).double()
    assert bool((dR <= bound[:n0 * 24].cpu()).all()), float((dR / bound[:n0 * 24].cpu()).max())
    well = sin[:n0 * 24].cpu() > 0.1
    assert float(dR[well].max()) < 1e-5                          # the plain statement for every ordinary joint
    P_o = O.fk_positions(R.reshape(-1, 24, 3, 3)[:n0].cpu(), par, off.cpu())
    assert float((P.reshape(-1, 24, 3)[:n0].cpu() - P_o).abs().max()) < 1e-5
    # (1b) the 4 096 worst-conditioned joints of all 6.3 M
    worst = torch.topk(sin, 4096, largest=False).indices
    R_w = O.rot6d_to_matrix(c6.reshape(-1, 6)[worst].cpu())
    dRw = (Rf[worst].cpu() - R_w).abs().amax(dim=(1, 2)).double()
    assert bool((dRw <= bound[worst].cpu()).all()), float((dRw / bound[worst].cpu()).max())
    # (2) every R is a rotation: R^T R = I, det = +1 (my_tools.py:19-39 builds an orthonormal right-handed frame)
    eye = torch.eye(3, device=dev)
    ortho = (Rf.transpose(1, 2) @ Rf - eye).abs().amax(dim=(1, 2)).double()
    assert bool((ortho <= bound).all()), float((ortho / bound).max())
    assert float(ortho[sin > 0.1].max()) < 1e-5
    det = (torch.linalg.det(Rf.double()) - 1.0).abs()
    assert bool((det <= 3.0 * bound).all()), float((det / bound).max())
    # first column = normalised first 6D vector
    a = c6.reshape(-1, 6)[:, 0:3]
    assert float((Rf[:, :, 0] - a / a.norm(dim=1, keepdim=True)).abs().max()) < 2e-6
    # (3) FK: the root sits at its offset; bone j is G_parent . off_j, so its length deviates from |off_j| by at most
    # |off_j| x (sum of the ancestors' ||R^T R - I||, first order) -- bounded with what was just measured; and the
    # frame-major FK kernel on the same rotations gives the same positions
    Pf = P.reshape(-1, 24, 3)
    assert float((Pf[:, 0] - off[0]).abs().max()) == 0.0
    e = 3.0 * ortho.reshape(-1, 24)                              # >= spectral norm of R^T R - I
    chain = [None] * 24
    chain[0] = e[:, 0]
    for j in range(1, 24):
        chain[j] = chain[par[j]] + e[:, j]
    for j in range(1, 24):
        bone = (Pf[:, j] - Pf[:, par[j]]).double().norm(dim=1)
        lim = float(off[j].norm()) * chain[par[j]] + 2e-6
        assert bool(((bone - float(off[j].double().norm())).abs() <= lim).all()), j
    P2 = torch.empty(B * T, 24, 3, device=dev)
    ops.fk_fwd(R.reshape(B * T, 24, 9), 9, off, 0, par, B * T, None, P2)
    assert float((P2 - Pf).abs().max()) < 1e-5
    # and without the 6D copy the same R / P
    R3, P3 = torch.empty_like(R), torch.empty_like(P)
    ops.decode_tail_fwd(dec, off, T, B, bp, None, R3, P3)
    assert torch.equal(R3, R) and torch.equal(P3, P)


def test_training_step_config2_properties(dev):
    """Config 2: len64 config at batch 256, phase 2: finite losses that add up, exact zeros outside the neighbour masks,
    the reference's None-gradient set, bit-identical repeat, and a loss that goes down over a few steps."""
    from hm_vae_amd.trainer_motion_vae import Trainer
    cfg = load_cfg("len64")
    B, T = 256, cfg["train_seq_len"]
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.cuda()
    plan = O.Plan(cfg)
    batch = tuple(t.to(dev) for t in O.synthetic_batch(B, T, seed=2))
    eps = [e.to(dev) for e in O.level_eps(plan, B, seed=3)]
    it = cfg["iteration_interval"]
    tr.model.eps_override = eps
    out = tr.model(batch, cfg, it)
    v = [float(x) for x in out[:5]]
    assert all(np.isfinite(v))
    total = cfg["rec_6d_w"] * v[2] + cfg["rec_rot_w"] * v[3] + cfg["rec_pose_w"] * v[4] + v[1]
    assert abs(total - v[0]) < 1e-5 * abs(v[0])
    grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in tr.model.named_parameters() if p.requires_grad}
    none = sorted(n for n, g in grads.items() if g is None)
    assert len(none) == 8 and all("latent" in n for n in none)          # the dead middle heads, phase 2
    for k, m in O.oracle_masks(plan).items():
        g = grads[k.replace("mask", "weight")]
        assert bool(torch.isfinite(g).all()) and float((g * (1 - m.to(dev))).abs().max()) == 0.0
    # same inputs, same kernels -> same bits
    out2 = tr.model(batch, cfg, it)
    assert [float(x) for x in out2[:5]] == v
    for n, p in tr.model.named_parameters():
        if p.requires_grad and p.grad is not None:
            assert torch.equal(p.grad, grads[n]), n
    # a few optimiser steps on the same batch reduce the loss
    first = tr.gen_update(batch, cfg, it, False)[0]
    for s in range(1, 6):
        last = tr.gen_update(batch, cfg, it + s, False)[0]
    assert last < first
