"""GPU parity: ForwardKinematicsLayer on kinematic trees OTHER than SMPL-24 (fk_layer.py:13-34 takes any `parents` /
`positions`): the table-driven HIP kernels behind hmvae_fk_fwd / hmvae_fk_bwd against the oracle's restatement of
fk_layer.py:63-93 and torch autograd through it.  Tolerance: 1e-5 forward, 1e-4 relative backward (fp32)."""
import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu

TREES = {
    "chain5": [-1, 0, 1, 2, 3],
    "star7": [-1, 0, 0, 0, 0, 0, 0],
    "hand21": [-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19],
    "smpl22": [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19],       # SMPL without the hands
    "random40": None,
    "chain64": list(range(-1, 63)),
}


@pytest.fixture(scope="module")
def dev():
    from hm_vae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def tree(name):
    if TREES[name] is not None:
        return TREES[name]
    rng = np.random.default_rng(40)
    return [-1] + [int(rng.integers(0, j)) for j in range(1, 40)]


@pytest.mark.parametrize("name", list(TREES))
@pytest.mark.parametrize("rd", [6, 9])
def test_generic_tree_forward_backward(dev, name, rd):
    from hm_vae_amd import ops
    par = tree(name)
    J = len(par)
    n = 37                                              # not a multiple of the 16-frame tiles
    g = torch.Generator().manual_seed(J * 10 + rd)
    off = torch.randn(J, 3, generator=g) * 0.3
    raw = torch.randn(n, J, 6, generator=g)
    if rd == 6:
        x = raw.clone().requires_grad_(True)
        R_ref = O.rot6d_to_matrix(x)
    else:
        x = O.rot6d_to_matrix(raw).detach().clone().requires_grad_(True)      # valid rotations as (n, J, 3, 3)
        R_ref = x
    P_ref = O.fk_positions(R_ref, par, off)
    wP = torch.randn(n, J, 3, generator=g)
    wR = torch.randn(n, J, 3, 3, generator=g)
    ((P_ref * wP).sum() + (R_ref * wR).sum()).backward()
    xd = x.detach().reshape(n, J, rd).contiguous().to(dev)
    offd = off.to(dev)
    R = torch.full((n, J, 9), float("nan"), device=dev)
    P = torch.full((n, J, 3), float("nan"), device=dev)
    ops.fk_fwd(xd, rd, offd, 0, par, n, R, P)
    assert np.abs(P.cpu().numpy() - P_ref.detach().numpy()).max() < 1e-5 * max(1.0, float(P_ref.abs().max()))
    assert np.abs(R.cpu().numpy().reshape(n, J, 3, 3) - R_ref.detach().numpy()).max() < 1e-6
    d = torch.full((n, J, rd), float("nan"), device=dev)
    ops.fk_bwd(xd, rd, offd, 0, par, n, wR.reshape(n, J, 9).contiguous().to(dev), wP.to(dev), d)
    assert rel_err(d.cpu().numpy(), x.grad.reshape(n, J, rd).numpy()) < 1e-4
    # positions only (dR absent), per-frame offsets
    offs = (off[None] * torch.linspace(0.5, 1.5, n)[:, None, None]).contiguous()
    x2 = x.detach().clone().requires_grad_(True)
    R2 = O.rot6d_to_matrix(x2) if rd == 6 else x2
    P2_ref = torch.stack([O.fk_positions(R2[i:i + 1], par, offs[i])[0] for i in range(n)])
    (P2_ref * wP).sum().backward()
    P2 = torch.empty(n, J, 3, device=dev)
    ops.fk_fwd(xd, rd, offs.to(dev), J * 3, par, n, None, P2)
    assert np.abs(P2.cpu().numpy() - P2_ref.detach().numpy()).max() < 1e-5 * max(1.0, float(P2_ref.abs().max()))
    d2 = torch.full((n, J, rd), float("nan"), device=dev)
    ops.fk_bwd(xd, rd, offs.to(dev), J * 3, par, n, None, wP.to(dev), d2)
    assert rel_err(d2.cpu().numpy(), x2.grad.reshape(n, J, rd).numpy()) < 1e-4


def test_layer_surface_on_a_custom_skeleton(dev):
    """The module itself (reference signature: device, parents, positions) with autograd, on a 21-joint hand."""
    from hm_vae_amd.fk_layer import ForwardKinematicsLayer
    par = TREES["hand21"]
    g = torch.Generator().manual_seed(3)
    off = torch.randn(21, 3, generator=g) * 0.1
    layer = ForwardKinematicsLayer(dev, par, off.numpy())
    r6 = torch.randn(5, 21, 6, generator=g)
    x = r6.clone().requires_grad_(True)
    ref = O.fk_positions(O.rot6d_to_matrix(x), par, off)
    ref.pow(2).sum().backward()
    xd = r6.to(dev).requires_grad_(True)
    pos = layer(xd)
    assert rel_err(pos.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    pos.pow(2).sum().backward()
    assert rel_err(xd.grad.cpu().numpy(), x.grad.numpy()) < 1e-4


def test_invalid_trees_are_refused(dev):
    from hm_vae_amd import ops, _lib
    x = torch.zeros(2, 3, 9, device=dev)
    off = torch.zeros(3, 3, device=dev)
    P = torch.empty(2, 3, 3, device=dev)
    for bad in ([0, 0, 1], [-1, 2, 0], [-1, -1, 0]):           # root with a parent; child before its parent; two roots
        with pytest.raises(_lib.HmvaeError):
            ops.fk_fwd(x, 9, off, 0, bad, 2, None, P)
