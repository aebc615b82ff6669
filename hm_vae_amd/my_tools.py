"""rotation helpers with the reference's names (my_tools.py:6-39), on HIP."""
import torch

from . import functional as Fn
from .topology import SMPL24_PARENTS


def normalize_vector(v):
    """my_tools.py:6-8 (plumbing-sized helper; torch op on the caller's device)."""
    return torch.nn.functional.normalize(v, dim=-1, eps=1e-6)


def cross_product(u, v):
    """my_tools.py:10-17"""
    return torch.stack((u[..., 1] * v[..., 2] - u[..., 2] * v[..., 1],
                        u[..., 2] * v[..., 0] - u[..., 0] * v[..., 2],
                        u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]), dim=-1)


def rotation_matrix_from_ortho6d(poses):
    """poses (..., J, 6) -> (..., J, 3, 3), columns (x, y, z); my_tools.py:19-39.  Runs the fused HIP lift; the
    24-joint case shares the FK kernel (positions are simply not requested)."""
    shape = poses.shape
    if shape[-1] != 6:
        raise ValueError("expected (..., 6)")
    flat = poses.reshape(-1, 6)
    n = flat.shape[0]
    pad = (-n) % 24
    if pad:
        flat = torch.cat((flat, flat.new_zeros(pad, 6)), dim=0)
    frames = flat.reshape(-1, 24, 6)
    off = torch.zeros(24, 3, device=poses.device)
    R, _ = Fn.FkFn.apply(frames, off, SMPL24_PARENTS, True)
    R = R.reshape(-1, 3, 3)[:n]
    return R.reshape(*shape[:-1], 3, 3)
