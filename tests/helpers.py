"""Shared helpers for the parity tests (oracle side). Test-only."""
import os
import json

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CONFIGS = os.path.join(ROOT, "configs")

CFG_FILES = {"len64": "len64_no_aug_hm_vae.yaml", "len8": "len8_data_aug_hm_vae.yaml"}


def load_cfg(tag):
    """len16: the len8 config at train_seq_len = 16 (the T == 16 branch of the reference's stride / upsampling ladders)."""
    with open(os.path.join(CONFIGS, CFG_FILES["len8" if tag == "len16" else tag])) as f:
        cfg = yaml.safe_load(f)
    if tag == "len16":
        cfg["train_seq_len"] = 16
    return cfg


def load_model_golden(tag):
    g = np.load(os.path.join(GOLDEN, "model_%s.npz" % tag))
    meta = json.load(open(os.path.join(GOLDEN, "model_meta.json")))[tag]
    return g, meta


def golden_batch(g):
    r6 = torch.from_numpy(g["in_r6"])
    rmat = torch.from_numpy(g["in_rmat"])
    b, T, _ = r6.shape
    z72 = torch.zeros(b, T, 72)
    return (r6, rmat, z72, z72.clone(), z72.clone(), z72.clone(), torch.from_numpy(g["in_root_v"]))


def golden_eps(g):
    return [torch.from_numpy(g["eps%d" % i]) for i in range(4)]


def offsets():
    return np.load(os.path.join(GOLDEN, "skeleton_offsets.npy"))


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def scaled_err(a, b, tol, rms_mult=4.0):
    """Worst |a - b| / (tol * |b| + rms_mult * tol * rms(b)) over the tensor: <= 1 means every entry sits within `tol` of
    its own magnitude plus rms_mult * tol of the tensor's RMS.  rel_err above normalises by the LARGEST entry, so a wrong
    block of small entries in a wide-range gradient could hide behind it; this bound is relative to the typical entry.
    rms_mult = 4: an entry's fp32 error follows the size of its TERMS, not of their sum -- measured at B = 256 the
    reference's own fp32 evaluation sits at 1.1 of the rms_mult = 1 bound against float64 (enc.layers.0.0.weight, sums over
    64 x 256 terms), the HIP step at 1.7-2.5 (tools/diag_config2.py)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    bound = tol * np.abs(b) + rms_mult * tol * np.sqrt(np.mean(b * b))
    diff = np.abs(a - b)
    if not diff.any():
        return 0.0
    return float((diff / np.maximum(bound, 1e-300)).max())


def load_traj_cfg(T=None):
    with open(os.path.join(CONFIGS, "trajectory_model.yaml")) as f:
        cfg = yaml.safe_load(f)
    if T is not None:
        cfg["train_seq_len"] = T
    return cfg


def load_traj_golden(tag):
    g = np.load(os.path.join(GOLDEN, "traj_%s.npz" % tag))
    meta = json.load(open(os.path.join(GOLDEN, "traj_meta.json")))[tag]
    return g, meta


def traj_batch(g):
    jp = torch.from_numpy(g["in_joint_pos"])
    b, T, _ = jp.shape
    z = lambda n: torch.zeros(b, T, n)
    return (z(144), z(216), torch.from_numpy(g["in_rot_pos"]), jp, z(72), z(72), torch.from_numpy(g["in_root_v"]))
