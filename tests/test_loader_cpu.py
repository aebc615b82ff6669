"""The batch source either side of the hot path: the reference's 7-tuple layout (utils_motion_vae.py:124-187) from
`T x 579` npy sequences.  Rotations and seq_rot_pos stay raw; joint positions / velocities are standardised with the
shipped mean/std table (zero std -> 1, :104-110)."""
import os

import numpy as np
import torch

from helpers import ROOT


def test_npy_loader_slices_and_standardisation(tmp_path):
    from hm_vae_amd.utils_motion_vae import NpySequenceLoader, load_mean_std, get_train_loaders_all_data_seq
    rng = np.random.default_rng(0)
    seqs = []
    for i in range(5):
        a = rng.standard_normal((20 + i, 579)).astype(np.float32)
        np.save(os.path.join(str(tmp_path), "s%d.npy" % i), a)
        seqs.append(a)
    np.save(os.path.join(str(tmp_path), "short.npy"), rng.standard_normal((4, 579)).astype(np.float32))  # < T: skipped
    mean, std = load_mean_std()
    raw = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "all_amass_data_mean_std.npy"))
    assert mean.shape == (579,) and np.all(std[raw[1] == 0] == 1.0) and np.all(std != 0)
    T, B = 16, 2
    batches = list(NpySequenceLoader(str(tmp_path), B, T, seed=3))
    assert len(batches) >= 1
    for tup in batches:
        assert len(tup) == 7
        assert [tuple(t.shape) for t in tup] == [(B, T, 144), (B, T, 216), (B, T, 72), (B, T, 72), (B, T, 72), (B, T, 72),
                                                 (B, T, 3)]
        for b in range(B):
            # find the clip in the sources through its raw 6D block
            hit = None
            for a in seqs:
                for t0 in range(a.shape[0] - T + 1):
                    if np.array_equal(a[t0:t0 + T, 0:144], tup[0][b].numpy()):
                        hit = a[t0:t0 + T]
            assert hit is not None
            n = (hit - mean) / std
            assert np.array_equal(tup[1][b].numpy(), hit[:, 144:360])
            assert np.array_equal(tup[2][b].numpy(), hit[:, 360:432])           # seq_rot_pos: raw
            assert np.allclose(tup[3][b].numpy(), n[:, 360:432], atol=1e-6)     # seq_joint_pos: standardised
            assert np.allclose(tup[4][b].numpy(), n[:, 432:504], atol=1e-6)
            assert np.allclose(tup[5][b].numpy(), n[:, 504:576], atol=1e-6)
            assert np.allclose(tup[6][b].numpy(), n[:, 576:579], atol=1e-6)
    tr, va, te = get_train_loaders_all_data_seq({"batch_size": B, "train_seq_len": T}, None, str(tmp_path))
    assert next(iter(te))[0].shape[0] == 1


def test_synthetic_loader_layout_cpu_only_for_vae():
    from hm_vae_amd.utils_motion_vae import get_train_loaders_all_data_seq
    tr, va, te = get_train_loaders_all_data_seq({"batch_size": 3, "train_seq_len": 8, "model_name": "TrajectoryModel"},
                                                torch.device("cpu"))
    assert tr.with_positions and len(tr) == 64 and te.B == 1
