"""Batch source for the training entrypoint.

The reference's `MotionSeqData` (utils_motion_vae.py:17-232) reads AMASS-derived `T x 579` npy files from paths on
the author's machine; that data is not part of this repository (out of scope for the hot path).
What IS part of the boundary is the 7-tuple batch layout (utils_motion_vae.py:187):
    (seq_rot_6d (B,T,144), seq_rot_mat (B,T,216), seq_rot_pos (B,T,72), seq_joint_pos (B,T,72),
     seq_linear_v (B,T,72), seq_angular_v (B,T,72), seq_root_v (B,T,3))
`SyntheticMotionLoader` yields that tuple with AMASS-shaped synthetic content (valid rotations from N(0,1) 6D draws,
SURVEY.md §8d), already resident on the device; `NpySequenceLoader` reads the reference's on-disk format if a
directory of such files is supplied.
"""
import glob
import os

import numpy as np
import torch


def load_mean_std():
    """(mean, std) rows of the 579-column statistics table, zero std replaced by 1 (utils_motion_vae.py:104-110)."""
    from . import topology
    ms = np.load(topology.resolve_asset("./utils/data/for_all_data_motion_model/all_amass_data_mean_std.npy"))
    ms = ms.astype(np.float32)
    ms[1, ms[1, :] == 0] = 1.0
    return ms[0], ms[1]


def _lift(raw):
    """6D -> rotation matrices on whatever device `raw` lives (HIP kernel on GPU)."""
    from . import my_tools
    return my_tools.rotation_matrix_from_ortho6d(raw)


class SyntheticMotionLoader:
    def __init__(self, batch_size, seq_len, device, n_batches=64, seed=0, with_positions=False):
        self.B, self.T, self.device, self.n, self.seed = batch_size, seq_len, device, n_batches, seed
        self.with_positions = with_positions
        self._fk = None

    def __len__(self):
        return self.n

    def make(self, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        B, T = self.B, self.T
        raw = torch.randn(B * T, 24, 6, generator=g).to(self.device)
        R = _lift(raw)
        r6 = torch.cat((R[..., :, 0], R[..., :, 1]), dim=-1).reshape(B, T, 144).contiguous()
        rmat = R.reshape(B, T, 216).contiguous()
        z72 = torch.zeros(B, T, 72, device=self.device)
        root_v = torch.randn(B, T, 3, generator=g).to(self.device)
        if not self.with_positions:
            return (r6, rmat, z72, z72, z72, z72, root_v)
        # TrajectoryModel reads the root-relative joint positions, raw [2] and standardised [3] (columns 360:432)
        if self._fk is None:
            from .fk_layer import ForwardKinematicsLayer
            self._fk = ForwardKinematicsLayer(device=self.device)
            mean, std = load_mean_std()
            self._mean = torch.from_numpy(mean[360:432]).to(self.device)
            self._std = torch.from_numpy(std[360:432]).to(self.device)
        pos = self._fk(R).reshape(B, T, 72)
        return (r6, rmat, pos.contiguous(), ((pos - self._mean) / self._std).contiguous(), z72, z72, root_v)

    def __iter__(self):
        for i in range(self.n):
            yield self.make(self.seed + i)


class NpySequenceLoader:
    """Reads `T x 579` float npy files (layout of utils/process_all_data_motion.py:155-158 in the reference):
    [0:144] 6D, [144:360] rotation matrices, [360:432] FK xyz, [432:504] lin-v, [504:576] lin-v, [576:579] root-v.
    As in MotionSeqData.__getitem__ (utils_motion_vae.py:124-187): rotations and seq_rot_pos stay raw, joint positions and
    the velocities are standardised with the shipped mean/std table."""

    def __init__(self, folder, batch_size, seq_len, seed=0):
        self.files = sorted(glob.glob(os.path.join(folder, "*.npy")))
        if not self.files:
            raise FileNotFoundError("no .npy sequences under %s" % folder)
        self.B, self.T = batch_size, seq_len
        self.rng = np.random.default_rng(seed)
        self.mean, self.std = load_mean_std()

    def __len__(self):
        return max(1, len(self.files) // self.B)

    def __iter__(self):
        order = self.rng.permutation(len(self.files))
        for s in range(0, len(order) - self.B + 1, self.B):
            clips = []
            for idx in order[s:s + self.B]:
                seq = np.load(self.files[idx], allow_pickle=False)
                if seq.shape[0] < self.T:
                    continue
                t0 = int(self.rng.integers(0, seq.shape[0] - self.T + 1))
                clips.append(seq[t0:t0 + self.T])
            if len(clips) < self.B:
                continue
            raw = np.stack(clips).astype(np.float32)
            x = torch.from_numpy(raw)
            n = torch.from_numpy((raw - self.mean) / self.std)
            yield (x[:, :, 0:144], x[:, :, 144:360], x[:, :, 360:432], n[:, :, 360:432], n[:, :, 432:504],
                   n[:, :, 504:576], n[:, :, 576:579])


def get_train_loaders_all_data_seq(config, device=None, data_folder=None, n_batches=64):
    """Same name as the reference factory (utils_motion_vae.py:194-232): returns (train, val, test) loaders."""
    B, T = config["batch_size"], config["train_seq_len"]
    if data_folder:
        mk = lambda seed: NpySequenceLoader(data_folder, B, T, seed)
        return mk(0), mk(1), NpySequenceLoader(data_folder, 1, T, 2)
    dev = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    wp = config.get("model_name") == "TrajectoryModel"
    return (SyntheticMotionLoader(B, T, dev, n_batches, 0, wp), SyntheticMotionLoader(B, T, dev, 8, 10_000, wp),
            SyntheticMotionLoader(1, T, dev, 4, 20_000, wp))
