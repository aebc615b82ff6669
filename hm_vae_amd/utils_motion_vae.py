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


FPS_STRIDES = (1, 2, 3, 4, 5, 6, 8, 10, 12)          # change_fps, utils_motion_vae.py:71
CLIP_DTYPE = np.dtype([("row0", "<i8"), ("stride", "<i4"), ("reserved", "<i4"), ("rot", "<f4", (9,)), ("pad_", "<f4", (3,))])


def rand_rotation_matrix(randnums, deflection=1.0):
    """utils_motion_vae.py:16-57 — three uniforms in [0, 1] -> rotation matrix (float64)."""
    theta, phi, z = (float(v) for v in randnums)
    theta, phi, z = theta * 2.0 * deflection * np.pi, phi * 2.0 * np.pi, z * 2.0 * deflection
    r = np.sqrt(z)
    V = (np.sin(phi) * r, np.cos(phi) * r, np.sqrt(2.0 - z))
    st, ct = np.sin(theta), np.cos(theta)
    return (np.outer(V, V) - np.eye(3)).dot(np.array(((ct, st, 0), (-st, ct, 0), (0, 0, 1))))


class DeviceMotionDataset:
    """The `T x 579` sequences of the reference's on-disk format, packed back to back in ONE device tensor.

    The reference's loader (`MotionSeqData` + DataLoader(workers=1), utils_motion_vae.py:83-232) reads one npy file per
    item on the host, standardises it with numpy and ships 7 small tensors per item; here the dataset is read once,
    lives in HBM (AMASS at 30 fps is ~10 GB of 579-column rows), and a batch is ONE kernel launch
    (hmvae_motion_batch) driven by a B x 64-byte descriptor table.  `source`: a folder of npy files (optionally with
    the reference's {"0": "name.npy", ...} id json) or a list of arrays."""

    def __init__(self, source, device, ids_json=None):
        from . import _lib
        _lib.load()
        if isinstance(source, (list, tuple)):
            arrays = [np.asarray(a, dtype=np.float32) for a in source]
        else:
            if ids_json:
                import json
                ids = json.load(open(ids_json))
                names = [ids[str(i)] for i in sorted(int(k) for k in ids)]
            else:
                names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(source, "*.npy")))
            if not names:
                raise FileNotFoundError("no .npy sequences under %s" % source)
            arrays = [np.load(os.path.join(source, n), allow_pickle=False).astype(np.float32) for n in names]
        for a in arrays:
            if a.ndim != 2 or a.shape[1] != 579:
                raise ValueError("sequences must be T x 579 (got %r)" % (a.shape,))
        self.seq_len = np.array([a.shape[0] for a in arrays], dtype=np.int64)
        self.seq_start = np.concatenate(([0], np.cumsum(self.seq_len)[:-1])).astype(np.int64)
        self.n_rows = int(self.seq_len.sum())
        self.device = torch.device(device)
        self.frames = torch.from_numpy(np.concatenate(arrays, axis=0)).to(self.device)
        from . import topology
        mean, std = np.load(topology.resolve_asset("./utils/data/for_all_data_motion_model/all_amass_data_mean_std.npy"))
        std = std.copy()
        std[std == 0] = 1.0                   # utils_motion_vae.py:100-101
        self.mean = torch.from_numpy(mean.astype(np.float64)).to(self.device)
        self.std = torch.from_numpy(std.astype(np.float64)).to(self.device)

    def __len__(self):
        return len(self.seq_len)

    def assemble(self, clips, T, rot_flag):
        """clips: numpy structured array (CLIP_DTYPE), one row per clip -> the 7-tuple of (B, T, n) device tensors."""
        from . import _lib
        B = len(clips)
        last = clips["row0"] + (T - 1) * clips["stride"].astype(np.int64)
        if (clips["row0"] < 0).any() or (last >= self.n_rows).any() or (clips["stride"] < 1).any():
            raise ValueError("clip descriptor outside the packed dataset")
        tab = torch.from_numpy(clips.view(np.uint8).reshape(B, 64)).to(self.device, non_blocking=True)
        dev = self.device
        outs = [torch.empty(B, T, n, device=dev) for n in (144, 216, 72, 72, 72, 72, 3)]
        _lib.check(_lib.load().hmvae_motion_batch(_lib.ptr(self.frames), self.n_rows, tab.data_ptr(), self.mean.data_ptr(),
                                                  self.std.data_ptr(), B, T, 1 if rot_flag else 0,
                                                  *[_lib.ptr(o) for o in outs], _lib.stream()), "motion_batch")
        return tuple(outs)


class DeviceMotionLoader:
    """Batches of the reference's 7-tuple from a DeviceMotionDataset, with MotionSeqData's sampling rules
    (utils_motion_vae.py:124-187): optional fps augmentation (frame stride from FPS_STRIDES, up to 10 draws, :66-83), a
    uniformly random crop of train_seq_len frames, sequences that are too short replaced by a random other one, optional
    random root rotation.  Epochs are shuffled permutations with the incomplete last batch dropped (DataLoader(shuffle,
    drop_last), :216-218)."""

    def __init__(self, dataset, batch_size, seq_len, fps_aug=False, root_rot=False, shuffle=True, seed=0):
        self.ds, self.B, self.T = dataset, int(batch_size), int(seq_len)
        self.fps_aug, self.root_rot, self.shuffle = bool(fps_aug), bool(root_rot), bool(shuffle)
        self.rng = np.random.default_rng(seed)
        if int(dataset.seq_len.max()) < self.T:
            raise ValueError("no sequence has %d frames" % self.T)

    def __len__(self):
        return len(self.ds) // self.B

    def sample_clip(self, idx):
        """(sequence, stride, crop start, rotation or None) for dataset item `idx`."""
        ds, T, rng = self.ds, self.T, self.rng
        while True:
            n, stride = int(ds.seq_len[idx]), 1
            if self.fps_aug:
                for _ in range(10):
                    f = int(FPS_STRIDES[rng.integers(len(FPS_STRIDES))])
                    if len(range(0, n, f)) >= T:
                        stride = f
                        break
            n2 = len(range(0, n, stride))
            if T <= n2:
                crop = int(rng.integers(0, n2 - T + 1))
                rot = rand_rotation_matrix(rng.uniform(size=3)) if self.root_rot else None
                return idx, stride, crop, rot
            idx = int(rng.integers(len(ds)))

    def sample_batch(self, idx):
        """sample_clip for a whole batch with array operations (the per-clip Python loop costs ~1.4 ms per 256 clips, more
        than the GPU needs for the training step).  Returns (seq, stride, crop, rot (B,3,3) or None)."""
        ds, T, rng = self.ds, self.T, self.rng
        seq = np.asarray(idx, dtype=np.int64).copy()
        B = len(seq)
        stride, crop = np.ones(B, dtype=np.int64), np.zeros(B, dtype=np.int64)
        todo = np.arange(B)
        fps = np.asarray(FPS_STRIDES, dtype=np.int64)
        while len(todo):
            n = ds.seq_len[seq[todo]]
            st = np.ones(len(todo), dtype=np.int64)
            if self.fps_aug:
                draws = fps[rng.integers(len(fps), size=(len(todo), 10))]
                ok = -(-n[:, None] // draws) >= T                      # frames left at that stride: ceil(n / f)
                first = ok.argmax(axis=1)
                st = np.where(ok.any(axis=1), draws[np.arange(len(todo)), first], 1)
            n2 = -(-n // st)
            good = n2 >= T
            g = todo[good]
            stride[g] = st[good]
            crop[g] = np.floor(rng.random(len(g)) * (n2[good] - T + 1)).astype(np.int64)
            bad = todo[~good]
            seq[bad] = rng.integers(len(ds), size=len(bad))
            todo = bad
        rot = None
        if self.root_rot:
            u = rng.uniform(size=(B, 3))
            theta, phi, z = u[:, 0] * 2.0 * np.pi, u[:, 1] * 2.0 * np.pi, u[:, 2] * 2.0
            r = np.sqrt(z)
            V = np.stack((np.sin(phi) * r, np.cos(phi) * r, np.sqrt(2.0 - z)), axis=1)
            st_, ct = np.sin(theta), np.cos(theta)
            Rz = np.zeros((B, 3, 3))
            Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = ct, st_, -st_, ct, 1.0
            rot = np.matmul(V[:, :, None] * V[:, None, :] - np.eye(3)[None], Rz)
        return seq, stride, crop, rot

    def describe(self, choices):
        """choices: a list of (seq, stride, crop, rot) or the 4-tuple of arrays of sample_batch."""
        if isinstance(choices, tuple) and len(choices) == 4 and isinstance(choices[0], np.ndarray):
            seq, stride, crop, rot = choices
        else:
            seq = np.array([c[0] for c in choices], dtype=np.int64)
            stride = np.array([c[1] for c in choices], dtype=np.int64)
            crop = np.array([c[2] for c in choices], dtype=np.int64)
            rot = None if choices[0][3] is None else np.stack([np.asarray(c[3]) for c in choices])
        clips = np.zeros(len(seq), dtype=CLIP_DTYPE)
        clips["row0"] = self.ds.seq_start[seq] + crop * stride
        clips["stride"] = stride
        clips["rot"] = (np.broadcast_to(np.eye(3), (len(seq), 3, 3)) if rot is None else rot).astype(np.float32).reshape(-1, 9)
        return clips

    def __iter__(self):
        order = self.rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        for s in range(0, len(order) - self.B + 1, self.B):
            yield self.ds.assemble(self.describe(self.sample_batch(order[s:s + self.B])), self.T, self.root_rot)


TRAIN_IDS = "train_all_amass_motion_data.json"        # utils_motion_vae.py:205-206
VAL_IDS = "val_all_amass_motion_data.json"


def _split_files(data_folder, config):
    """(train ids json, val ids json) the way the reference factory finds them (utils_motion_vae.py:196-206): under
    <data_root>/for_all_data_motion_model/, here also accepted next to the npy files or named in the config
    ('train_ids_json' / 'val_ids_json').  (None, None) when there is no split."""
    cands = [(config.get("train_ids_json"), config.get("val_ids_json"))]
    for d in (data_folder, os.path.join(data_folder, "for_all_data_motion_model"),
              os.path.join(os.path.dirname(os.path.abspath(data_folder)), "for_all_data_motion_model")):
        cands.append((os.path.join(d, TRAIN_IDS), os.path.join(d, VAL_IDS)))
    for tr, va in cands:
        if tr and va and os.path.exists(tr) and os.path.exists(va):
            return tr, va
    return None, None


def get_train_loaders_all_data_seq(config, device=None, data_folder=None, n_batches=64):
    """Same name as the reference factory (utils_motion_vae.py:194-232): returns (train, val, test) loaders.

    With a data folder: train draws from the sequences of train_all_amass_motion_data.json, val and test from those of
    val_all_amass_motion_data.json (two device-resident datasets), exactly the reference's split.  If the id files are not
    there the folder is ONE pool: that is said loudly and the val / test loaders are None -- the training script then
    skips validation instead of printing training-set losses under a 'Val' label."""
    B, T = config["batch_size"], config["train_seq_len"]
    if data_folder:
        dev = device or torch.device("cuda")
        from . import ddp
        fps, rot = bool(config.get("fps_aug_flag", False)), bool(config.get("random_root_rot_flag", False))
        r = 1000 * ddp.rank()            # every rank draws its own clips (one process per GPU, batch_size sequences each)
        tr_json, va_json = _split_files(data_folder, config)
        if tr_json is None:
            import warnings
            warnings.warn("hm_vae_amd: no %s / %s for %s: every sequence is used for training and validation is OFF"
                          % (TRAIN_IDS, VAL_IDS, data_folder))
            ds = DeviceMotionDataset(data_folder, dev)
            return (DeviceMotionLoader(ds, B, T, fps, rot, True, r), None, None)
        ds_tr = DeviceMotionDataset(data_folder, dev, ids_json=tr_json)
        ds_va = DeviceMotionDataset(data_folder, dev, ids_json=va_json)
        return (DeviceMotionLoader(ds_tr, B, T, fps, rot, True, r), DeviceMotionLoader(ds_va, B, T, fps, rot, False, r + 1),
                DeviceMotionLoader(ds_va, B, T, fps, rot, True, r + 2))
    dev = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    wp = config.get("model_name") == "TrajectoryModel"
    from . import ddp
    r = 100_000 * ddp.rank()
    return (SyntheticMotionLoader(B, T, dev, n_batches, r, wp), SyntheticMotionLoader(B, T, dev, 8, r + 10_000, wp),
            SyntheticMotionLoader(1, T, dev, 4, r + 20_000, wp))
