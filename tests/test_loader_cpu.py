"""The batch source either side of the hot path: the reference's 7-tuple layout (utils_motion_vae.py:124-187) from
`T x 579` npy sequences.  Rotations and seq_rot_pos stay raw; joint positions / velocities are standardised with the
shipped mean/std table (zero std -> 1, :104-110)."""
import os

import numpy as np
import torch

from helpers import ROOT


def test_npy_loader_slices_and_standardisation(tmp_path):
    from hm_vae_amd.utils_motion_vae import NpySequenceLoader, load_mean_std
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


def test_synthetic_loader_layout_cpu_only_for_vae():
    from hm_vae_amd.utils_motion_vae import get_train_loaders_all_data_seq
    tr, va, te = get_train_loaders_all_data_seq({"batch_size": 3, "train_seq_len": 8, "model_name": "TrajectoryModel"},
                                                torch.device("cpu"))
    assert tr.with_positions and len(tr) == 64 and te.B == 1


def test_device_loader_host_sampler_follows_the_reference_rules():
    """DeviceMotionLoader.sample_clip / describe (host side of hmvae_motion_batch): MotionSeqData's control flow
    (utils_motion_vae.py:66-83, 124-141) -- strides from the fps list that leave >= T frames, crops inside the strided
    sequence, too-short sequences replaced -- and descriptors that address the packed table."""
    import pytest
    from oracle import hmvae_oracle as O
    from hm_vae_amd import _lib
    from hm_vae_amd.utils_motion_vae import DeviceMotionDataset, DeviceMotionLoader, FPS_STRIDES, rand_rotation_matrix
    seqs = O.synthetic_motion_files()
    ds = DeviceMotionDataset(seqs, "cpu")
    assert len(ds) == 4 and ds.n_rows == 30 + 47 + 200 + 12 and list(ds.seq_start) == [0, 30, 77, 277]
    assert FPS_STRIDES == O.FPS_STRIDES
    T = 16
    ld = DeviceMotionLoader(ds, 2, T, fps_aug=True, root_rot=True, seed=1)
    assert len(ld) == 2
    strides = set()
    for trial in range(300):
        asked = trial % 4
        seq, stride, crop, rot = ld.sample_clip(asked)
        n2 = len(range(0, int(ds.seq_len[seq]), stride))
        assert stride in FPS_STRIDES and n2 >= T and 0 <= crop <= n2 - T
        assert seq == asked or ds.seq_len[asked] < T or len(range(0, int(ds.seq_len[asked]), 1)) < T
        assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-12)
        strides.add(stride)
        clip = ld.describe([(seq, stride, crop, rot)])[0]
        assert clip["row0"] == ds.seq_start[seq] + crop * stride
        assert clip["row0"] + (T - 1) * clip["stride"] < ds.seq_start[seq] + ds.seq_len[seq]
    assert len(strides) > 3
    # the vectorised batch sampler obeys the same rules
    for fps_aug in (False, True):
        ldb = DeviceMotionLoader(ds, 64, T, fps_aug=fps_aug, root_rot=True, seed=3)
        asked = np.arange(64) % 4
        seq, stride, crop, rot = ldb.sample_batch(asked)
        n2 = -(-ds.seq_len[seq] // stride)
        assert (n2 >= T).all() and (crop >= 0).all() and (crop <= n2 - T).all() and np.isin(stride, FPS_STRIDES).all()
        assert ((seq == asked) | (ds.seq_len[asked] < T)).all() and (seq[asked != 3] == asked[asked != 3]).all()
        assert fps_aug or (stride == 1).all()
        assert np.allclose(rot @ rot.transpose(0, 2, 1), np.eye(3), atol=1e-12) and np.allclose(np.linalg.det(rot), 1.0)
        clips = ldb.describe((seq, stride, crop, rot))
        assert (clips["row0"] + (T - 1) * clips["stride"] < ds.seq_start[seq] + ds.seq_len[seq]).all()
        one = ldb.describe([(int(seq[5]), int(stride[5]), int(crop[5]), rot[5])])[0]
        assert one == clips[5]
    u = np.array([0.3, 0.6, 0.9])
    assert np.array_equal(rand_rotation_matrix(u), O.rand_rotation_from_uniforms(u))
    # the 12-frame sequence can never be served at T = 16 without fps augmentation either
    ld2 = DeviceMotionLoader(ds, 2, T, seed=2)
    assert all(ld2.sample_clip(3)[0] != 3 for _ in range(20))
    with pytest.raises(ValueError):
        DeviceMotionLoader(ds, 2, 500)
    with pytest.raises(ValueError):
        bad = ld.describe([(2, 12, 0, None)])
        bad["row0"] = ds.n_rows - 3
        ds.assemble(bad, T, False)
    with pytest.raises(_lib.HmvaeError):                      # no CPU path: the table must live on a HIP device
        ds.assemble(ld.describe([ld.sample_clip(0)]), T, True)


def test_factory_keeps_the_reference_train_val_split(tmp_path):
    """get_train_loaders_all_data_seq (utils_motion_vae.py:194-232): train draws from train_all_amass_motion_data.json,
    val and test from val_all_amass_motion_data.json; without the id files the folder is one training pool and there is
    NO validation loader (never training-set losses under a 'Val' label)."""
    import json
    import warnings
    from hm_vae_amd.utils_motion_vae import get_train_loaders_all_data_seq, TRAIN_IDS, VAL_IDS
    rng = np.random.default_rng(0)
    folder = tmp_path / "npy"
    folder.mkdir()
    lens = {"a.npy": 40, "b.npy": 50, "c.npy": 60, "d.npy": 70, "e.npy": 80}
    for name, n in lens.items():
        np.save(str(folder / name), rng.standard_normal((n, 579)).astype(np.float32))
    cfg = dict(batch_size=2, train_seq_len=16, fps_aug_flag=False, random_root_rot_flag=False)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        tr, va, te = get_train_loaders_all_data_seq(cfg, "cpu", str(folder))
    assert va is None and te is None and len(tr.ds) == 5
    assert any("validation is OFF" in str(w.message) for w in seen)
    meta = tmp_path / "for_all_data_motion_model"
    meta.mkdir()
    json.dump({"0": "a.npy", "1": "c.npy", "2": "e.npy"}, open(str(meta / TRAIN_IDS), "w"))
    json.dump({"0": "b.npy", "1": "d.npy"}, open(str(meta / VAL_IDS), "w"))
    tr, va, te = get_train_loaders_all_data_seq(cfg, "cpu", str(folder))
    assert list(tr.ds.seq_len) == [40, 60, 80] and list(va.ds.seq_len) == [50, 70]
    assert te.ds is va.ds and tr.shuffle and not va.shuffle and te.shuffle
