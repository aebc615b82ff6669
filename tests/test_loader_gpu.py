"""GPU parity of the batch-assembly kernel (hmvae_motion_batch) against (1) the outputs of the
reference's MotionSeqData.__getitem__ recorded in tests/golden/loader.npz and (2) the CPU oracle on larger batches.

Tolerance: raw slices and the float64-standardised blocks bit-exact; the fp32 3x3 products of the root-rotation
augmentation within 1e-6 (summation order).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import GOLDEN, ROOT, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from hm_vae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def _stats():
    ms = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "all_amass_data_mean_std.npy"))
    ms[1, ms[1, :] == 0] = 1.0
    return ms


def _check(out, ref, rot):
    for k in (2, 3, 4, 5):
        assert np.array_equal(out[k], ref[k]), k
    if not rot:
        for k in (0, 1, 6):
            assert np.array_equal(out[k], ref[k]), k
    else:
        assert np.array_equal(out[1][:, 9:], ref[1][:, 9:]) and np.array_equal(out[0][:, 6:], ref[0][:, 6:])
        for k in (0, 1, 6):
            assert rel_err(out[k], ref[k]) < 1e-6, k


def test_motion_batch_vs_reference_dataset_outputs(dev):
    from hm_vae_amd.utils_motion_vae import DeviceMotionDataset, DeviceMotionLoader
    g = np.load(os.path.join(GOLDEN, "loader.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "loader_meta.json")))
    ds = DeviceMotionDataset(O.synthetic_motion_files(), dev)
    # the recorded cases grouped by (T, rot): one launch per group, clips in fixture order
    for T, rot in ((16, False), (16, True), (8, True)):
        cases = [m for m in meta if m["T"] == T and m["rot"] == rot]
        ld = DeviceMotionLoader(ds, len(cases), T, root_rot=rot)
        choices = [(m["seq"], m["stride"], m["crop"], O.rand_rotation_from_uniforms(m["randnums"]) if rot else None)
                   for m in cases]
        outs = [o.cpu().numpy() for o in ds.assemble(ld.describe(choices), T, rot)]
        assert [o.shape[2] for o in outs] == [144, 216, 72, 72, 72, 72, 3]
        for i, m in enumerate(cases):
            _check([o[i] for o in outs], [g["c%d|out%d" % (m["case"], k)] for k in range(7)], rot)


@pytest.mark.parametrize("rot", [False, True])
def test_motion_batch_full_size_vs_oracle(dev, rot):
    """B=256 x T=64 (the bench batch) from 40 synthetic sequences with fps augmentation, every clip against the oracle."""
    from hm_vae_amd.utils_motion_vae import DeviceMotionDataset, DeviceMotionLoader
    rng = np.random.default_rng(3)
    seqs = [rng.standard_normal((int(n), 579)).astype(np.float32) for n in rng.integers(70, 900, size=40)]
    ms = _stats()
    ds = DeviceMotionDataset(seqs, dev)
    B, T = 256, 64
    ld = DeviceMotionLoader(ds, B, T, fps_aug=True, root_rot=rot, seed=9)
    choices = [ld.sample_clip(int(i) % len(seqs)) for i in range(B)]
    outs = [o.cpu().numpy() for o in ds.assemble(ld.describe(choices), T, rot)]
    for i, (seq, stride, crop, M) in enumerate(choices):
        _check([o[i] for o in outs], O.motion_item(seqs[seq], ms, T, crop, stride, M), rot)
    # and the loader protocol: an epoch of 7-tuples on the device, drop_last
    ld2 = DeviceMotionLoader(ds, 16, T, seed=1)
    batches = list(ld2)
    assert len(batches) == len(ld2) == 2 and all(len(b) == 7 and b[0].is_cuda and b[0].shape == (16, T, 144) for b in batches)


def test_trainer_step_on_device_dataset(dev, tmp_path):
    """get_train_loaders_all_data_seq(data_folder=...) -> DeviceMotionLoader -> Trainer.gen_update (len8 config with both
    augmentations, as shipped) runs and yields finite losses; the batch equals the oracle's items for the same choices."""
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_motion_vae import get_train_loaders_all_data_seq
    from helpers import load_cfg
    rng = np.random.default_rng(5)
    for i in range(6):
        raw = torch.randn(40 + 7 * i, 24, 6)
        R = O.rot6d_to_matrix(raw)
        a = rng.standard_normal((raw.shape[0], 579)).astype(np.float32)
        a[:, 0:144] = O.matrix_to_rot6d(R).reshape(-1, 144).numpy()
        a[:, 144:360] = R.reshape(-1, 216).numpy()
        np.save(os.path.join(str(tmp_path), "m%d.npy" % i), a)
    cfg = load_cfg("len8")
    cfg["batch_size"] = 3
    assert cfg["fps_aug_flag"] and cfg["random_root_rot_flag"]
    import json
    from hm_vae_amd.utils_motion_vae import TRAIN_IDS, VAL_IDS
    json.dump({str(i): "m%d.npy" % i for i in range(6)}, open(os.path.join(str(tmp_path), TRAIN_IDS), "w"))
    json.dump({"0": "m4.npy", "1": "m5.npy", "2": "m0.npy"}, open(os.path.join(str(tmp_path), VAL_IDS), "w"))
    tr_ld, va_ld, te_ld = get_train_loaders_all_data_seq(cfg, dev, str(tmp_path))
    assert len(tr_ld) == 2 and len(va_ld) == 1 and te_ld.ds is va_ld.ds and tr_ld.fps_aug and tr_ld.root_rot
    tr = Trainer(cfg)
    tr.cuda()
    for it, batch in enumerate(tr_ld):
        out = tr.gen_update(batch, cfg, it, False)
        assert all(np.isfinite(v) for v in out)
        # augmented root matrices are still rotations, and the 6D input is their first two columns
        Rm = batch[1].reshape(-1, 24, 3, 3)[:, 0].cpu()
        assert float((Rm @ Rm.transpose(1, 2) - torch.eye(3)).abs().max()) < 1e-5
        assert torch.equal(batch[0].reshape(-1, 24, 2, 3)[:, :, 0].cpu(), batch[1].reshape(-1, 24, 3, 3)[:, :, :, 0].cpu())
