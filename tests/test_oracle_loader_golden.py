"""Pin the oracle's data-path restatement (oracle/hmvae_oracle.py::motion_item) against the reference's own
`MotionSeqData.__getitem__` (utils_motion_vae.py:124-187) run in the build container on synthetic T x 579 files with every
random choice logged (tools/gen_golden.py::loader_golden -> tests/golden/loader.npz, loader_meta.json).

Tolerance: raw slices bit-exact; standardised values bit-exact (float64 arithmetic rounded once, like the reference);
the fp32 3x3 products of the root-rotation augmentation within 1e-6 (summation order of the matmul).
"""
import json
import os

import numpy as np

from oracle import hmvae_oracle as O
from helpers import GOLDEN, ROOT, rel_err


def _fixture():
    g = np.load(os.path.join(GOLDEN, "loader.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "loader_meta.json")))
    ms = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "all_amass_data_mean_std.npy"))
    ms[1, ms[1, :] == 0] = 1.0
    return g, meta, ms, O.synthetic_motion_files()


def test_motion_item_matches_reference_dataset():
    g, meta, ms, seqs = _fixture()
    assert len(meta) == 16
    seen = set()
    for m in meta:
        rot = O.rand_rotation_from_uniforms(m["randnums"]) if m["rot"] else None
        out = O.motion_item(seqs[m["seq"]], ms, m["T"], m["crop"], m["stride"], rot)
        ref = [g["c%d|out%d" % (m["case"], k)] for k in range(7)]
        assert [o.shape for o in out] == [r.shape for r in ref] == \
            [(m["T"], n) for n in (144, 216, 72, 72, 72, 72, 3)]
        for k in (2, 3, 4, 5):                       # raw positions and the float64-standardised blocks: exact
            assert np.array_equal(out[k], ref[k]), (m, k)
        if not m["rot"]:
            for k in (0, 1, 6):
                assert np.array_equal(out[k], ref[k]), (m, k)
        else:
            assert np.array_equal(out[1][:, 9:], ref[1][:, 9:]) and np.array_equal(out[0][:, 6:], ref[0][:, 6:])
            for k in (0, 1, 6):
                assert rel_err(out[k], ref[k]) < 1e-6, (m, k)
        seen.add((m["fps"], m["rot"], m["stride"] > 1, m["asked"] != m["seq"]))
    # the fixture exercises: fps strides > 1, a too-short sequence being re-drawn, both augmentations together
    assert any(s[2] for s in seen) and any(s[3] for s in seen) and (True, True, True, False) in seen


def test_rand_rotation_is_a_rotation_times_minus_reflection():
    """utils_motion_vae.py:16-57: (V V^T - I) Rz is orthogonal with determinant +1."""
    rng = np.random.default_rng(0)
    for _ in range(10):
        M = O.rand_rotation_from_uniforms(rng.uniform(size=3))
        assert np.allclose(M @ M.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(M) - 1.0) < 1e-12
