"""One tiny training step of the flagship model on cuda:0, checked against the CPU oracle (driver hook)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from hm_vae_amd import _lib
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from oracle import hmvae_oracle as O            # checker only

    _lib.require_gpu()
    cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
    torch.manual_seed(0)
    tr = Trainer(cfg)
    plan = O.Plan(cfg)
    seeded = O.name_seeded_params(plan)
    sd = tr.model.state_dict()
    sd.update({k: v.clone() for k, v in seeded.items()})
    tr.model.load_state_dict(sd)
    tr.cuda()
    B = 4
    batch = O.synthetic_batch(B, cfg["train_seq_len"], seed=1)
    eps = O.level_eps(plan, B, seed=2)
    off = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "skeleton_offsets.npy"))
    orc = O.OracleVAE(cfg, seeded, offsets=off)
    ref = orc.train_forward(batch, 0, eps)
    tr.model.eps_override = eps
    ret = tr.gen_update(batch, cfg, 0, False)
    want = [float(ref[k]) for k in ("total", "kl", "rec_6d", "rec_rot", "rec_pose")]
    err = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(ret[:5], want))
    print("smoke: losses hip=%s oracle=%s max rel err %.2e" % (["%.6f" % v for v in ret[:5]], ["%.6f" % v for v in want], err))
    assert err < 1e-4, err
    print("smoke OK; library:", _lib.lib_path())
