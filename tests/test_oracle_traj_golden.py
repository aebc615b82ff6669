"""Pin the oracle's TrajectoryModel restatement (oracle/hmvae_oracle.py::OracleTrajectory) against the unmodified
reference (trajectory_pred_model.py) run on CPU in the build container: tests/golden/traj_*.npz, tools/gen_golden.py.

Tolerance: 2e-6 rel on forward values, 2e-5 rel on gradients (fp32 summation order only).
"""
import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import load_traj_cfg, load_traj_golden, traj_batch, rel_err

torch.set_num_threads(4)
TAGS = ["len128", "len40"]


def _oracle(tag):
    g, meta = load_traj_golden(tag)
    cfg = load_traj_cfg(meta["T"])
    plan = O.TrajPlan(cfg)
    m = O.OracleTrajectory(cfg, O.traj_name_seeded_params(plan), g["mean_root"], g["std_root"])
    return g, meta, cfg, plan, m


@pytest.mark.parametrize("tag", TAGS)
def test_traj_constants_and_state_dict_layout(tag):
    g, meta, cfg, plan, m = _oracle(tag)
    assert len(meta["state_dict_keys"]) == 18
    assert plan.edge_num == [24, 14, 9, 7, 7] and plan.ch_base == [3, 6, 12, 24, 48]
    consts = {}
    for i in range(plan.L):
        consts["enc.layers.%d.0.mask" % i] = m.masks["enc.layers.%d.0.mask" % i]
        consts["enc.layers.%d.1.weight" % i] = m.pool_w[i]
    for k, v in consts.items():
        assert k in meta["state_dict_keys"], k
        s = float(v.double().sum())
        ws = float((v.double().reshape(-1) * torch.arange(1, v.numel() + 1).double() % 7.0).sum())
        assert abs(s - float(g["const|" + k + "|sum"])) < 1e-6 * max(1.0, abs(s)), k
        assert abs(ws - float(g["const|" + k + "|wsum"])) < 1e-6 * max(1.0, abs(ws)), k
    assert [n for n, _ in m.trainable()] == meta["trainable_order"]
    assert len(meta["trainable_order"]) == 10 and meta["none_grad"] == []


@pytest.mark.parametrize("tag", TAGS)
def test_traj_forward_backward_matches_reference(tag):
    g, meta, cfg, plan, m = _oracle(tag)
    out = m.train_forward(traj_batch(g))
    ret = g["ret"]          # (total, kl, 0, 0, 0, 0, rec_root_v, 0, rec_root_trans)  trajectory_pred_model.py:265
    assert rel_err([float(out["total"]), float(out["rec_root_v"]), float(out["rec_root_trans"])],
                   [ret[0], ret[6], ret[8]]) < 2e-6
    assert np.all(ret[[1, 2, 3, 4, 5, 7]] == 0.0)
    assert rel_err(out["root_v_out"].detach().numpy(), g["root_v_out"]) < 2e-6
    # per-layer activations through a forward re-run with captures
    x = traj_batch(g)[3].transpose(1, 2)
    p, P = m.plan, m.params
    with torch.no_grad():
        for i in range(p.L):
            x = O.skel_conv(x, P["enc.layers.%d.0.weight" % i], m.masks["enc.layers.%d.0.mask" % i],
                            P["enc.layers.%d.0.bias" % i], 1, p.pad, p.pad_mode)
            x = torch.nn.functional.leaky_relu(torch.matmul(m.pool_w[i], x), 0.2)
            a = x.reshape(-1).numpy()
            assert rel_err(a[g["act%d|idx" % i]], g["act%d|val" % i]) < 2e-6
            assert abs(np.linalg.norm(a.astype(np.float64)) - float(g["act%d|norm" % i])) < 2e-6 * float(g["act%d|norm" % i])
    for name, t in m.trainable():
        gr = t.grad.reshape(-1).numpy()
        pre = "grad|" + name
        assert abs(np.linalg.norm(gr.astype(np.float64)) - float(g[pre + "|norm"])) < 2e-5 * float(g[pre + "|norm"])
        if pre + "|full" in g:
            assert rel_err(gr, g[pre + "|full"]) < 2e-5, name
        else:
            assert rel_err(gr[g[pre + "|idx"]], g[pre + "|val"]) < 5e-5, name


@pytest.mark.parametrize("tag", TAGS)
def test_traj_test_branch_matches_reference(tag):
    """TrajectoryModel.test (:339-395): ground-truth and predicted absolute trajectories, (T,B,24,3)."""
    g, meta, cfg, plan, m = _oracle(tag)
    data = traj_batch(g)
    b, T, _ = data[3].shape
    with torch.no_grad():
        pose = data[2].reshape(b, T, 24, 3).transpose(0, 1)
        gt = m.trajectory(pose, data[6].transpose(0, 1))
        pred = m.trajectory(pose, m.root_velocity(data[3]).transpose(0, 1))
    assert rel_err(gt.numpy(), g["test_gt"]) < 1e-6
    assert rel_err(pred.numpy(), g["test_pred"]) < 2e-6


@pytest.mark.parametrize("tag", TAGS)
def test_traj_two_adam_steps_match_reference_trainer(tag):
    g, meta, cfg, plan, m = _oracle(tag)
    opt, sched = O.make_adam(m, cfg)
    rets = []
    for s in range(2):
        opt.zero_grad()
        out = m.train_forward(traj_batch(g))
        opt.step()
        sched.step()
        # Trainer.gen_update order (trainer_motion_vae.py:69-96): the trajectory loss rides in the 9th slot
        rets.append([float(out["total"]), 0, 0, 0, 0, 0, float(out["rec_root_v"]), 0, float(out["rec_root_trans"])])
    assert rel_err(np.array(rets), g["upd|returns"]) < 2e-6
    seeded = O.traj_name_seeded_params(plan)
    for name, t in m.trainable():
        v = t.detach().reshape(-1).numpy()
        idx = g["upd|" + name + "|idx"]
        delta = v[idx].astype(np.float64) - seeded[name].reshape(-1).numpy()[idx].astype(np.float64)
        assert np.abs(delta - g["upd|" + name + "|delta"]).max() < 2e-3 * cfg["lr"] + 1e-9, name
