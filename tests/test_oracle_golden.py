"""Pin the CPU oracle (oracle/hmvae_oracle.py) against outputs of the unmodified reference.

Fixtures: tests/golden/* produced by tools/gen_golden.py (reference imported on CPU in the build
container) + the reference's own known-answer data (rest_pose_coord.npy, skeleton.py:464-477 dump).
Tolerance: 1e-6 abs/rel on forward values, 2e-5 rel on gradients (fp32 summation order only).
"""
import os
import json

import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import GOLDEN, load_cfg, load_model_golden, golden_batch, golden_eps, offsets, rel_err

torch.set_num_threads(4)

# skeleton.py:464-477 (docstring at the end of the reference file): expected derivation, 24-joint skeleton
DOC_SEQ = [[[0], [1, 4, 7, 10], [2, 5, 8, 11], [3, 6, 9], [12, 15], [13, 16, 18, 20, 22], [14, 17, 19, 21, 23]],
           [[0], [1, 2], [3, 4], [5, 6], [7], [8, 9, 10], [11, 12, 13]],
           [[0], [1], [2], [3], [4], [5, 6], [7, 8]]]
DOC_POOL = [[[0], [1, 4], [7, 10], [2, 5], [8, 11], [3], [6, 9], [12, 15], [13], [16, 18], [20, 22], [14], [17, 19],
             [21, 23]],
            [[0], [1, 2], [3, 4], [5, 6], [7], [8], [9, 10], [11], [12, 13]],
            [[0], [1], [2], [3], [4], [5, 6], [7, 8]]]
DOC_EDGES = [[(0, 24), [0, 4], [4, 10], [0, 5], [5, 11], (0, 3), [3, 9], [9, 15], (9, 13), [13, 18], [18, 22], (9, 14),
              [14, 19], [19, 23]],
             [(0, 24), [0, 10], [0, 11], [0, 9], [9, 15], (9, 13), [13, 22], (9, 14), [14, 23]],
             [(0, 24), [0, 10], [0, 11], [0, 9], [9, 15], [9, 22], [9, 23]]]


def test_topology_matches_reference_dump_and_docstring():
    topo = json.load(open(os.path.join(GOLDEN, "topology.json")))
    edges = O.skeleton_edges(O.SMPL24_PARENTS)
    assert [list(e) for e in edges] == topo["edges0"]
    cur = edges
    for lvl in range(4):
        ref = topo["levels"][lvl]
        assert O.edge_neighbours(cur, 2) == ref["neighbours"]
        seq, pool, new = O.pool_plan(cur, last_pool=(lvl == 3))
        assert seq == ref["seq_list"]
        assert pool == ref["pooling_list"]
        assert [list(e) for e in new] == ref["new_edges"]
        if lvl < 3:
            assert seq == DOC_SEQ[lvl] and pool == DOC_POOL[lvl]
            assert [list(e) for e in new] == [list(e) for e in DOC_EDGES[lvl]]
            cur = new
    # level-0 neighbour counts
    counts = [len(n) for n in O.edge_neighbours(edges, 2)]
    assert counts == [7, 8, 8, 8, 7, 7, 9, 4, 4, 9, 3, 3, 8, 9, 9, 5, 7, 7, 5, 5, 4, 4, 3, 3]


def test_fk_rest_pose_known_answer():
    rest = np.load(os.path.join(GOLDEN, "rest_pose_coord.npy"))
    off = torch.from_numpy(offsets())
    ident = torch.eye(3)[None, None].repeat(3, 24, 1, 1)
    P = O.fk_positions(ident, O.SMPL24_PARENTS, off)
    assert np.abs(P[0].numpy() - rest).max() < 1e-6


def test_rot6d_fk_against_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "fk_rot6d.npz"))
    off = torch.from_numpy(offsets())
    x = torch.from_numpy(g["r6"]).requires_grad_(True)
    R = O.rot6d_to_matrix(x)
    P = O.fk_positions(R, O.SMPL24_PARENTS, off)
    assert np.abs(R.detach().numpy() - g["R"]).max() < 1e-6
    assert np.abs(P.detach().numpy() - g["P"]).max() < 1e-6
    assert np.abs(O.fk_positions(x.detach(), O.SMPL24_PARENTS, off).numpy() - g["P_from6d"]).max() < 1e-6
    ((R * torch.from_numpy(g["wR"])).sum() + (P * torch.from_numpy(g["wP"])).sum()).backward()
    d6 = x.grad.numpy()
    # huge/tiny-magnitude rows have huge/tiny gradients: compare relative per sample
    for n in range(d6.shape[0]):
        assert rel_err(d6[n], g["d6"][n]) < 2e-5, n
    assert np.isfinite(d6).all()
    # zero 6D vector -> zero matrix, no NaN
    assert np.abs(R.detach().numpy()[0, 0]).max() == 0.0
    Rin = torch.from_numpy(g["R"]).requires_grad_(True)
    (O.fk_positions(Rin, O.SMPL24_PARENTS, off) * torch.from_numpy(g["wP"])).sum().backward()
    assert rel_err(Rin.grad.numpy(), g["dR_fkonly"]) < 2e-5
    # leaves get no position gradient (SURVEY §7 hard parts)
    for leaf in (10, 11, 15, 22, 23):
        assert np.abs(Rin.grad.numpy()[:, leaf]).max() == 0.0


def test_analytic_anchors_from_survey():
    off = torch.from_numpy(offsets())
    jj = torch.arange(24).float()[:, None] * 6 + torch.arange(6).float()[None, :]
    x = torch.sin(0.37 * jj + 0.11)[None].requires_grad_(True)
    R = O.rot6d_to_matrix(x)
    P = O.fk_positions(R, O.SMPL24_PARENTS, off)
    R00 = np.array([[0.1235326, 0.8956316, -0.4272982], [0.5196364, 0.3084568, 0.7967637],
                    [0.8454098, -0.3204660, -0.4272983]])
    assert np.abs(R[0, 0].detach().numpy() - R00).max() < 1e-6
    exp = {0: (0.0031233, -0.3514074, 0.0120366), 9: (0.1428229, -0.2846152, 0.1387344),
           15: (0.3996280, -0.2718941, 0.1538848), 22: (0.3538485, -0.0198225, 0.4713876),
           23: (0.1189282, -0.0961535, 0.2308300)}
    for j, v in exp.items():
        assert np.abs(P[0, j].detach().numpy() - np.array(v)).max() < 1e-6
    (P ** 2).sum().backward()
    d0 = np.array([0.0245563, -0.8009977, 0.4887502, 0.6978111, -1.3011769, 0.6978112])
    assert np.abs(x.grad[0, 0].numpy() - d0).max() < 2e-6
    assert np.abs(x.grad[0, 23].numpy()).max() == 0.0


def _oracle(tag):
    cfg = load_cfg(tag)
    plan = O.Plan(cfg)
    m = O.OracleVAE(cfg, O.name_seeded_params(plan), offsets=offsets())
    return cfg, plan, m


@pytest.mark.parametrize("tag", ["len64", "len8", "len16"])
def test_constants_and_state_dict_layout(tag):
    g, meta = load_model_golden(tag)
    cfg, plan, m = _oracle(tag)
    keys = meta["state_dict_keys"]
    assert len(keys) == 74
    L = plan.L
    consts = {}
    for i in range(L):
        consts["enc.layers.%d.0.mask" % i] = m.masks["enc.layers.%d.0.mask" % i]
        consts["enc.layers.%d.1.weight" % i] = m.pool_w[i]
        consts[plan.dec_conv_key(i) + ".mask"] = m.masks[plan.dec_conv_key(i) + ".mask"]
        consts[plan.dec_unpool_key(i) + ".weight"] = m.unpool_w[i]
        consts["dec.unpools.%d.weight" % i] = m.unpool_w[i]
    for k, v in consts.items():
        assert k in keys, k
        s = float(v.double().sum())
        ws = float((v.double().reshape(-1) * torch.arange(1, v.numel() + 1).double() % 7.0).sum())
        assert abs(s - float(g["const|" + k + "|sum"])) < 1e-6 * max(1.0, abs(s)), k
        assert abs(ws - float(g["const|" + k + "|wsum"])) < 1e-6 * max(1.0, abs(ws)), k
    assert [n for n, _ in m.trainable()] == meta["trainable_order"]
    assert len(meta["trainable_order"]) == 30


@pytest.mark.parametrize("tag", ["len64", "len8", "len16"])
@pytest.mark.parametrize("phase", ["p1", "p2"])
def test_train_forward_backward_matches_reference(tag, phase):
    g, meta = load_model_golden(tag)
    cfg, plan, m = _oracle(tag)
    it = 0 if phase == "p1" else cfg["iteration_interval"]
    out = m.train_forward(golden_batch(g), it, golden_eps(g))
    losses = np.array([float(out[k]) for k in ("total", "kl", "rec_6d", "rec_rot", "rec_pose")])
    assert rel_err(losses, g[phase + "|losses"]) < 1e-6
    assert rel_err([float(k) for k in out["kl_list"]], g[phase + "|kl_list"]) < 1e-6
    for i in range(4):
        assert rel_err(out["z_dist"][i].detach().numpy(), g[phase + "|z_dist%d" % i]) < 2e-6
        assert rel_err(out["z_list"][i].detach().numpy(), g[phase + "|z%d" % i]) < 2e-6
    assert rel_err(out["dec_out"].detach().numpy(), g[phase + "|dec_out"]) < 2e-6
    assert rel_err(out["R"].detach().numpy(), g[phase + "|R"]) < 2e-6
    assert rel_err(out["pos"].detach().numpy(), g[phase + "|pos"]) < 2e-6
    assert rel_err(out["dec_out"].grad.numpy(), g[phase + "|d_dec_out"]) < 2e-5
    none = []
    for name, p in m.trainable():
        if p.grad is None:
            none.append(name)
            continue
        gr = p.grad.reshape(-1).numpy()
        pre = phase + "|grad|" + name
        assert abs(np.linalg.norm(gr.astype(np.float64)) - float(g[pre + "|norm"])) < 2e-5 * float(g[pre + "|norm"]) + 1e-12
        if pre + "|full" in g:
            assert rel_err(gr, g[pre + "|full"]) < 2e-5, name
        else:
            assert rel_err(gr[g[pre + "|idx"]], g[pre + "|val"]) < 5e-5, name
    assert none == meta[phase + "_none_grad"]
    # 10 tensors without grad in phase 1, 8 in phase 2
    assert len(none) == (10 if phase == "p1" else 8)


@pytest.mark.parametrize("tag", ["len64", "len8", "len16"])
def test_two_adam_steps_match_reference_trainer(tag):
    g, meta = load_model_golden(tag)
    cfg, plan, m = _oracle(tag)
    opt, sched = O.make_adam(m, cfg)
    batch, eps = golden_batch(g), golden_eps(g)
    rets = []
    for s in range(2):
        opt.zero_grad()
        out = m.train_forward(batch, s, eps)
        opt.step()
        sched.step()
        rets.append([float(out["total"]), float(out["kl"]), float(out["rec_6d"]), float(out["rec_rot"]),
                     float(out["rec_pose"]), 0.0, 0.0, 0.0, 0.0])
    assert rel_err(np.array(rets), g["upd|returns"]) < 2e-6
    seeded = O.name_seeded_params(plan)
    for name, p in m.trainable():
        v = p.detach().reshape(-1).numpy()
        idx = g["upd|" + name + "|idx"]
        delta = v[idx].astype(np.float64) - seeded[name].reshape(-1).numpy()[idx].astype(np.float64)
        ref = g["upd|" + name + "|delta"]
        # Adam's first steps move each touched weight by ~lr: compare the deltas themselves
        assert np.abs(delta - ref).max() < 2e-3 * cfg["lr"] + 1e-9, name
    assert abs(opt.param_groups[0]["lr"] - meta["lr_after_2"]) < 1e-12


@pytest.mark.parametrize("tag", ["len64", "len8", "len16"])
def test_decoder_only_sampling_matches_reference(tag):
    g, meta = load_model_golden(tag)
    cfg, plan, m = _oracle(tag)
    with torch.no_grad():
        c6, R, pos, _ = m.decode([torch.from_numpy(g["samp|z%d" % i]) for i in range(4)])
    assert rel_err(c6.numpy(), g["samp|c6"]) < 2e-6
    assert rel_err(R.numpy(), g["samp|R"]) < 1e-4           # the lift amplifies 1e-6 noise on short 6D vectors
    assert rel_err(pos.numpy(), g["samp|pos"]) < 1e-4
