"""Test-time optimisation loop (seq_two_hier_sa_vae.py:1356-1429, masked criterion :717-735): the HIP
path against the CPU oracle's restatement of the same loop.  The released routine does not run (stale call signatures), so
there are no reference vectors for it: parity unpinned -- the oracle pieces it is assembled from (decoder, lift, FK, MSE) are
the ones the golden vectors pin.  Tolerance: 1e-4 relative (fp32) for one iteration's losses and gradients; the short Adam
trajectories are compared at 2e-3 (Adam's first steps are +-lr whatever the gradient's size, so rounding noise in tiny
gradient components is amplified to the step size)."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import hmvae_oracle as O
from helpers import CONFIGS, offsets, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    from hm_vae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def interp_cfg():
    with open(os.path.join(CONFIGS, "len_64_test_interpolation.yaml")) as f:
        return yaml.safe_load(f)


def make_model(cfg, dev):
    from hm_vae_amd.trainer_motion_vae import Trainer
    tr = Trainer(cfg)
    plan = O.Plan(cfg)
    seeded = O.name_seeded_params(plan)
    sd = tr.model.state_dict()
    for k, v in seeded.items():
        sd[k] = v.clone()
    tr.model.load_state_dict(sd)
    tr.cuda()
    return tr.model, plan, seeded


def smooth_rotations(B, T, seed):
    """A slowly turning pose per joint (valid rotation matrices), (B,T,24,3,3)."""
    g = torch.Generator().manual_seed(seed)
    a0 = torch.randn(B, 1, 24, 6, generator=g)
    a1 = torch.randn(B, 1, 24, 6, generator=g) * 0.5
    t = torch.linspace(0, 1, T).reshape(1, T, 1, 1)
    return O.rot6d_to_matrix((a0 + t * a1).reshape(-1, 6)).reshape(B, T, 24, 3, 3)


def joint_mask(B, T, window, seed):
    from hm_vae_amd.refine import temporal_mask
    m = torch.from_numpy(temporal_mask(T, window))[None, :, None].repeat(B, 1, 24)
    g = torch.Generator().manual_seed(seed)
    drop = (torch.rand(B, 1, 24, generator=g) < 0.2).float()        # a few joints never observed (completion)
    return m * (1 - drop)


def test_masked_tail_loss_unit_mask_equals_plain(dev):
    """mask == 1 everywhere: the masked kernel returns the training kernel's sums and gradient (to rounding: the masked
    variant sums a joint's squared errors before weighting them); a 0/1 mask equals zeroing the targets' error by hand."""
    from hm_vae_amd import ops
    B, T = 70, 8
    bp = ops.pad_batch(B)
    off = torch.from_numpy(offsets()).to(dev)
    torch.manual_seed(3)
    dec, g6 = torch.randn(144 * T, bp, device=dev), torch.randn(144 * T, bp, device=dev)
    gR, gP = torch.randn(216 * T, bp, device=dev), torch.randn(72 * T, bp, device=dev)
    nb = ops.decode_tail_loss_blocks(T, B, bp)
    d1, d2 = torch.empty_like(dec), torch.empty_like(dec)
    p1, p2 = torch.zeros(nb, 3, device=dev), torch.zeros(nb, 3, device=dev)
    ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 0.3, 0.2, 0.1, d1, p1)
    ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 0.3, 0.2, 0.1, d2, p2, mask=torch.ones(24 * T, bp, device=dev))
    assert float((d1 - d2).abs().max()) < 1e-6 * float(d1.abs().max())
    assert float(((p1 - p2).sum(0).abs() / p1.sum(0)).max()) < 1e-6
    # joints 3 and 17 unobserved in odd frames: same as a prediction that is exact there -- their own 6D/R/position
    # terms vanish from sums and gradient (the gradient other joints' positions send through them stays)
    mk = torch.ones(24, T, bp, device=dev)
    mk[3, 1::2] = 0
    mk[17, 1::2] = 0
    ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 0.3, 0.2, 0.1, d2, p2, mask=mk.reshape(24 * T, bp))
    # oracle on the host: autograd through lift + FK with the mask applied
    x = dec[:, :B].reshape(24, 6, T, B).permute(3, 2, 0, 1).cpu().double().requires_grad_(True)      # (B,T,24,6)
    R = O.rot6d_to_matrix(x.reshape(-1, 6)).reshape(B * T, 24, 3, 3)
    P = O.fk_positions(R, O.SMPL24_PARENTS, torch.from_numpy(offsets()).double())
    m = mk[:, :, :B].permute(2, 1, 0).cpu().double()                                                    # (B,T,24)
    t6 = g6[:, :B].reshape(24, 6, T, B).permute(3, 2, 0, 1).cpu().double()
    tR = gR[:, :B].reshape(24, 9, T, B).permute(3, 2, 0, 1).cpu().double()
    tP = gP[:, :B].reshape(24, 3, T, B).permute(3, 2, 0, 1).cpu().double()
    s6 = (((x - t6) ** 2).sum(-1) * m).sum()
    sR = (((R.reshape(B, T, 24, 9) - tR) ** 2).sum(-1) * m).sum()
    sP = (((P.reshape(B, T, 24, 3) - tP) ** 2).sum(-1) * m).sum()
    (0.3 * s6 + 0.2 * sR + 0.1 * sP).backward()
    got = p2.double().sum(0).cpu().numpy()
    assert rel_err(got, np.array([float(s6), float(sR), float(sP)])) < 1e-5
    gd = d2[:, :B].reshape(24, 6, T, B).permute(3, 2, 0, 1).cpu().numpy()
    assert rel_err(gd, x.grad.numpy()) < 1e-4
    with pytest.raises(ValueError):
        ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 0.3, 0.2, 0.1, d2, p2, mask=torch.ones(24 * T, device=dev))


def test_l2_anchor_kernel(dev):
    from hm_vae_amd import ops
    torch.manual_seed(5)
    nb = ops.l2_anchor_blocks()
    for n in (1, 255, 4097, 300000):
        p, p0, g = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
        g0 = g.clone()
        part = torch.zeros(nb, device=dev)
        assert ops.l2_anchor(p, p0, 0.25, 1.0 / n, g, part) == nb
        e = (p - p0).double()
        assert abs(float(part.double().sum()) - float((e ** 2).mean())) < 1e-5 * float((e ** 2).mean())
        assert float((g - (g0 + 0.25 * (p - p0))).abs().max()) < 1e-6
        part2 = torch.zeros(nb, device=dev)
        ops.l2_anchor(p, None, 0.0, 1.0, None, part2)               # zero anchor, loss only
        assert abs(float(part2.double().sum()) - float((p.double() ** 2).sum())) < 1e-5 * float((p.double() ** 2).sum())


@pytest.mark.parametrize("decoder_phase", [False, True])
def test_one_iteration_matches_oracle(dev, decoder_phase):
    from hm_vae_amd.refine import LatentRefiner
    cfg = interp_cfg()
    cfg["reg_w"] = 0.05                       # the shipped config has 0: exercise the term
    model, plan, seeded = make_model(cfg, dev)
    B, T = 3, cfg["train_seq_len"]
    target, mask = smooth_rotations(B, T, 1), joint_mask(B, T, cfg["interpolation_window"], 2)
    g = torch.Generator().manual_seed(9)
    z0 = torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=g)
    zd = torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=g)
    ref = LatentRefiner(model, cfg, target, mask, z0, zd)
    names = ref._decoder_names()
    # move the decoder off its anchor so that the regulariser and its gradient are not identically zero
    gen = torch.Generator().manual_seed(4)
    anchor_cpu = {n: (seeded[n] + 0.01 * torch.randn(seeded[n].shape, generator=gen)) for n in names}
    for n in names:
        ref.anchor[n].copy_(anchor_cpu[n].to(dev))
    ref._decoder_moved = True                 # (the loop skips the identically-zero term until the decoder's first update)
    ls = ref.loss_and_grads(decoder_phase).cpu().numpy()

    orc = O.OracleVAE(cfg, seeded, offsets=offsets())
    z0o, zdo = z0.clone().requires_grad_(True), zd.clone().requires_grad_(True)
    o = orc.refine_forward(z0o, zdo, target, mask, anchor=anchor_cpu)
    want = np.array([float(o[k]) for k in ("total", "rec_6d", "rec_rot", "rec_pose", "reg", "reg_decoder")])
    got = np.array([ls[0], ls[2], ls[3], ls[4], ls[5], ls[6]])
    assert rel_err(got, want) < TOL, (got, want)
    assert float(o["reg_decoder"]) > 0 and float(o["reg"]) > 0
    assert rel_err(ref.z_shallow.grad.cpu().numpy(), z0o.grad.numpy()) < TOL
    assert rel_err(ref.z_deep.grad.cpu().numpy(), zdo.grad.numpy()) < TOL
    if decoder_phase:
        for n in names:
            assert rel_err(model.engine().grad_view(n).cpu().numpy(), orc.params[n].grad.numpy()) < TOL, n
    ref.restore()


def test_keyframe_root_loss_kernel(dev):
    """hmvae_keyframe_root_loss against float64 autograd of cal_key_frame_root_loss as released (:1031-1045): ragged batch,
    windows with different selections (one with a single selected index: no term), padding columns poisoned."""
    from hm_vae_amd import ops
    T, B = 40, 70
    bp = ops.pad_batch(B)
    g = torch.Generator().manual_seed(7)
    v = torch.randn(B, T, 3, generator=g, dtype=torch.float64, requires_grad=True)
    gt = torch.randn(B, T, 3, generator=g, dtype=torch.float64)
    mean = torch.tensor([0.01, -0.02, 0.005], dtype=torch.float64)
    std = torch.tensor([0.7, 1.3, 0.05], dtype=torch.float64)
    mask = (torch.rand(B, T, 24, generator=g) < 0.6).float()
    mask[0, 0, :] = 0
    mask[0, 0, 5] = 1                                # one selected index only
    mask[1, 0, :] = 1
    acc = torch.cumsum(torch.cat((torch.zeros(B, 1, 3, dtype=torch.float64), (mean + std * v)[:, 1:]), dim=1), dim=1)
    terms = []
    for b in range(B):
        idx = torch.nonzero(mask[b][0]).squeeze(-1)
        sp, sg = acc[b][idx], gt[b][idx]
        terms.append((((sp[1:] - sp[:-1]) - (sg[1:] - sg[:-1])) ** 2).reshape(-1))
    terms = torch.cat(terms)
    n = terms.numel()
    w_t = 0.37
    (w_t * terms.mean()).backward()
    vs, gs = torch.zeros(3 * T, bp, device=dev), torch.zeros(3 * T, bp, device=dev)
    mk = torch.zeros(24 * T, bp, device=dev)
    ops.aos_to_soa(v.detach().float().to(dev).contiguous(), vs, B, T, 3, bp)
    ops.aos_to_soa(gt.float().to(dev).contiguous(), gs, B, T, 3, bp)
    ops.aos_to_soa(mask.to(dev).contiguous(), mk, B, T, 24, bp)
    vs[:, B:] = 7.0
    mk[:, B:] = 1.0
    dv = torch.full((3 * T, bp), 9.0, device=dev)
    part = torch.zeros(ops.keyframe_root_loss_blocks(T, B, bp), device=dev)
    nb = ops.keyframe_root_loss(vs, gs, mk, mean.float().to(dev), std.float().to(dev), T, B, bp, 2 * w_t / n, dv, part)
    out = torch.zeros(1, device=dev)
    ops.finalize_sum(part, nb, 1, 0, 1.0 / n, out)
    assert abs(float(out) - float(terms.mean())) < 1e-5 * float(terms.mean())
    got = torch.empty(B, T, 3, device=dev)
    ops.soa_to_aos(dv, got, B, T, 3, bp)
    assert rel_err(got.cpu().numpy(), v.grad.numpy()) < 1e-5
    assert float(dv[:, B:].abs().max()) == 0.0


def test_trajectory_term_matches_oracle(dev):
    """optimize_trajectory (:1376-1389): one iteration's losses and latent gradients with the trajectory term against the
    oracle (frozen trajectory model of the shipped trajectory config at the window length, name-seeded weights)."""
    from hm_vae_amd.refine import LatentRefiner
    from hm_vae_amd.trainer_motion_vae import Trainer
    from helpers import load_traj_cfg
    cfg = interp_cfg()
    cfg.update(reg_w=0.05, optimize_trajectory=True, reg_w_trajectory=3.0)
    model, plan, seeded = make_model(cfg, dev)
    B, T = 3, cfg["train_seq_len"]
    tcfg = load_traj_cfg(T)
    ttr = Trainer(tcfg)
    tplan = O.TrajPlan(tcfg)
    tseed = O.traj_name_seeded_params(tplan)
    sd = ttr.model.state_dict()
    for k, v in tseed.items():
        sd[k] = v.clone()
    ttr.model.load_state_dict(sd)
    ttr.cuda()
    tm = ttr.model
    target, mask = smooth_rotations(B, T, 21), joint_mask(B, T, cfg["interpolation_window"], 22)
    g = torch.Generator().manual_seed(29)
    z0 = torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=g)
    zd = torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=g)
    root = torch.cumsum(0.02 * torch.randn(B, T, 3, generator=g), dim=1)
    ref = LatentRefiner(model, cfg, target, mask, z0, zd, trajectory_model=ttr, root_trans=root)
    ls = ref.loss_and_grads(False).cpu().numpy()

    orc = O.OracleVAE(cfg, seeded, offsets=offsets())
    mean, std = tm.mean_vals[0].cpu(), tm.std_vals[0].cpu()
    otr = O.OracleTrajectory(tcfg, tseed, mean[576:579].numpy(), std[576:579].numpy())
    z0o, zdo = z0.clone().requires_grad_(True), zd.clone().requires_grad_(True)
    o = orc.refine_forward(z0o, zdo, target, mask, traj=otr, root_trans=root, joint_mean=mean[360:432], joint_std=std[360:432])
    assert float(o["trajectory"]) > 0
    want = np.array([float(o[k]) for k in ("total", "rec_6d", "rec_rot", "rec_pose", "reg", "trajectory")])
    got = np.array([ls[0], ls[2], ls[3], ls[4], ls[5], ls[7]])
    assert rel_err(got, want) < TOL, (got, want)
    assert rel_err(ref.z_shallow.grad.cpu().numpy(), z0o.grad.numpy()) < TOL
    assert rel_err(ref.z_deep.grad.cpu().numpy(), zdo.grad.numpy()) < TOL
    # the term is what moved the gradient: without it the latents' gradient differs
    cfg2 = dict(cfg, optimize_trajectory=False)
    ref2 = LatentRefiner(model, cfg2, target, mask, z0, zd)
    ref2.loss_and_grads(False)
    assert rel_err(ref2.z_deep.grad.cpu().numpy(), zdo.grad.numpy()) > 10 * TOL
    with pytest.raises(ValueError):
        LatentRefiner(model, cfg, target, mask, z0, zd)                   # the term needs the trajectory model
    # the loop with the term: hipGraph replay (second iteration captured, then replayed) == eager launches
    hists = []
    for use_graph in (False, True):
        r = LatentRefiner(model, cfg, target, mask, z0, zd, use_graph=use_graph, trajectory_model=ttr, root_trans=root)
        hists.append(r.run(5))
        r.restore()
    assert rel_err(hists[1], hists[0]) < 2e-6
    assert (hists[0][:, 7] > 0).all() and hists[0][-1, 0] < hists[0][0, 0]
    del r, ref, ref2
    import gc
    gc.collect()


def test_short_loop_matches_oracle_adam(dev, monkeypatch):
    """3 latent iterations then 3 decoder iterations (the phase switch pulled in from 50) against torch.optim.Adam on the
    oracle: same loss history."""
    from hm_vae_amd import refine
    monkeypatch.setattr(refine, "DECODER_PHASE_AFTER", 2)
    cfg = interp_cfg()
    cfg["reg_w"] = 0.05
    model, plan, seeded = make_model(cfg, dev)
    B, T = 2, cfg["train_seq_len"]
    target, mask = smooth_rotations(B, T, 11), joint_mask(B, T, cfg["interpolation_window"], 12)
    g = torch.Generator().manual_seed(19)
    z0 = torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=g)
    zd = torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=g)
    ref = refine.LatentRefiner(model, cfg, target, mask, z0, zd)
    names = ref._decoder_names()
    orc = O.OracleVAE(cfg, seeded, offsets=offsets())
    anchor = {n: orc.params[n].detach().clone() for n in names}
    # (1) Every point of the HIP trajectory, teacher-forced: the oracle evaluates the losses AT the refiner's current
    # latents and decoder, so each iteration is a single-step comparison at 1e-4 -- what the kernels compute -- free of
    # what Adam does to rounding differences over several steps.
    rows = []
    for i in range(6):
        zs, zdp = ref.z_shallow.detach().cpu().clone(), ref.z_deep.detach().cpu().clone()
        with torch.no_grad():
            for n in names:
                orc.params[n].copy_(model.engine().params[n].detach().cpu())
        o = orc.refine_forward(zs, zdp, target, mask, anchor=anchor, backward=False)
        want_i = np.array([float(o[k]) for k in ("total", "rec_6d", "rec_rot", "rec_pose", "reg")])
        rows.append(ref.step().detach().cpu().numpy().copy())
        assert rel_err(rows[-1][[0, 2, 3, 4, 5]], want_i) < TOL, (i, rows[-1], want_i)
        if i > 3:                                     # the decoder has moved: its regulariser against the oracle's
            assert abs(rows[-1][6] - float(o["reg_decoder"])) < 1e-3 * float(o["reg_decoder"]) + 1e-12
    hist = np.array(rows)
    ref.restore()
    for n in names:                                   # the model's decoder is what it was
        assert torch.equal(model.engine().params[n].detach().cpu(), seeded[n]), n

    # (2) the free-running trajectories (HipAdam vs torch.optim.Adam on the oracle) stay together: sign-like Adam steps at
    # lr 0.1 amplify fp32 rounding differences, hence the looser bound on this part only
    orc = O.OracleVAE(cfg, seeded, offsets=offsets())
    z0o, zdo = z0.clone().requires_grad_(True), zd.clone().requires_grad_(True)
    opt_z = torch.optim.Adam([z0o, zdo], lr=cfg["opt_lr"], weight_decay=cfg["weight_decay"])
    opt_d = torch.optim.Adam([orc.params[n] for n in names], lr=cfg["opt_lr"] * 0.001, weight_decay=cfg["weight_decay"])
    want = []
    for i in range(6):
        o = orc.refine_forward(z0o, zdo, target, mask, anchor=anchor)
        want.append([float(o[k]) for k in ("total", "rec_6d", "rec_rot", "rec_pose", "reg", "reg_decoder")])
        (opt_d if i > 2 else opt_z).step()
    want = np.array(want)
    got = hist[:, [0, 2, 3, 4, 5, 6]]
    assert want[-1, 0] < want[0, 0]
    for i in range(6):
        assert rel_err(got[i, :5], want[i, :5]) < 2e-3, (i, got[i], want[i])
    # the decoder regulariser starts from exactly zero and grows by (lr * 0.001)^2-sized steps
    assert np.all(got[:4, 5] == 0.0) and got[5, 5] > 0
    assert abs(got[5, 5] - want[5, 5]) < 0.05 * want[5, 5]


def test_full_loop_interpolates_and_restores(dev):
    """The shipped interpolation configuration (150 iterations, decoder phase from 51) through the model method."""
    cfg = interp_cfg()
    model, plan, seeded = make_model(cfg, dev)
    B, T = 4, cfg["train_seq_len"]
    from hm_vae_amd.refine import temporal_mask
    target = smooth_rotations(B, T, 21)
    mask = torch.from_numpy(temporal_mask(T, cfg["interpolation_window"]))[None, :, None].repeat(B, 1, 24)
    torch.manual_seed(0)
    c6, R, pos, hist = model.refine_window(target, mask)
    assert hist.shape == (cfg["opt_it"], 8) and np.all(np.isfinite(hist))
    assert hist[-1, 0] < 0.5 * hist[0, 0], (hist[0], hist[-1])
    assert hist[50, 0] <= hist[0, 0]
    # outputs are a valid decode: rotations orthonormal, 6D consistent with R's first column
    Rf = R.reshape(-1, 3, 3)
    assert float((Rf.transpose(1, 2) @ Rf - torch.eye(3, device=dev)).abs().max()) < 1e-4
    # reported masked loss == the criterion evaluated on the outputs of one more decode (decoder restored -> compare on
    # the latent-phase value instead: iteration 0's loss against the torch criterion at the initial latents is covered by
    # test_one_iteration_matches_oracle); here: the decoder is restored bit for bit
    for n, t in model.engine().params.items():
        if n in seeded:
            assert torch.equal(t.detach().cpu(), seeded[n]), n
    l, per_frame = model.l2_masked_criterion(R, target.to(dev), mask.to(dev))
    assert per_frame.shape == (B, T) and float(l) >= 0


def test_one_refiner_serves_window_after_window(dev):
    """TwoHierSAVAEModel.refine_window keeps ONE refiner (buffers, optimisers, two captured graphs) per window shape: a
    second and third window replay the same graphs from iteration 0, and the state is fully reset in between -- window A
    run again after window B repeats its first run."""
    from hm_vae_amd import graph_step
    cfg = interp_cfg()
    cfg["opt_it"] = 60                                # 51 latent + 9 decoder iterations
    model, plan, seeded = make_model(cfg, dev)
    B, T = 2, cfg["train_seq_len"]
    wins = [(smooth_rotations(B, T, 41 + k), joint_mask(B, T, cfg["interpolation_window"], 51 + k)) for k in range(2)]
    g = torch.Generator().manual_seed(61)
    z = [(torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=g),
          torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=g)) for _ in range(2)]
    runs = []
    import gc
    gc.collect()                                      # refiners of earlier tests park their graphs when they are collected
    parked = len(graph_step._RETIRED)
    for k in (0, 1, 0):
        c6, R, pos, hist = model.refine_window(wins[k][0], wins[k][1], z_shallow=z[k][0], z_deep=z[k][1])
        runs.append((hist.copy(), pos.cpu().numpy().copy()))
        for n, t in model.engine().params.items():    # decoder restored bit for bit after every window
            if n in seeded:
                assert torch.equal(t.detach().cpu(), seeded[n]), n
    assert len(model._refiners) == 1
    ref = next(iter(model._refiners.values()))
    assert len(ref._graphs) == 2 and len(graph_step._RETIRED) == parked        # two captures in all, nothing retired
    assert rel_err(runs[2][0], runs[0][0]) < 1e-6 and rel_err(runs[2][1], runs[0][1]) < 1e-6
    assert rel_err(runs[1][0][:, 0], runs[0][0][:, 0]) > 1e-3                  # (window B really is another problem)


def test_graph_replay_equals_eager(dev, monkeypatch):
    """The per-phase hipGraph replay runs the same launches as the eager loop: same loss history, same result."""
    from hm_vae_amd import graph_step, refine
    monkeypatch.setattr(refine, "DECODER_PHASE_AFTER", 3)
    monkeypatch.setattr(graph_step, "MAX_RETIRED", 10 ** 6)      # (a long test session may have parked its quota already)
    cfg = interp_cfg()
    model, plan, seeded = make_model(cfg, dev)
    B, T = 1, cfg["train_seq_len"]                    # the reference's batch: one window
    target, mask = smooth_rotations(B, T, 31), joint_mask(B, T, cfg["interpolation_window"], 32)
    g = torch.Generator().manual_seed(33)
    z0 = torch.randn(B, plan.edge_num[1], plan.lat_dim[0], generator=g)
    zd = torch.randn(B, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1], generator=g)
    out = []
    for use_graph in (False, True):
        ref = refine.LatentRefiner(model, cfg, target, mask, z0, zd, use_graph=use_graph)
        hist = ref.run(9)
        c6, R, pos = ref.decode()
        assert (len(ref._graphs) == 2) == use_graph
        ref.restore()
        out.append((hist, pos.cpu().numpy()))
    assert rel_err(out[1][0][:, 0], out[0][0][:, 0]) < 1e-5
    assert rel_err(out[1][1], out[0][1]) < 1e-5
