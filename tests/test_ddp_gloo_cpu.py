"""world_size-2 `gloo` tests of the data-parallel path (hm_vae_amd/ddp.py + the engine's flat gradient bucket).

No GPU: the per-rank gradients come from the CPU oracle; what is under test is the host logic that has to be right
by construction on 8 GPUs — bucket layout, which tensors are left out of the collective in each phase, the mean
all-reduce, parameter broadcast and batch sharding — and the identity it relies on (mean of equal-shard gradients ==
full-batch gradient).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_cfg, offsets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tag, phase2, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import hmvae_oracle as O
    from hm_vae_amd import ddp
    from hm_vae_amd.engine import VaeEngine
    from hm_vae_amd.trainer_motion_vae import Trainer
    r, w, _ = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and ddp.is_distributed() and ddp.rank() == rank and ddp.world_size() == world
    cfg = load_cfg(tag)
    plan = O.Plan(cfg)
    seeded = O.name_seeded_params(plan)
    # every rank starts from different weights; broadcast must make them rank 0's
    torch.manual_seed(100 + rank)
    tr = Trainer(cfg)
    params = [p for p in tr.model.parameters() if p.requires_grad]
    ddp.broadcast_parameters(params)
    chk = torch.stack([p.detach().double().sum() for p in params])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(g, gathered[0]) for g in gathered)

    B = 4
    it = cfg["iteration_interval"] if phase2 else 0
    full = O.synthetic_batch(B, cfg["train_seq_len"], seed=5)
    eps_full = O.level_eps(plan, B, seed=6)
    shard = ddp.shard_batch(full, rank, world)
    n = B // world
    eps_shard = [e.reshape(B, -1, e.shape[1])[rank * n:(rank + 1) * n].reshape(-1, e.shape[1]) for e in eps_full]
    orc = O.OracleVAE(cfg, seeded, offsets=offsets())
    orc.train_forward(shard, it, eps_shard)
    # the engine plans its bucket on CPU (no kernels run); fill each live tensor's slice with this rank's gradient
    eng = VaeEngine(cfg, dict(tr.model.named_parameters()), offsets(), "cpu")
    order, offs, n1, n2, always, ph2, never = eng.grad_layout()
    eng.assign_grads(phase2)
    live = 0
    for name, p in orc.trainable():
        mine = eng.params[name]
        if p.grad is None:
            assert mine.grad is None, name
            continue
        assert mine.grad is not None, name
        mine.grad.copy_(p.grad)
        live += 1
    assert live == (22 if phase2 else 20)
    nfl = eng.active_grad_floats(phase2)
    assert nfl == (n2 if phase2 else n1)
    tail_before = eng.flat_grad[nfl:].clone()
    # the two-bucket path of graph_step.py (decoder slice asynchronously, then the rest, then one scaling) must give
    # what the single all-reduce gives
    two = eng.flat_grad.clone()
    nd = eng.n_dec_floats
    assert 0 < nd < nfl and all(offs[nm] < nd for nm in order if nm.startswith("dec.") and nm in always) \
        and all(offs[nm] >= nd for nm in order if nm.startswith("enc."))
    h = ddp.all_reduce_sum_async(two[:nd])
    ddp.all_reduce_sum_(two[nd:nfl])
    ddp.wait(h)
    ddp.scale_(two[:nfl])
    # the three-bucket path: [decoder | coarse encoder half + deep latent head | fine encoder layers (+ phase-2 head)]
    three = eng.flat_grad.clone()
    nh = eng.n_hi_floats
    hi_names = ["enc.layers.%d.0.weight" % i for i in range(eng.enc_cut, eng.p.L)] + \
        ["enc.latent_enc_layers.%d.weight" % (eng.p.L - 1)]
    lo_names = ["enc.layers.%d.0.weight" % i for i in range(eng.enc_cut)] + ["enc.latent_enc_layers.0.weight"]
    assert nd < nh < nfl and all(nd <= offs[nm] < nh for nm in hi_names) and all(offs[nm] >= nh for nm in lo_names)
    h1 = ddp.all_reduce_sum_async(three[:nd])
    h2 = ddp.all_reduce_sum_async(three[nd:nh])
    ddp.all_reduce_sum_(three[nh:nfl])
    ddp.wait(h1)
    ddp.wait(h2)
    ddp.scale_(three[:nfl])
    ddp.all_reduce_mean_(eng.flat_grad[:nfl])
    assert torch.allclose(two, eng.flat_grad, rtol=1e-6, atol=0)
    assert torch.allclose(three, eng.flat_grad, rtol=1e-6, atol=0)
    assert torch.equal(eng.flat_grad[nfl:], tail_before)          # tensors without gradient are not communicated
    if rank == 0:
        ref = O.OracleVAE(cfg, seeded, offsets=offsets())
        ref.train_forward(full, it, eps_full)
        worst = 0.0
        for name, p in ref.trainable():
            if p.grad is None:
                continue
            a = eng.params[name].grad.numpy().astype(np.float64)
            b = p.grad.numpy().astype(np.float64)
            worst = max(worst, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
        with open(os.path.join(out_dir, "result_%s_%d.txt" % (tag, int(phase2))), "w") as f:
            f.write("%.3e" % worst)
    # uneven shards are refused (the mean-of-means identity needs equal shards)
    try:
        ddp.shard_batch(tuple(t[:3] for t in full), rank, world)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("phase2", [False, True])
def test_two_rank_gradient_bucket_allreduce(tmp_path, phase2):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, "len8", phase2, str(tmp_path)), nprocs=2, join=True)
    worst = float(open(os.path.join(str(tmp_path), "result_len8_%d.txt" % int(phase2))).read())
    # mean of the two half-batch gradients == full-batch gradient, up to fp32 summation order
    assert worst < 2e-5, worst


def _traj_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from oracle import hmvae_oracle as O
    from hm_vae_amd import ddp
    from hm_vae_amd.traj_engine import TrajEngine
    from hm_vae_amd.trainer_motion_vae import Trainer
    from helpers import load_traj_cfg, load_traj_golden
    ddp.init_from_env(backend="gloo")
    T, B = 40, 4
    cfg = load_traj_cfg(T)
    plan = O.TrajPlan(cfg)
    seeded = O.traj_name_seeded_params(plan)
    g, _ = load_traj_golden("len40")
    torch.manual_seed(100 + rank)
    tr = Trainer(cfg)
    params = [p for p in tr.model.parameters() if p.requires_grad]
    ddp.broadcast_parameters(params)
    chk = torch.stack([p.detach().double().sum() for p in params])
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    assert all(torch.equal(x, gathered[0]) for x in gathered)
    full = O.synthetic_traj_batch(B, T, seed=7)
    shard = ddp.shard_batch(full, rank, world)
    orc = O.OracleTrajectory(cfg, seeded, g["mean_root"], g["std_root"])
    orc.train_forward(shard)
    eng = TrajEngine(cfg, dict(tr.model.named_parameters()), g["std_root"], "cpu")
    eng.assign_grads()
    for name, p in orc.trainable():
        eng.params[name].grad.copy_(p.grad)
    nfl = eng.active_grad_floats()
    assert nfl == eng.flat_grad.numel()
    ddp.all_reduce_mean_(eng.flat_grad[:nfl])
    if rank == 0:
        ref = O.OracleTrajectory(cfg, seeded, g["mean_root"], g["std_root"])
        ref.train_forward(full)
        worst = 0.0
        for name, p in ref.trainable():
            a = eng.params[name].grad.numpy().astype(np.float64)
            b = p.grad.numpy().astype(np.float64)
            worst = max(worst, float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
        with open(os.path.join(out_dir, "result_traj.txt"), "w") as f:
            f.write("%.3e" % worst)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trajectory_bucket_allreduce(tmp_path):
    """TrajectoryModel: both losses are batch means, so the mean of the two half-batch gradients is the full-batch one."""
    port = _free_port()
    mp.spawn(_traj_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    worst = float(open(os.path.join(str(tmp_path), "result_traj.txt")).read())
    assert worst < 5e-5, worst


def _segments_worker(rank, world, port, out_dir, three):
    """ddp.reduce_between_segments as GraphedStep.run drives it, on CPU tensors over gloo: segment B / C are closures that
    write their slices of the bucket when they run (the role of the graph replays), so a slice reduced before its segment
    ran, a wrong boundary or a missing wait shows up as a wrong mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from hm_vae_amd import ddp
    from hm_vae_amd.engine import VaeEngine
    from helpers import load_cfg, offsets
    from oracle import hmvae_oracle as O
    ddp.init_from_env(backend="gloo")
    cfg = load_cfg("len8")
    plan = O.Plan(cfg)
    eng = VaeEngine(cfg, O.name_seeded_params(plan), offsets(), torch.device("cpu"))        # the real bucket layout, planned on CPU
    eng.grad_layout()
    n_dec, n_hi, n_act = eng.n_dec_floats, eng.n_hi_floats, eng.active_grad_floats(True)
    assert 0 < n_dec < n_hi < n_act <= eng.flat_grad.numel()
    g = torch.Generator().manual_seed(11 + rank)
    full = torch.randn(eng.flat_grad.numel(), generator=g)                                  # this rank's complete gradient
    flat = eng.flat_grad
    flat.zero_()
    flat[:n_dec] = full[:n_dec]                                                             # segment A has run
    order = []

    def seg_b():
        order.append("b")
        hi = n_hi if three else n_act
        flat[n_dec:hi] = full[n_dec:hi]

    def seg_c():
        order.append("c")
        flat[n_hi:n_act] = full[n_hi:n_act]

    ddp.reduce_between_segments(flat, n_dec, n_hi, n_act, seg_b, seg_c if three else None)
    assert order == (["b", "c"] if three else ["b"])
    both = [torch.zeros_like(full) for _ in range(world)]
    dist.all_gather(both, full)
    want = sum(both) / world
    err = float((flat[:n_act] - want[:n_act]).abs().max())
    tail_untouched = bool((flat[n_act:] == 0).all())
    if rank == 0:
        with open(os.path.join(out_dir, "result_seg_%d.txt" % int(three)), "w") as f:
            f.write("%.3e %d" % (err, int(tail_untouched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("three", [False, True])
def test_two_rank_split_step_reduce_sequence(tmp_path, three):
    """The multi-graph DDP step's collective sequence (two or three bucket slices, the first ones asynchronous) gives the
    mean of the ranks' complete gradients on the engine's real bucket boundaries, and leaves the inactive tail alone."""
    port = _free_port()
    mp.spawn(_segments_worker, args=(2, port, str(tmp_path), three), nprocs=2, join=True)
    err, tail = open(os.path.join(str(tmp_path), "result_seg_%d.txt" % int(three))).read().split()
    assert float(err) < 1e-6 and tail == "1"


def test_single_process_helpers_are_noops():
    from hm_vae_amd import ddp
    assert not ddp.is_distributed() and ddp.rank() == 0 and ddp.world_size() == 1
    t = torch.arange(4.0)
    assert ddp.all_reduce_mean_(t) is t and torch.equal(t, torch.arange(4.0))
    ddp.broadcast_parameters([torch.zeros(3)])


def test_launcher_counts_gpus_from_kfd_topology(tmp_path, monkeypatch):
    """bench.py's launcher parent counts GPUs from the kernel driver's topology files (simd_count > 0), without any HIP
    call: two CPU nodes + three GPU nodes -> 3; HIP_VISIBLE_DEVICES narrows it; no topology -> None (no check)."""
    import bench
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert bench.kfd_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.kfd_gpu_count(str(tmp_path)) == 2
    assert bench.kfd_gpu_count(str(tmp_path / "nothing_here")) is None
