"""Data parallelism: one process per GPU, ONE flat fp32 gradient bucket, one all-reduce (mean) per step.

The reference's only multi-GPU switch is nn.DataParallel (train_motion_vae.py:49-53: single process, gradients
SUMMED across replicas).  Here the batch is sharded by rank; because every loss is a mean over equal-size shards,
mean-of-rank-gradients equals the single-GPU large-batch gradient.  Parameters that have no
gradient in the current phase sit at the tail of the bucket and are left out of the collective on every rank.
Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def init_from_env(backend=None):
    """torchrun-style env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # HMVAE_DIST_BACKEND=gloo: rehearse the multi-rank control flow where RCCL cannot run (all ranks on one GPU)
            backend = os.environ.get("HMVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rk, world_size=world)
    return rk, world, local


def all_reduce_mean_(flat):
    """In-place mean over ranks of a contiguous 1-D tensor (a prefix of the flat gradient bucket)."""
    if not is_distributed():
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / dist.get_world_size())
    return flat


def all_reduce_sum_async(flat):
    """Start an in-place SUM all-reduce; returns the work handle (None when not distributed)."""
    if not is_distributed():
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)


def all_reduce_sum_(flat):
    if is_distributed() and flat.numel():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def wait(handle):
    if handle is not None:
        handle.wait()


def group_ready():
    """A process group exists (any world size): collectives can be issued.  A world of ONE rank on RCCL is how the
    in-graph all-reduce path is exercised on a one-GPU box (the collective is the identity, the capture is real)."""
    return dist.is_available() and dist.is_initialized()


def reduce_between_segments(flat, n_dec, n_hi, n_act, run_b, run_c=None):
    """The split DDP step after segment A (forward + decoder backward: flat[:n_dec] final) has been queued:

        all-reduce(flat[:n_dec]) asynchronously  ||  run_b()   (encoder backward; with run_c: its coarse half, after
                                                              which flat[n_dec:n_hi] is final)
        [all-reduce(flat[n_dec:n_hi]) asynchronously  ||  run_c()   (the fine layers: flat[n_hi:n_act] final)]
        all-reduce(the remaining slice), wait for the asynchronous ones, sum -> mean over flat[:n_act]

    run_b / run_c only QUEUE device work (graph replays); the collectives run on the backend's own stream behind the work
    queued before them.  Same result as one all-reduce of flat[:n_act] after everything: the slices are disjoint.
    graph_step.GraphedStep.run is the caller; tests/test_ddp_gloo_cpu.py drives it with two gloo ranks."""
    dist_on = is_distributed()
    handles = [all_reduce_sum_async(flat[:n_dec])] if dist_on else []
    run_b()
    start = n_dec
    if run_c is not None:
        if dist_on:
            handles.append(all_reduce_sum_async(flat[n_dec:n_hi]))
        start = n_hi
        run_c()
    if dist_on:
        all_reduce_sum_(flat[start:n_act])
        for h in handles:
            wait(h)
        scale_(flat[:n_act])


def scale_(flat):
    """sum -> mean over the ranks."""
    if is_distributed():
        flat.mul_(1.0 / dist.get_world_size())
    return flat


def broadcast_parameters(params, src=0):
    """Make every rank start from rank `src`'s weights (one coalesced broadcast)."""
    if not is_distributed():
        return
    params = [p for p in params]
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    o = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[o:o + n].view_as(p))
            o += n


def shard_batch(data, rk, world):
    """Contiguous batch shard of a tuple of (B, ...) tensors; B must divide evenly (equal shards keep the mean-of-
    means identity exact)."""
    B = data[0].shape[0]
    if B % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (B, world))
    n = B // world
    return tuple(t[rk * n:(rk + 1) * n] for t in data)
