"""CPU structural checks of the fused engine's planning (no kernels run): construction, buffer/row maps, the flat
gradient bucket layout and the class surface."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hmvae_oracle as O
from helpers import GOLDEN, load_cfg, offsets


@pytest.mark.parametrize("tag", ["len64", "len8", "len16"])
def test_engine_plans_on_cpu(tag):
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.engine import VaeEngine
    from hm_vae_amd import _lib
    cfg = load_cfg(tag)
    tr = Trainer(cfg)
    meta = json.load(open(os.path.join(GOLDEN, "model_meta.json")))[tag]
    assert list(tr.model.state_dict().keys()) == meta["state_dict_keys"]
    assert [n for n, p in tr.model.named_parameters() if p.requires_grad] == meta["trainable_order"]
    eng = VaeEngine(cfg, dict(tr.model.named_parameters()), offsets(), "cpu")
    plan = O.Plan(cfg)
    # plan agrees with the oracle's (which is pinned to the reference)
    p = eng.p
    assert p.neigh == plan.neigh and p.pooling == plan.pooling
    assert p.enc_cin == plan.enc_cin and p.enc_cout == plan.enc_cout
    assert p.dec_cin == plan.dec_cin and p.dec_cout == plan.dec_cout and p.dec_up == plan.dec_up
    assert p.strides == plan.strides and p.tsteps == plan.tsteps and p.lat_dim == plan.lat_dim
    # flat gradient bucket: phase-1 prefix excludes exactly the reference's None-grad tensors
    order, offs, n1, n2, always, ph2, never = eng.grad_layout()
    trainable = set(meta["trainable_order"])
    assert set(order) == trainable
    assert trainable - always == set(meta["p1_none_grad"])
    assert trainable - always - ph2 == set(meta["p2_none_grad"])
    assert n1 < n2 <= eng.flat_grad.numel()
    for n in order:
        assert offs[n] % 4 == 0
    # mask-effective forward FLOPs per sequence (225.47 MFLOP for len64 convs)
    if tag == "len64":
        assert abs(eng.flops_per_seq / 1e6 - 225.47) < 0.5
    # no compute without a GPU: loud failure, not a fallback
    with pytest.raises(_lib.HmvaeError):
        eng.workspace(4)
    with pytest.raises(_lib.HmvaeError):
        tr.model(O.synthetic_batch(2, cfg["train_seq_len"]), cfg, 0)


@pytest.mark.parametrize("tag", ["len128", "len40"])
def test_traj_engine_plans_on_cpu(tag):
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.traj_engine import TrajEngine
    from hm_vae_amd import _lib
    from helpers import load_traj_cfg
    meta = json.load(open(os.path.join(GOLDEN, "traj_meta.json")))[tag]
    cfg = load_traj_cfg(meta["T"])
    tr = Trainer(cfg)
    assert list(tr.model.state_dict().keys()) == meta["state_dict_keys"]
    assert [n for n, p in tr.model.named_parameters() if p.requires_grad] == meta["trainable_order"]
    assert [n for n, p in tr.model.named_parameters()] == meta["all_param_order"]
    eng = TrajEngine(cfg, dict(tr.model.named_parameters()), [1.0, 1.0, 1.0], "cpu")
    plan = O.TrajPlan(cfg)
    p = eng.p
    assert p.neigh == plan.neigh and p.pooling == plan.pooling and p.edge_num == plan.edge_num
    assert p.enc_cin == plan.enc_cin and p.enc_cout == plan.enc_cout and p.fc_in == plan.fc_in == 336
    order, offs, n1, n2 = eng.grad_layout()
    assert order == meta["trainable_order"] and n1 == n2 == eng.flat_grad.numel()
    # mask-effective forward MACs per sequence = sum over layers of nnz(mask[:, :, 0]) * K * T, + the 336x3 head per frame
    macs = sum(int(m[:, :, 0].sum()) * plan.K * plan.T for m in O.traj_masks(plan).values()) + 3 * 336 * plan.T
    assert eng.flops_per_seq == 2 * macs
    with pytest.raises(_lib.HmvaeError):
        eng.workspace(4)
    with pytest.raises(_lib.HmvaeError):
        tr.model(O.synthetic_traj_batch(2, meta["T"]), cfg, 0)


def test_c_abi_exports_match_header():
    """Every function declared in include/hmvae.h is exported by the built library and bound in _lib."""
    import re
    from hm_vae_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "hmvae.h")).read()
    declared = set(re.findall(r"\b(hmvae_[a-z0-9_]+)\s*\(", hdr))
    lib = _lib.load()
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.hmvae_version() >= 1
    assert b"argument" in lib.hmvae_error_string(-1)
    # argument validation happens on the host before any launch
    d = _lib.GemmDesc()
    import ctypes as C
    assert lib.hmvae_gather_gemm(C.byref(d), None) == -1


def test_retired_graphs_are_parked_not_destroyed():
    """graph_step.retire keeps replaced graphs alive on HIP runtimes older than 7.2 (the bundled 7.0.2 crashes after a
    multi-stream hipGraphExec is destroyed: graph_step.py, tools/repro/); callers that capture per work item stop
    capturing at MAX_RETIRED and say so once.  On a newer runtime nothing is parked."""
    import warnings
    from hm_vae_amd import graph_step
    before, park, warned = list(graph_step._RETIRED), graph_step.PARK, graph_step._warned[0]
    try:
        graph_step.PARK, graph_step._warned[0] = True, False
        a, b = object(), object()
        graph_step.retire(a, None, b)
        assert graph_step._RETIRED[-2:] == [a, b] and graph_step.may_capture()
        graph_step._RETIRED.extend([object()] * graph_step.MAX_RETIRED)
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            assert not graph_step.may_capture()
            assert not graph_step.may_capture()
        assert len(seen) == 1 and "parked" in str(seen[0].message)
        graph_step.PARK = False                       # a runtime that destroys graphs safely: nothing is kept
        n = len(graph_step._RETIRED)
        graph_step.retire(None)
        assert len(graph_step._RETIRED) == n and graph_step.may_capture()
    finally:
        graph_step._RETIRED[:] = before
        graph_step.PARK, graph_step._warned[0] = park, warned


def test_adam_table_cache_never_evicts_live_entries():
    """HipAdam's table cache: entries a captured hipGraph addresses (`live`) hold the device tensors the replay reads and
    counts in -- eviction may only drop the others, oldest first."""
    from hm_vae_amd.optim import HipAdam
    opt = HipAdam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)
    for i in range(opt.MAX_CACHE):
        opt._cache[("k", i)] = dict(live=(i % 4 == 0), tab=object(), step=object())
    opt._evict()
    assert len(opt._cache) == opt.MAX_CACHE - 1
    assert ("k", 1) not in opt._cache and ("k", 0) in opt._cache and ("k", 2) in opt._cache       # oldest non-live one went
    for k in list(opt._cache):
        opt._cache[k]["live"] = True
    opt._cache[("k", "new")] = dict(live=True)
    opt._evict()                                                                                      # nothing evictable: it grows
    assert len(opt._cache) == opt.MAX_CACHE and all(e["live"] for e in opt._cache.values())
