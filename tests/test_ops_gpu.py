"""GPU parity tests, operator level: every HIP kernel (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): 1e-4 relative, fp32.  Inputs are seeded; sizes are small enough for
the oracle to finish in seconds but include ragged batches (B not a multiple of 64, B > 64).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hmvae_oracle as O
from helpers import GOLDEN, offsets, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    from hm_vae_amd import _lib
    _lib.require_gpu()
    _lib.load()
    return torch.device("cuda:0")


def soa_from_bct(x, dev):
    """(B, C, T) cpu -> device [C*T, bp] through the HIP transpose kernel."""
    from hm_vae_amd import ops
    B, Cn, T = x.shape
    bp = ops.pad_batch(B)
    src = x.contiguous().to(dev)
    out = torch.full((Cn * T, bp), float("nan"), device=dev)
    ops.aos_to_soa(src, out, B, 1, Cn * T, bp)
    return out, bp


def bct_from_soa(s, B, Cn, T):
    from hm_vae_amd import ops
    out = torch.empty(B, Cn * T, device=s.device)
    ops.soa_to_aos(s, out, B, 1, Cn * T, s.shape[1])
    return out.reshape(B, Cn, T).cpu()


def test_library_loaded_and_transposes(dev):
    from hm_vae_amd import ops, _lib
    assert os.path.exists(_lib.lib_path())
    for B, T, Cn in ((3, 5, 7), (70, 4, 144), (129, 2, 65)):
        x = torch.randn(B, T, Cn)
        bp = ops.pad_batch(B)
        s = torch.full((Cn * T, bp), float("nan"), device=dev)
        ops.aos_to_soa(x.to(dev), s, B, T, Cn, bp)
        ref = torch.zeros(Cn, T, bp)
        ref[:, :, :B] = x.permute(2, 1, 0)
        assert torch.equal(s.cpu().reshape(Cn, T, bp), ref)          # padded columns are exact zeros
        back = torch.empty(B, T, Cn, device=dev)
        ops.soa_to_aos(s, back, B, T, Cn, bp)
        assert torch.equal(back.cpu(), x)


def _level_neigh(level):
    from hm_vae_amd import topology
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(level):
        edges = topology.derive_pooling(edges)[2]
    return topology.find_neighbor(edges, 2)


def v4_variants(side, bp):
    """(companion side, columns per unit, K split) of every launch shape of the bf16-plane gather-GEMM (gemm_v4.hip) the
    problem supports: 64- and 128-row groups, 64 / 128 columns, unsplit and split three ways (last-arriver reduction)."""
    out = []
    for mtw in (1, 2):
        sd = side.v4_side(mtw)
        if sd is None:
            continue
        for nc in (64, 128):
            if bp % nc == 0:
                out.extend((sd, nc, sp) for sp in (1, 3))
    return out


def check_v4(side, dense, src, bp, bias, flags, ref, read_back, what):
    """Every v4 launch shape of `side` against the oracle's result `ref` (1e-4), plus the accumulate flag once."""
    from hm_vae_amd import ops, _lib
    var = v4_variants(side, bp)
    assert var, what
    packed = set()
    for sd, nc, sp in var:
        if id(sd) not in packed:
            ops.MultiPack(src.device).run([(sd, dense)])
            packed.add(id(sd))
        out = torch.full((read_back[0], bp), float("nan"), device=src.device)
        sd.run_v4(src, out, bp, bias, flags, 0.2, nc, sp)
        assert rel_err(read_back[1](out), ref) < TOL, (what, sd.g.mt, nc, sp)
        if sp > 1 and not (flags & _lib.FLAG_LEAKY):
            sd.run_v4(src, out, bp, bias, flags | _lib.FLAG_ACCUM, 0.2, nc, sp)
            assert rel_err(read_back[1](out), 2 * ref) < TOL, (what, "accumulate", sd.g.mt, nc, sp)
            sd.run_v4(src, out, bp, bias, flags, 0.2, nc, sp)           # counters were left at zero: a third launch agrees
            assert rel_err(read_back[1](out), ref) < TOL, (what, "relaunch", sd.g.mt, nc, sp)


CONV_CASES = [
    dict(level=0, cin=144, cout=288, K=15, s=2, T=64, B=5, mode="reflect", leaky=False),      # enc0 of len64
    dict(level=3, cin=336, cout=672, K=15, s=2, T=8, B=70, mode="reflect", leaky=False),       # enc3, pad ~ T
    dict(level=3, cin=672, cout=336, K=15, s=1, T=8, B=3, mode="reflect", leaky=True),         # dec0
    dict(level=0, cin=576, cout=144, K=15, s=1, T=64, B=2, mode="reflect", leaky=False),       # dec3 (6 rows/joint)
    dict(level=1, cin=168, cout=336, K=3, s=2, T=8, B=4, mode="reflect", leaky=False),         # len8 config, K=3
    dict(level=2, cin=36, cout=72, K=5, s=1, T=6, B=65, mode="zeros", leaky=True),
]


@pytest.mark.parametrize("T,B,s", [(22, 130, 2), (16, 64, 1)])
def test_gather_launch_shapes_agree(dev, T, B, s):
    """Every launch shape the autotuner may pick for the v2 gather-GEMM -- (column units, K split) and the position-per-wave
    shape (column units, 1, 4), incl. a last workgroup with fewer than four positions -- against the oracle's convolution,
    forward and backward-data."""
    from hm_vae_amd import ops
    neigh = _level_neigh(0)
    cin, cout, K = len(neigh) * 6, len(neigh) * 12, 15
    mask = O.conv_mask(neigh, cin, cout, K)
    g = torch.Generator().manual_seed(T)
    w = (torch.randn(cout, cin, K, generator=g) * mask / 10.0).requires_grad_(True)
    b = (0.1 * torch.randn(cout, generator=g)).requires_grad_(True)
    x = torch.randn(B, cin, T, generator=g, requires_grad=True)
    y = O.skel_conv(x, w, mask, b, s, 7, "reflect")
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    T_out = y.shape[2]
    op = ops.conv_gemm(dev, neigh, cin, cout, K, s, 7, "reflect", T)
    op.pack(w.detach().to(dev))
    xs, bp = soa_from_bct(x.detach(), dev)
    gys, _ = soa_from_bct(gy, dev)
    shapes_f, shapes_b = op.fwd.v2_candidates(bp), op.bwd.v2_candidates(bp)
    assert any(len(c) == 3 for c in shapes_f) and any(len(c) == 3 for c in shapes_b)
    for cand in shapes_f:
        op.fwd.tune = cand
        ys = torch.full((cout * T_out, bp), float("nan"), device=dev)
        op.fwd.run_v2(xs, ys, bp, b.detach().to(dev), 0, 0.2)
        assert rel_err(bct_from_soa(ys, B, cout, T_out).numpy(), y.detach().numpy()) < TOL, cand
    for cand in shapes_b:
        op.bwd.tune = cand
        dxs = torch.full((cin * T, bp), float("nan"), device=dev)
        op.bwd.run_v2(gys, dxs, bp, None, 0, 0.2)
        assert rel_err(bct_from_soa(dxs, B, cin, T).numpy(), x.grad.numpy()) < TOL, cand


@pytest.mark.parametrize("case", CONV_CASES)
def test_skeleton_conv_fwd_bwd(dev, case):
    from hm_vae_amd import ops, _lib
    torch.manual_seed(17)
    neigh = _level_neigh(case["level"])
    cin, cout, K, s, T, B = (case[k] for k in ("cin", "cout", "K", "s", "T", "B"))
    pad = (K - 1) // 2
    mask = O.conv_mask(neigh, cin, cout, K)
    fan = mask[:, :, 0].sum(1, keepdim=True)[:, :, None] * K
    w = (torch.randn(cout, cin, K) * mask / fan.sqrt()).requires_grad_(True)
    b = (0.1 * torch.randn(cout)).requires_grad_(True)
    x = torch.randn(B, cin, T, requires_grad=True)
    y = O.skel_conv(x, w, mask, b, s, pad, "reflect" if case["mode"] == "reflect" else "constant")
    if case["leaky"]:
        y = F.leaky_relu(y, 0.2)
    gy = torch.randn_like(y)
    y.backward(gy)
    T_out = y.shape[2]

    op = ops.conv_gemm(dev, neigh, cin, cout, K, s, pad, case["mode"], T)
    wd = w.detach().to(dev)
    # poison masked-out weights: the kernels must apply the mask themselves (skeleton.py:96)
    wd_poison = wd + (1 - mask.to(dev)) * 3.0
    op.pack(wd_poison)
    xs, bp = soa_from_bct(x.detach(), dev)
    ys = torch.full((cout * T_out, bp), float("nan"), device=dev)
    op.forward(xs, ys, bp, bias=b.detach().to(dev), flags=_lib.FLAG_LEAKY if case["leaky"] else 0)
    assert rel_err(bct_from_soa(ys, B, cout, T_out).numpy(), y.detach().numpy()) < TOL

    g_pre = gy if not case["leaky"] else gy * torch.where(y.detach() > 0, torch.ones_like(gy), torch.full_like(gy, 0.2))
    gys, _ = soa_from_bct(g_pre, dev)
    if case["leaky"]:                      # exercise the leaky backward kernel too
        gy_raw, _ = soa_from_bct(gy, dev)
        tmp = torch.empty_like(gy_raw)
        ops.leaky_bwd(gy_raw, ys, tmp, tmp.numel())
        assert rel_err(tmp.cpu().numpy(), gys.cpu().numpy()) < 1e-6
    dxs = torch.full((cin * T, bp), float("nan"), device=dev)
    op.backward_data(gys, dxs, bp)
    assert rel_err(bct_from_soa(dxs, B, cin, T).numpy(), x.grad.numpy()) < TOL
    # accumulate flag
    op.backward_data(gys, dxs, bp, accumulate=True)
    assert rel_err(bct_from_soa(dxs, B, cin, T).numpy(), 2 * x.grad.numpy()) < TOL

    # HMVAE_FLAG_LEAKY_GRAD: the LeakyReLU backward of the layer in front rides in the data-gradient launch's epilogue -- against
    # the plain launch followed by hmvae_leaky_bwd, bit for bit; with a row map that exempts every third channel; accumulating
    act = torch.randn(cin * T, bp, device=dev)
    act_base = np.where(np.arange(cin) % 3 == 2, -1, np.arange(cin) * T).astype(np.int64)
    exempt = torch.from_numpy(np.repeat(act_base < 0, T)).to(dev)[:, None]
    plain = torch.empty_like(dxs)
    op.backward_data(gys, plain, bp)
    want = torch.empty_like(plain)
    ops.leaky_bwd(plain, act, want, want.numel())
    want_map = torch.where(exempt, plain, want)
    sides = [(op.bwd, lambda sd, o, fl: sd.run_v2(gys, o, bp, None, fl, 0.2))]
    for sd4, nc, sp in v4_variants(op.bwd, bp):
        ops.MultiPack(dev).run([(sd4, wd_poison)])
        sides.append((sd4, lambda sd, o, fl, nc=nc, sp=sp: sd.run_v4(gys, o, bp, None, fl, 0.2, nc, sp), (nc, sp)))
    sides[0] = sides[0] + ("v2",)
    for sd, launch, shape in sides:
        ref_plain = torch.empty_like(dxs)
        launch(sd, ref_plain, 0)
        ops.leaky_bwd(ref_plain, act, want, want.numel())
        for tab, expect in ((None, want), (sd._act_table(act_base), torch.where(exempt, ref_plain, want))):
            got = torch.full_like(dxs, float("nan"))
            sd._act_now = (act, tab)
            launch(sd, got, _lib.FLAG_LEAKY_GRAD)
            assert torch.equal(got, expect), (sd.g.mt, tab is None, shape)
            launch(sd, got, _lib.FLAG_LEAKY_GRAD | _lib.FLAG_ACCUM)             # (g + got) * f with got = g * f
            f = torch.where(act > 0, 1.0, 0.2) if tab is None else torch.where(exempt | (act > 0), 1.0, 0.2)
            assert rel_err(got.cpu().numpy(), ((ref_plain + expect) * f).cpu().numpy()) < 1e-6, (sd.g.mt, tab is None)
            sd._act_now = None
    fused = op.backward_data(gys, dxs, bp, act=act, act_base=act_base)            # the public call (the autotuner's pick)
    assert fused or op.bwd._tuned.get(bp, (0,))[0] == "v3"
    if fused:
        assert rel_err(dxs.cpu().numpy(), want_map.cpu().numpy()) < TOL

    # the bf16-plane kernel (v4) in every launch shape, forward and backward-data, against the same reference
    lk = _lib.FLAG_LEAKY if case["leaky"] else 0
    check_v4(op.fwd, wd_poison, xs, bp, b.detach().to(dev), lk, y.detach().numpy(),
             (cout * T_out, lambda o: bct_from_soa(o, B, cout, T_out).numpy()), "fwd")
    check_v4(op.bwd, wd_poison, gys, bp, None, 0, x.grad.numpy(),
             (cin * T, lambda o: bct_from_soa(o, B, cin, T).numpy()), "bwd")

    # the window kernel (v3) in every workgroup shape, forward and backward-data, against the same reference
    n_v3 = 0
    for chunk, ts in ((8, 1), (4, 2), (2, 4), (1, 8)):
        if op.fwd.window_plan(chunk, ts) is not None:
            y3 = torch.full((cout * T_out, bp), float("nan"), device=dev)
            op.fwd.run_v3(xs, y3, bp, b.detach().to(dev), _lib.FLAG_LEAKY if case["leaky"] else 0, 0.2, chunk, ts)
            assert rel_err(bct_from_soa(y3, B, cout, T_out).numpy(), y.detach().numpy()) < TOL, (chunk, ts)
            n_v3 += 1
        if op.bwd.window_plan(chunk, ts) is not None:
            dx3 = torch.full((cin * T, bp), float("nan"), device=dev)
            op.bwd.run_v3(gys, dx3, bp, None, 0, 0.2, chunk, ts)
            assert rel_err(bct_from_soa(dx3, B, cin, T).numpy(), x.grad.numpy()) < TOL, (chunk, ts)
            op.bwd.run_v3(gys, dx3, bp, None, _lib.FLAG_ACCUM, 0.2, chunk, ts)
            assert rel_err(bct_from_soa(dx3, B, cin, T).numpy(), 2 * x.grad.numpy()) < TOL, (chunk, ts)
            n_v3 += 1
    assert n_v3 > 0 or op.folded          # composite relations (virtual taps) run on the v2 kernel only

    dw = torch.zeros(cout, cin, K, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    op.backward_weight(gys, xs, dw, bp, slab)
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < TOL
    assert float((dw.cpu() * (1 - mask)).abs().max()) == 0.0         # masked entries stay exactly zero
    # bias gradient = row sums over (t, batch)
    rows = (torch.arange(cout, dtype=torch.int32) * T_out).to(dev)
    db = torch.empty(cout, device=dev)
    ops.row_sums(gys, db, rows, cout, T_out, 1, bp)
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < TOL


@pytest.mark.parametrize("level,ch,co,Tc,B", [(3, 96, 48, 4, 3), (2, 48, 24, 8, 70), (0, 24, 6, 32, 2)])
def test_conv_behind_upsample_unpool(dev, level, ch, co, Tc, B):
    """Decoder layer front: nn.Upsample(x2 linear) + SkeletonUnpool folded into the convolution's GEMMs (dec0 / dec1 /
    dec3 shapes): forward and backward-data on the coarse tensor, weight gradient on the conv's own input."""
    from hm_vae_amd import ops, topology, _lib
    torch.manual_seed(11 + level)
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(level):
        edges = topology.derive_pooling(edges)[2]
    pooling = topology.derive_pooling(edges)[1] if level < 3 else [[i] for i in range(len(edges))]
    neigh = topology.find_neighbor(edges, 2)
    n_f, n_c = len(neigh), len(pooling)
    src = np.zeros(n_f, dtype=np.int64)
    for i, grp in enumerate(pooling):
        for j in grp:
            src[j] = i
    K, pad, T = 15, 7, 2 * Tc
    cin, cout = n_f * ch, n_f * co
    mask = O.conv_mask(neigh, cin, cout, K)
    fan = mask[:, :, 0].sum(1, keepdim=True)[:, :, None] * K
    w = (torch.randn(cout, cin, K) * mask / fan.sqrt()).requires_grad_(True)
    b = (0.1 * torch.randn(cout)).requires_grad_(True)
    xc = torch.randn(B, n_c * ch, Tc, requires_grad=True)
    idx = torch.as_tensor((src[:, None] * ch + np.arange(ch)[None, :]).reshape(-1))
    U = F.interpolate(xc[:, idx, :], scale_factor=2, mode="linear", align_corners=False)
    y = F.leaky_relu(O.skel_conv(U, w, mask, b, 1, pad, "reflect"), 0.2)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = ops.conv_gemm(dev, neigh, cin, cout, K, 1, pad, "reflect", T, need_dx=True, upsample=True, unpool_src=src)
    assert op.folded and op.fused_front
    wd = w.detach().to(dev)
    op.pack(wd + (1 - mask.to(dev)) * 3.0)            # poisoned masked-out weights must not leak through merged channels
    xs, bp = soa_from_bct(xc.detach(), dev)
    ys = torch.full((cout * T, bp), float("nan"), device=dev)
    op.forward(xs, ys, bp, bias=b.detach().to(dev), flags=_lib.FLAG_LEAKY)
    assert rel_err(bct_from_soa(ys, B, cout, T).numpy(), y.detach().numpy()) < TOL
    g_pre = gy * torch.where(y.detach() > 0, torch.ones_like(gy), torch.full_like(gy, 0.2))
    gys, _ = soa_from_bct(g_pre, dev)
    dxs = torch.full((n_c * ch * Tc, bp), float("nan"), device=dev)
    op.backward_data(gys, dxs, bp)
    assert rel_err(bct_from_soa(dxs, B, n_c * ch, Tc).numpy(), xc.grad.numpy()) < TOL
    wpo = wd + (1 - mask.to(dev)) * 3.0
    check_v4(op.fwd, wpo, xs, bp, b.detach().to(dev), _lib.FLAG_LEAKY, y.detach().numpy(),
             (cout * T, lambda o: bct_from_soa(o, B, cout, T).numpy()), "fwd (upsample + unpool folded)")
    check_v4(op.bwd, wpo, gys, bp, None, 0, xc.grad.numpy(),
             (n_c * ch * Tc, lambda o: bct_from_soa(o, B, n_c * ch, Tc).numpy()), "bwd (upsample + unpool folded)")
    us, _ = soa_from_bct(U.detach(), dev)
    dw = torch.zeros(cout, cin, K, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    op.backward_weight(gys, us, dw, bp, slab)
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < TOL
    assert float((dw.cpu() * (1 - mask)).abs().max()) == 0.0


@pytest.mark.parametrize("level,ci,co,T,B", [(0, 6, 12, 64, 3), (2, 24, 48, 16, 70), (3, 48, 96, 8, 2)])
def test_conv_followed_by_pool(dev, level, ci, co, T, B):
    """Encoder layer: SkeletonPool (mean of <= 2 edges) + LeakyReLU folded into the stride-2 convolution (enc0 / enc2 /
    enc3 shapes): forward gives the pooled activated tensor, backward-data and the deduplicated weight gradient take the
    gradient w.r.t. the pooled pre-activation, the bias gradient is a scaled row sum of it."""
    from hm_vae_amd import ops, topology, _lib
    torch.manual_seed(23 + level)
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(level):
        edges = topology.derive_pooling(edges)[2]
    pooling = topology.derive_pooling(edges, last_pool=(level == 3))[1]
    neigh = topology.find_neighbor(edges, 2)
    n_f, n_p = len(neigh), len(pooling)
    K, pad = 15, 7
    cin, cout = n_f * ci, n_f * co
    mask = O.conv_mask(neigh, cin, cout, K)
    fan = mask[:, :, 0].sum(1, keepdim=True)[:, :, None] * K
    w = (torch.randn(cout, cin, K) * mask / fan.sqrt()).requires_grad_(True)
    b = (0.1 * torch.randn(cout)).requires_grad_(True)
    x = torch.randn(B, cin, T, requires_grad=True)
    y = O.skel_conv(x, w, mask, b, 2, pad, "reflect")
    T_out = y.shape[2]
    Pm = torch.zeros(n_p * co, cout)
    for i, grp in enumerate(pooling):
        for j in grp:
            for c in range(co):
                Pm[i * co + c, j * co + c] = 1.0 / len(grp)
    pre = torch.einsum("pr,brt->bpt", Pm, y)
    pre.retain_grad()
    a = F.leaky_relu(pre, 0.2)
    ga = torch.randn_like(a)
    a.backward(ga)
    op = ops.conv_gemm(dev, neigh, cin, cout, K, 2, pad, "reflect", T, need_dx=True, pool_groups=pooling)
    assert op.folded and op.fused_pool and op.wg_pooled_dz
    wd = w.detach().to(dev)
    op.pack(wd + (1 - mask.to(dev)) * 3.0)            # poisoned masked-out weights must not leak through merged rows
    xs, bp = soa_from_bct(x.detach(), dev)
    As = torch.full((n_p * co * T_out, bp), float("nan"), device=dev)
    op.forward(xs, As, bp, bias=b.detach().to(dev), flags=_lib.FLAG_LEAKY)
    assert rel_err(bct_from_soa(As, B, n_p * co, T_out).numpy(), a.detach().numpy()) < TOL
    dps, _ = soa_from_bct(pre.grad, dev)
    dxs = torch.full((cin * T, bp), float("nan"), device=dev)
    op.backward_data(dps, dxs, bp)
    assert rel_err(bct_from_soa(dxs, B, cin, T).numpy(), x.grad.numpy()) < TOL
    wpo = wd + (1 - mask.to(dev)) * 3.0
    check_v4(op.fwd, wpo, xs, bp, b.detach().to(dev), _lib.FLAG_LEAKY, a.detach().numpy(),
             (n_p * co * T_out, lambda o: bct_from_soa(o, B, n_p * co, T_out).numpy()), "fwd (pool folded)")
    check_v4(op.bwd, wpo, dps, bp, None, 0, x.grad.numpy(),
             (cin * T, lambda o: bct_from_soa(o, B, cin, T).numpy()), "bwd (pool folded)")
    dw = torch.zeros(cout, cin, K, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    op.backward_weight(dps, xs, dw, bp, slab)
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < TOL
    assert float((dw.cpu() * (1 - mask)).abs().max()) == 0.0
    rows = torch.as_tensor(op.pool_dst_base[op.pool_rmap].astype(np.int32)).to(dev)
    scale = torch.as_tensor(op.pool_scale[op.pool_rmap].astype(np.float32)).to(dev)
    db = torch.empty(cout, device=dev)
    ops.row_sums_scaled(dps, db, rows, scale, cout, T_out, 1, bp)
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < TOL


@pytest.mark.parametrize("B,k_edges,n_in,n_out", [(5, 7, 384, 48), (70, 14, 12, 384), (3, 9, 384, 24)])
def test_linear_per_edge(dev, B, k_edges, n_in, n_out):
    from hm_vae_amd import ops
    torch.manual_seed(3)
    w = (torch.randn(n_out, n_in) / n_in ** 0.5).requires_grad_(True)
    b = (0.1 * torch.randn(n_out)).requires_grad_(True)
    x = torch.randn(B, k_edges, n_in, requires_grad=True)
    y = F.linear(x, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)
    op = ops.linear_gemm(dev, n_out, n_in, k_edges, n_in, n_out)
    op.pack(w.detach().to(dev).reshape(n_out, n_in, 1))
    xs, bp = soa_from_bct(x.detach().reshape(B, k_edges * n_in, 1), dev)
    ys = torch.full((k_edges * n_out, bp), float("nan"), device=dev)
    op.forward(xs, ys, bp, bias=b.detach().to(dev))
    assert rel_err(bct_from_soa(ys, B, k_edges * n_out, 1).reshape(B, k_edges, n_out).numpy(), y.detach().numpy()) < TOL
    gys, _ = soa_from_bct(gy.reshape(B, k_edges * n_out, 1), dev)
    dxs = torch.full((k_edges * n_in, bp), float("nan"), device=dev)
    op.backward_data(gys, dxs, bp)
    assert rel_err(bct_from_soa(dxs, B, k_edges * n_in, 1).reshape(B, k_edges, n_in).numpy(), x.grad.numpy()) < TOL
    dw = torch.zeros(n_out, n_in, 1, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    op.backward_weight(gys, xs, dw, bp, slab)
    assert rel_err(dw.cpu().numpy()[:, :, 0], w.grad.numpy()) < TOL
    rows = torch.arange(n_out, dtype=torch.int32).to(dev)
    db = torch.empty(n_out, device=dev)
    ops.row_sums(gys, db, rows, n_out, k_edges, n_out, bp)
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < TOL


@pytest.mark.parametrize("level,ch,T,B", [(0, 12, 32, 5), (1, 24, 16, 70), (3, 96, 4, 3)])
def test_pool_and_unpool(dev, level, ch, T, B):
    from hm_vae_amd import ops, topology
    torch.manual_seed(level)
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(level):
        edges = topology.derive_pooling(edges)[2]
    _, pooling, _ = topology.derive_pooling(edges, last_pool=(level == 3))
    n_in, n_out = len(edges), len(pooling)
    x = torch.randn(B, n_in * ch, T, requires_grad=True)
    Wp = O.pool_matrix(pooling, n_in, ch)
    y = F.leaky_relu(torch.matmul(Wp, x), 0.2)
    gy = torch.randn_like(y)
    y.backward(gy)
    pool = ops.PoolOp(dev, pooling, n_in, ch)
    xs, bp = soa_from_bct(x.detach(), dev)
    ys = torch.full((n_out * ch * T, bp), float("nan"), device=dev)
    pool.forward(xs, ys, T, bp)
    assert rel_err(bct_from_soa(ys, B, n_out * ch, T).numpy(), y.detach().numpy()) < 1e-6
    gys, _ = soa_from_bct(gy, dev)
    dxs = torch.full((n_in * ch * T, bp), float("nan"), device=dev)
    pool.backward(gys, ys, dxs, T, bp)
    assert rel_err(bct_from_soa(dxs, B, n_in * ch, T).numpy(), x.grad.numpy()) < 1e-6

    # upsample + unpool (decoder direction): coarse (n_out edges) -> fine (n_in edges)
    for up in (True, False):
        z = torch.randn(B, n_out * ch, T, requires_grad=True)
        Wu = O.unpool_matrix(pooling, ch)
        zz = F.interpolate(z, scale_factor=2, mode="linear", align_corners=False) if up else z
        u = torch.matmul(Wu, zz)
        gu = torch.randn_like(u)
        u.backward(gu)
        unp = ops.UnpoolOp(dev, pooling, ch, up)
        zs, _ = soa_from_bct(z.detach(), dev)
        T2 = 2 * T if up else T
        us = torch.full((n_in * ch * T2, bp), float("nan"), device=dev)
        unp.forward(zs, us, T, bp)
        assert rel_err(bct_from_soa(us, B, n_in * ch, T2).numpy(), u.detach().numpy()) < 1e-6
        gus, _ = soa_from_bct(gu, dev)
        dzs = torch.full((n_out * ch * T, bp), float("nan"), device=dev)
        unp.backward(gus, dzs, T, bp)
        assert rel_err(bct_from_soa(dzs, B, n_out * ch, T).numpy(), z.grad.numpy()) < 1e-5


@pytest.mark.parametrize("B,k_edges,d", [(5, 7, 24), (70, 14, 12), (300, 9, 6)])
def test_reparam_kl(dev, B, k_edges, d):
    from hm_vae_amd import ops
    torch.manual_seed(B)
    dist = (0.5 * torch.randn(B, k_edges, 2 * d)).requires_grad_(True)
    eps = torch.randn(B * k_edges, d)
    mu = dist[:, :, :d].reshape(-1, d)
    lv = dist[:, :, d:].reshape(-1, d)
    z = eps * torch.exp(0.5 * lv) + mu
    kl = O.OracleVAE.kl(lv, mu)
    gz = torch.randn_like(z)
    kl_w = 0.37
    ((z * gz).sum() + kl_w * kl).backward()
    ds, bp = soa_from_bct(dist.detach().reshape(B, k_edges * 2 * d, 1), dev)
    zs = torch.full((k_edges * d, bp), float("nan"), device=dev)
    nblk = ops.kl_partial_blocks(k_edges, bp, d)
    part = torch.full((nblk,), float("nan"), device=dev)
    ops.reparam_kl_fwd(ds, eps.to(dev), zs, part, k_edges, d, B, bp)
    assert rel_err(bct_from_soa(zs, B, k_edges * d, 1).reshape(B * k_edges, d).numpy(), z.detach().numpy()) < 1e-5
    out = torch.empty(1, device=dev)
    ops.finalize_sum(part, nblk, 1, 0, 1.0 / (B * k_edges), out)
    assert abs(float(out) - float(kl)) < 1e-5 * max(1.0, abs(float(kl)))
    gzs, _ = soa_from_bct(gz.reshape(B, k_edges * d, 1), dev)
    dd = torch.full((k_edges * 2 * d, bp), float("nan"), device=dev)
    ops.reparam_kl_bwd(ds, eps.to(dev), gzs, dd, k_edges, d, B, bp, True, kl_w / (B * k_edges))
    assert rel_err(bct_from_soa(dd, B, k_edges * 2 * d, 1).reshape(B, k_edges, 2 * d).numpy(), dist.grad.numpy()) < TOL
    assert float(dd[:, B:].abs().max()) == 0.0 if bp > B else True


@pytest.mark.parametrize("B,k_edges,n_in,d,n_out,y_es,y_off", [(256, 7, 384, 24, 384, 384, 0), (70, 7, 384, 24, 384, 384, 0), (5, 14, 384, 12, 384, 1152, 768),
                                                                (130, 3, 64, 8, 128, 140, 8)])
def test_latent_bridge_one_launch(dev, B, k_edges, n_in, d, n_out, y_es, y_off):
    """hmvae_latent_bridge_fwd / _bwd: the encoder head (nn.Linear per edge, seq_two_hier_sa_vae.py:162-170), reparametrize +
    kl_loss (:359-391) and the decoder head (:267-272) of one hierarchy level as one launch per direction, against autograd
    on the oracle's formulation -- outputs written into a buffer whose edges are y_es rows apart from row y_off (the concat
    buffer of the last decoder layer), the input gradient plain, accumulated, and with the LeakyReLU backward of the layer in
    front riding along; padded batch columns stay zero."""
    from hm_vae_amd import ops
    torch.manual_seed(B + d)
    x = torch.randn(B, k_edges, n_in, requires_grad=True)
    w1 = (torch.randn(2 * d, n_in) / n_in ** 0.5).requires_grad_(True)
    b1 = (0.1 * torch.randn(2 * d)).requires_grad_(True)
    w2 = (torch.randn(n_out, d) / d ** 0.5).requires_grad_(True)
    b2 = (0.1 * torch.randn(n_out)).requires_grad_(True)
    eps = torch.randn(B * k_edges, d)
    dist = F.linear(x, w1, b1)
    mu, lv = dist[:, :, :d].reshape(-1, d), dist[:, :, d:].reshape(-1, d)
    z = eps * torch.exp(0.5 * lv) + mu
    kl = O.OracleVAE.kl(lv, mu)
    y = F.linear(z.reshape(B, k_edges, d), w2, b2)
    gy = torch.randn_like(y)
    kl_w = 0.41
    dist.retain_grad()
    ((y * gy).sum() + kl_w * kl).backward()
    xs, bp = soa_from_bct(x.detach().reshape(B, k_edges * n_in, 1), dev)
    ds = torch.full((k_edges * 2 * d, bp), float("nan"), device=dev)
    zs = torch.full((k_edges * d, bp), float("nan"), device=dev)
    ys = torch.full((k_edges * y_es, bp), float("nan"), device=dev)
    part = torch.full((k_edges * (bp // 16),), float("nan"), device=dev)
    W1, B1, W2, B2, E = (t.detach().to(dev).contiguous() for t in (w1, b1, w2, b2, eps))
    ops.latent_bridge_fwd(xs, W1, B1, E, W2, B2, ds, zs, ys, part, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, True)
    assert rel_err(bct_from_soa(ds, B, k_edges * 2 * d, 1).reshape(B, k_edges, 2 * d).numpy(), dist.detach().numpy()) < 1e-5
    assert rel_err(bct_from_soa(zs, B, k_edges * d, 1).reshape(B * k_edges, d).numpy(), z.detach().numpy()) < 1e-5
    yv = ys.reshape(k_edges, y_es, bp)[:, y_off:y_off + n_out, :B].permute(2, 0, 1).cpu()
    assert rel_err(yv.numpy(), y.detach().numpy()) < 1e-5
    assert torch.isnan(ys.reshape(k_edges, y_es, bp)[:, :y_off]).all() and torch.isnan(ys.reshape(k_edges, y_es, bp)[:, y_off + n_out:]).all()
    out = torch.empty(1, device=dev)
    ops.finalize_sum(part, part.numel(), 1, 0, 1.0 / (B * k_edges), out)
    assert abs(float(out) - float(kl)) < 1e-5 * max(1.0, abs(float(kl)))
    if bp > B:
        assert float(zs[:, B:].abs().max()) == 0.0
    # no sampling: z = mu
    ops.latent_bridge_fwd(xs, W1, B1, None, W2, B2, ds, zs, ys, part, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, False)
    assert rel_err(bct_from_soa(zs, B, k_edges * d, 1).reshape(B * k_edges, d).numpy(), mu.detach().numpy()) < 1e-5
    ops.latent_bridge_fwd(xs, W1, B1, E, W2, B2, ds, zs, ys, part, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, True)
    # backward
    gfull = torch.zeros(B, k_edges, y_es)
    gfull[:, :, y_off:y_off + n_out] = gy
    gys, _ = soa_from_bct(gfull.reshape(B, k_edges * y_es, 1), dev)
    dd = torch.full((k_edges * 2 * d, bp), float("nan"), device=dev)
    dxs = torch.full((k_edges * n_in, bp), float("nan"), device=dev)
    scale = kl_w / (B * k_edges)
    ops.latent_bridge_bwd(gys, W2, ds, E, W1, dd, dxs, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, scale)
    assert rel_err(bct_from_soa(dd, B, k_edges * 2 * d, 1).reshape(B, k_edges, 2 * d).numpy(), dist.grad.numpy()) < TOL
    want = x.grad.numpy()
    assert rel_err(bct_from_soa(dxs, B, k_edges * n_in, 1).reshape(B, k_edges, n_in).numpy(), want) < TOL
    if bp > B:
        assert float(dd[:, B:].abs().max()) == 0.0 and float(dxs[:, B:].abs().max()) == 0.0
    ops.latent_bridge_bwd(gys, W2, ds, E, W1, dd, dxs, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, scale, accumulate=True)
    assert rel_err(bct_from_soa(dxs, B, k_edges * n_in, 1).reshape(B, k_edges, n_in).numpy(), 2 * want) < TOL
    act = torch.randn(k_edges * n_in, bp, device=dev)
    ops.latent_bridge_bwd(gys, W2, ds, E, W1, dd, dxs, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, scale, act=act)
    f = torch.where(act > 0, 1.0, 0.2)[:, :B].reshape(k_edges, n_in, B).permute(2, 0, 1).cpu().numpy()
    assert rel_err(bct_from_soa(dxs, B, k_edges * n_in, 1).reshape(B, k_edges, n_in).numpy(), want * f) < TOL
    # dx NULL: stop at the head's pre-activation gradient
    dd2 = torch.full_like(dd, float("nan"))
    ops.latent_bridge_bwd(gys, W2, ds, E, None, dd2, None, k_edges, n_in, d, n_out, n_in, y_es, y_off, B, bp, scale)
    assert torch.equal(dd2, dd)


def test_randn_kernel_statistics(dev):
    """hmvae_randn: Philox4x32-10 + Box-Muller with the launch counter on the device -- moments of a standard normal, a fresh
    draw at every launch (also when the launch is replayed from a hipGraph), the same stream again from the same seed and
    counter, nothing written past n."""
    from hm_vae_amd import ops
    n = 1 << 20
    out = torch.full((n + 8,), 7.0, device=dev)
    state = torch.zeros(2, dtype=torch.int64, device=dev)
    ops.randn_(out[:n], 1234, state)
    a = out[:n].double().cpu().numpy()
    assert (out[n:] == 7.0).all() and state.tolist() == [1, 0]
    assert abs(a.mean()) < 5e-3 and abs(a.var() - 1) < 1e-2 and abs((a ** 3).mean()) < 2e-2 and abs((a ** 4).mean() - 3) < 5e-2
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5e-3 and abs(np.corrcoef(a[:-4], a[4:])[0, 1]) < 5e-3
    assert np.abs(a).max() < 6.5 and len(np.unique(a)) > 0.9 * n
    ops.randn_(out[:n], 1234, state)
    b = out[:n].double().cpu().numpy()
    assert state.tolist() == [2, 0] and abs(np.corrcoef(a, b)[0, 1]) < 5e-3
    state.zero_()
    ops.randn_(out[:n], 1234, state)
    assert np.array_equal(out[:n].double().cpu().numpy(), a)                 # same seed, same counter: the same stream
    ops.randn_(out[:n], 99, state.zero_())
    assert abs(np.corrcoef(a, out[:n].double().cpu().numpy())[0, 1]) < 5e-3  # another seed
    small = torch.zeros(10, device=dev)
    ops.randn_(small[:7], 5, state)
    assert (small[7:] == 0).all() and (small[:7] != 0).all()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.randn_(out[:n], 1234, state)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        ops.randn_(out[:n], 1234, state)
    g.replay()
    c1 = out[:n].clone()
    g.replay()
    torch.cuda.synchronize()
    assert abs(np.corrcoef(c1.cpu().numpy(), out[:n].cpu().numpy())[0, 1]) < 5e-3


def test_fk_against_reference_golden(dev):
    """AoS kernels vs vectors produced by the unmodified reference (tests/golden/fk_rot6d.npz)."""
    from hm_vae_amd import ops
    g = np.load(os.path.join(GOLDEN, "fk_rot6d.npz"))
    off = torch.from_numpy(offsets()).to(dev)
    par = O.SMPL24_PARENTS
    r6 = torch.from_numpy(g["r6"]).to(dev)
    n = r6.shape[0]
    R = torch.full((n, 24, 9), float("nan"), device=dev)
    P = torch.full((n, 24, 3), float("nan"), device=dev)
    ops.fk_fwd(r6, 6, off, 0, par, n, R, P)
    assert np.abs(R.cpu().numpy().reshape(n, 24, 3, 3) - g["R"]).max() < 1e-6      # incl. degenerate rows: bitwise-level
    assert np.abs(P.cpu().numpy() - g["P"]).max() < 1e-6
    # rotation-matrix input path and the rest pose known answer
    P2 = torch.empty_like(P)
    ops.fk_fwd(R, 9, off, 0, par, n, None, P2)
    assert np.abs(P2.cpu().numpy() - g["P"]).max() < 1e-6
    ident = torch.eye(3, device=dev).reshape(1, 1, 9).repeat(1, 24, 1).contiguous()
    P3 = torch.empty(1, 24, 3, device=dev)
    ops.fk_fwd(ident, 9, off, 0, par, 1, None, P3)
    rest = np.load(os.path.join(GOLDEN, "rest_pose_coord.npy"))
    assert np.abs(P3.cpu().numpy()[0] - rest).max() < 1e-6
    # backward: d(sum wR*R + wP*P)/d r6
    d6 = torch.full((n, 24, 6), float("nan"), device=dev)
    ops.fk_bwd(r6, 6, off, 0, par, n, torch.from_numpy(g["wR"]).reshape(n, 24, 9).contiguous().to(dev),
               torch.from_numpy(g["wP"]).to(dev), d6)
    d6 = d6.cpu().numpy()
    assert np.isfinite(d6).all()
    for i in range(n):
        assert rel_err(d6[i], g["d6"][i]) < TOL, i
    dR = torch.full((n, 24, 9), float("nan"), device=dev)
    ops.fk_bwd(R, 9, off, 0, par, n, None, torch.from_numpy(g["wP"]).to(dev), dR)
    assert rel_err(dR.cpu().numpy().reshape(n, 24, 3, 3), g["dR_fkonly"]) < TOL
    # per-frame offsets
    offs = off[None].repeat(n, 1, 1) * torch.linspace(0.5, 1.5, n, device=dev)[:, None, None]
    P4 = torch.empty_like(P)
    ops.fk_fwd(r6, 6, offs.contiguous(), 72, par, n, None, P4)
    ref = torch.stack([O.fk_positions(torch.from_numpy(g["r6"][i:i + 1]), par, offs[i].cpu()) for i in range(n)])[:, 0]
    assert rel_err(P4.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("B,T", [(3, 8), (70, 5)])
def test_decode_tail_loss_and_fwd(dev, B, T):
    from hm_vae_amd import ops
    torch.manual_seed(B)
    off_np = offsets()
    off = torch.from_numpy(off_np)
    dec = torch.randn(B, 144, T, requires_grad=True)
    tgt = O.synthetic_batch(B, T, seed=5)
    g6, gR = tgt[0], tgt[1]
    gP = O.fk_positions(gR.reshape(B * T, 24, 3, 3), O.SMPL24_PARENTS, off).reshape(B, T, 72)
    d6 = dec.transpose(1, 2).reshape(B * T, 24, 6)
    R = O.rot6d_to_matrix(d6)
    Pp = O.fk_positions(R, O.SMPL24_PARENTS, off)
    l6 = ((d6.reshape(B, T, -1) - g6) ** 2).mean()
    lR = ((R.reshape(B, T, -1) - gR) ** 2).mean()
    lP = ((Pp.reshape(B, T, -1) - gP) ** 2).mean()
    w6, wR, wP = 1.0, 1.0, 10.0
    (w6 * l6 + wR * lR + wP * lP).backward()
    bp = ops.pad_batch(B)
    decs, _ = soa_from_bct(dec.detach(), dev)

    def tgt_soa(t, Cn):
        s = torch.empty(Cn * T, bp, device=dev)
        ops.aos_to_soa(t.contiguous().to(dev), s, B, T, Cn, bp)
        return s

    g6s, gRs, gPs = tgt_soa(g6, 144), tgt_soa(gR, 216), tgt_soa(gP, 72)
    dd = torch.full((144 * T, bp), float("nan"), device=dev)
    part = torch.full((ops.decode_tail_loss_blocks(T, B, bp), 3), float("nan"), device=dev)
    offd = off.to(dev)
    nb = ops.decode_tail_loss(decs, g6s, gRs, gPs, offd, T, B, bp, w6 / (B * T * 144), wR / (B * T * 216),
                              wP / (B * T * 72), dd, part)
    assert nb == part.shape[0]
    sums = part.double().sum(0).cpu().numpy()
    assert abs(sums[0] / (B * T * 144) - float(l6)) < TOL * float(l6)
    assert abs(sums[1] / (B * T * 216) - float(lR)) < TOL * float(lR)
    assert abs(sums[2] / (B * T * 72) - float(lP)) < TOL * float(lP)
    assert rel_err(bct_from_soa(dd, B, 144, T).numpy(), dec.grad.numpy()) < TOL
    if bp > B:
        assert float(dd[:, B:].abs().max()) == 0.0
    # inference tail: SoA decoder output -> AoS 6D / R / pos
    c6o = torch.empty(B, T, 24, 6, device=dev)
    Ro = torch.empty(B, T, 24, 3, 3, device=dev)
    Po = torch.empty(B, T, 24, 3, device=dev)
    ops.decode_tail_fwd(decs, offd, T, B, bp, c6o, Ro, Po)
    assert rel_err(c6o.cpu().numpy().reshape(B * T, 24, 6), d6.detach().numpy()) < 1e-6
    assert rel_err(Ro.cpu().numpy().reshape(B * T, 24, 3, 3), R.detach().numpy()) < 1e-5
    assert rel_err(Po.cpu().numpy().reshape(B * T, 24, 3), Pp.detach().numpy()) < 1e-5
    # without the 6D copy (what test()/gen_seq launch): identical R / positions
    Ro2, Po2 = torch.zeros_like(Ro), torch.zeros_like(Po)
    ops.decode_tail_fwd(decs, offd, T, B, bp, None, Ro2, Po2)
    assert torch.equal(Ro2, Ro) and torch.equal(Po2, Po)


def test_adam_matches_torch(dev):
    import ctypes as C
    from hm_vae_amd import _lib
    torch.manual_seed(0)
    shapes = [(1000,), (37, 5), (3,)]
    ps = [torch.randn(s) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-4)
    dp = [p.clone().to(dev) for p in ps]
    dm = [torch.zeros_like(p) for p in dp]
    dv = [torch.zeros_like(p) for p in dp]
    dg = [torch.zeros_like(p) for p in dp]
    tab = (_lib.AdamTensor * len(dp))()
    for i in range(len(dp)):
        tab[i] = _lib.AdamTensor(dp[i].data_ptr(), dg[i].data_ptr(), dm[i].data_ptr(), dv[i].data_ptr(), dp[i].numel())
    tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
    lr_dev = torch.tensor([1e-3], device=dev)
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(5):
        gs = [torch.randn(s) for s in shapes]
        for r, g, d in zip(ref, gs, dg):
            r.grad = g.clone()
            d.copy_(g)
        opt.step()
        step_dev += 1
        _lib.check(_lib.load().hmvae_adam_step(tab_dev.data_ptr(), len(dp), max(p.numel() for p in dp), lr_dev.data_ptr(),
                                               step_dev.data_ptr(), 0.9, 0.999, 1e-8, 1e-4, _lib.stream()), "adam")
    for r, d in zip(ref, dp):
        assert rel_err(d.cpu().numpy(), r.detach().numpy()) < 1e-5


def test_wgrad_bf16x3_planes_match_fp32_path(dev, monkeypatch):
    """hmvae_split_bf16x3 is an exact split (p0 + p1 + p2 == x bit for bit) and the weight-gradient GEMM fed with the
    planes (six bf16 MFMA cross terms per product, fp32 accumulation) agrees with the fp32-MFMA path to fp32 rounding."""
    import ctypes as C
    from hm_vae_amd import ops, _lib
    x = torch.randn(4096 * 8, device=dev) * torch.logspace(-6, 6, 4096 * 8, device=dev)
    x[:16] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-25, 1e-30, 0.1, 1 / 3, 65504.0, 2 ** -20, 7.0, -7.5, 1e10, -1e-10])
    planes = torch.empty(3 * x.numel(), dtype=torch.int16, device=dev)
    _lib.check(_lib.load().hmvae_split_bf16x3(_lib.ptr(x), planes.data_ptr(), x.numel(), x.numel(), _lib.stream()), "split")
    p = planes.view(3, -1).to(torch.int32)
    parts = ((p & 0xffff) << 16).view(torch.float32)          # bf16 bits -> fp32
    back = (parts[0].double() + parts[1].double() + parts[2].double()).float()
    bad = (back != x).nonzero().reshape(-1)
    assert bad.numel() == 0, (x[bad[:8]], back[bad[:8]])        # (remainders below 2^-126 would flush: not in this range)
    # a masked conv's weight gradient both ways
    neigh = [[0, 1], [0, 1, 2], [1, 2, 3], [2, 3]]
    op = ops.conv_gemm(dev, neigh, 4 * 6, 4 * 12, 15, 1, 7, "reflect", 32, need_dx=True)
    B = 70
    bp = ops.pad_batch(B)
    xa = torch.zeros(24 * 32, bp, device=dev)
    dy = torch.zeros(48 * 32, bp, device=dev)
    xa[:, :B] = torch.randn(24 * 32, B, device=dev)
    dy[:, :B] = torch.randn(48 * 32, B, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setattr(ops, "WGRAD_BF16X3", mode)
        dw = torch.zeros(48, 24, 15, device=dev)
        op.backward_weight(dy, xa, dw, bp, slab)
        outs.append(dw)
    assert float(outs[0].abs().max()) > 0
    assert rel_err(outs[1].cpu().numpy(), outs[0].cpu().numpy()) < 2e-6


@pytest.mark.parametrize("cfg", [
    dict(kt=15, stride=1, pad=7, mode="reflect", t_in=32, B=70),
    dict(kt=15, stride=2, pad=7, mode="reflect", t_in=64, B=64),
    dict(kt=15, stride=2, pad=7, mode="zeros", t_in=22, B=130),          # t_out = 11: a partial last stage, zero rows
    dict(kt=3, stride=1, pad=1, mode="zeros", t_in=6, B=9),
    dict(kt=31, stride=1, pad=15, mode="reflect", t_in=40, B=8),         # the trajectory model's taps
])
def test_wgrad_lds_staged_kernel_matches_direct(dev, monkeypatch, cfg):
    """hmvae_wgrad_desc.mode = 1 (operands staged through LDS, activation window per channel) and mode = 2 (the same
    staging with every element split into three bf16 planes, six cross terms on the bf16 matrix pipe) against the direct
    fp32 kernel for every split the autotuner may pick, incl. the persistent schedule: same sums, different summation
    order -- and all three against a float64 evaluation of the convolution's weight gradient (skeleton.py:95-105)."""
    from hm_vae_amd import ops
    neigh = [[0, 1], [0, 1, 2], [1, 2, 3], [2, 3, 4], [3, 4]]
    cin, cout = 5 * 6, 5 * 12
    op = ops.conv_gemm(dev, neigh, cin, cout, cfg["kt"], cfg["stride"], cfg["pad"], cfg["mode"], cfg["t_in"], need_dx=True)
    assert op.lds_eligible() or op.wg_transposed
    if op.wg_transposed:
        pytest.skip("transposed orientation chosen: the LDS kernel does not apply")
    B = cfg["B"]
    bp = ops.pad_batch(B)
    t_out = op.t_out
    x = torch.zeros(cin * cfg["t_in"], bp, device=dev)
    dy = torch.zeros(cout * t_out, bp, device=dev)
    x[:, :B] = torch.randn(cin * cfg["t_in"], B, device=dev)
    dy[:, :B] = torch.randn(cout * t_out, B, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    monkeypatch.setattr(ops, "AUTOTUNE", False)
    # float64 yardstick: dW = d/dW sum(dy * conv1d(pad(x), W * mask))
    xb = x[:, :B].reshape(cin, cfg["t_in"], B).permute(2, 0, 1).double().cpu()
    gb = dy[:, :B].reshape(cout, t_out, B).permute(2, 0, 1).double().cpu()
    wt = torch.zeros(cout, cin, cfg["kt"], dtype=torch.float64, requires_grad=True)
    mask = O.conv_mask(neigh, cin, cout, cfg["kt"]).double()
    (O.skel_conv(xb, wt, mask, None, cfg["stride"], cfg["pad"], O._pad_mode(cfg["mode"])) * gb).sum().backward()
    exact = wt.grad.numpy()
    assert op.bf3_eligible()
    ref = None
    worst = {}
    assert op.pair_eligible() or cfg["kt"] != 15
    modes = ("0", "1", "2") + (("3",) if op.pair_eligible() else ())     # 3 = mode 2 on pairs of column tiles (8-wave workgroups)
    for mode in modes:
        monkeypatch.setattr(ops, "WGRAD_LDS", mode)
        for split in op.split_candidates(bp) + (["p"] if (op._k_chunks(bp) >= 8 and mode != "3") else []):
            op._split_tuned[bp] = split
            dw = torch.zeros(cout, cin, cfg["kt"], device=dev)
            op.backward_weight(dy, x, dw, bp, slab)
            assert op.wg_mode(bp) == int(mode)
            if ref is None:
                ref = dw
                assert float(ref.abs().max()) > 0
            else:
                assert rel_err(dw.cpu().numpy(), ref.cpu().numpy()) < 2e-6, (mode, split)
            worst[mode] = max(worst.get(mode, 0.0), rel_err(dw.cpu().numpy(), exact))
    # fp32-grade: the three-plane kernel is as close to float64 as the fp32 kernels (it rounds once per 32-term dot
    # product instead of once per term)
    assert max(worst.values()) < 2e-6, worst
    assert max(worst[m] for m in modes if m in ("2", "3")) <= 1.5 * worst["0"] + 1e-7, worst


@pytest.mark.parametrize("case", [
    dict(kind="dec", level=3, ch=96, co=48, Tc=4, B=70),        # first decoder layer: 8 x 4 position pairs, dense mask
    dict(kind="dec", level=2, ch=48, co=24, Tc=8, B=33),        # second: 16 x 8, unpooled edges (children), masked tiles
    dict(kind="dec", level=1, ch=5, co=3, Tc=4, B=256),         # ragged tiles: rows / channels not multiples of the tile
    dict(kind="enc", level=2, ci=24, co=48, T=16, B=70),        # stride 2 + folded SkeletonPool: children are rows (x 1/2)
    dict(kind="plain", T=16, B=5),                              # no composite at all: reflection only, 16 x 16 pairs
])
def test_wgrad_position_pairs_match_float64(dev, monkeypatch, case):
    """hmvae_wgrad_pq (csrc/wgrad_pq.hip): the weight gradient as ONE dense GEMM over (row, position) x (channel, source
    position) pairs on the bf16 pipe (three planes, six cross terms) + the contraction with the layer's position / tap
    relation in the epilogue, read from the tensor IN FRONT of Upsample / Unpool (decoder) or the layer's own input --
    against a float64 evaluation of the reference's arithmetic (skeleton.py:95-105 behind seq_two_hier_sa_vae.py:233-242),
    masked entries untouched, accumulate."""
    from hm_vae_amd import ops, topology
    torch.manual_seed(5)
    monkeypatch.setattr(ops, "WGRAD_PQ", "1")
    K, pad, B = 15, 7, case["B"]
    if case["kind"] == "plain":
        neigh = [[0, 1], [0, 1, 2], [1, 2]]
        cin, cout, T = 3 * 7, 3 * 9, case["T"]
        op = ops.conv_gemm(dev, neigh, cin, cout, K, 1, pad, "reflect", T, need_dx=True)
        mask = O.conv_mask(neigh, cin, cout, K).double()
        xs_b = torch.randn(B, cin, T, dtype=torch.float64)
        wt = torch.zeros(cout, cin, K, dtype=torch.float64, requires_grad=True)
        y = O.skel_conv(xs_b, wt, mask, None, 1, pad, "reflect")
        src_b, n_src, t_src, t_out = xs_b, cin, T, T
    else:
        edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
        for _ in range(case["level"]):
            edges = topology.derive_pooling(edges)[2]
        pooling = topology.derive_pooling(edges)[1] if case["level"] < 3 or case["kind"] == "enc" else [[i] for i in range(len(edges))]
        neigh = topology.find_neighbor(edges, 2)
        n_f, n_c = len(neigh), len(pooling)
        if case["kind"] == "dec":
            usrc = np.zeros(n_f, dtype=np.int64)
            for i, grp in enumerate(pooling):
                for j in grp:
                    usrc[j] = i
            ch, Tc = case["ch"], case["Tc"]
            cin, cout, T = n_f * ch, n_f * case["co"], 2 * Tc
            op = ops.conv_gemm(dev, neigh, cin, cout, K, 1, pad, "reflect", T, need_dx=True, upsample=True, unpool_src=usrc)
            mask = O.conv_mask(neigh, cin, cout, K).double()
            src_b = torch.randn(B, n_c * ch, Tc, dtype=torch.float64)
            idx = torch.as_tensor((usrc[:, None] * ch + np.arange(ch)[None, :]).reshape(-1))
            U = F.interpolate(src_b[:, idx, :], scale_factor=2, mode="linear", align_corners=False)
            wt = torch.zeros(cout, cin, K, dtype=torch.float64, requires_grad=True)
            y = O.skel_conv(U, wt, mask, None, 1, pad, "reflect")
            n_src, t_src, t_out = n_c * ch, Tc, T
        else:
            ci, co, T = case["ci"], case["co"], case["T"]
            cin, cout = n_f * ci, n_f * co
            op = ops.conv_gemm(dev, neigh, cin, cout, K, 2, pad, "reflect", T, need_dx=True, pool_groups=pooling)
            mask = O.conv_mask(neigh, cin, cout, K).double()
            src_b = torch.randn(B, cin, T, dtype=torch.float64)
            wt = torch.zeros(cout, cin, K, dtype=torch.float64, requires_grad=True)
            yc = O.skel_conv(src_b, wt, mask, None, 2, pad, "reflect")
            Pm = torch.zeros(n_c * co, cout, dtype=torch.float64)
            for i, grp in enumerate(pooling):
                for j in grp:
                    for c in range(co):
                        Pm[i * co + c, j * co + c] = 1.0 / len(grp)
            y = torch.einsum("pr,brt->bpt", Pm, yc)              # the pooled pre-activation: what the launch is handed
            n_src, t_src, t_out = cin, T, yc.shape[2]
    t = op.pq()
    assert t is not None and (t.T, t.S) == (t_out, t_src)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    exact = wt.grad.numpy()
    bp = ops.pad_batch(B)
    dys = torch.zeros(y.shape[1] * t_out, bp, device=dev)
    dys[:, :B] = gy.permute(1, 2, 0).reshape(-1, B).float().to(dev)
    srcs = torch.zeros(n_src * t_src, bp, device=dev)
    srcs[:, :B] = src_b.permute(1, 2, 0).reshape(-1, B).float().to(dev)
    if case["kind"] == "dec" and case["level"] == 2:
        assert t.n_tiles < -(-t.n_rows // (128 // t.T)) * -(-t.n_chan // (128 // t.S))      # fully masked tiles are not launched
    mk = torch.from_numpy(mask.numpy()).float().to(dev)
    dw = torch.full((cout, cin, K), 7.0, device=dev) * mk
    assert op.use_pq(bp)
    op.backward_weight(dys, None, dw, bp, None, src=srcs)
    got = dw.cpu().numpy()
    assert float(np.abs(exact).max()) > 0
    assert rel_err(got, exact) < 2e-6
    assert float(np.abs(got * (1 - mask.numpy())).max()) == 0.0
    op.backward_weight(dys, None, dw, bp, None, accumulate=True, src=srcs)
    assert rel_err(dw.cpu().numpy(), 2.0 * exact) < 2e-6
    # the direct launch on the layer's own input agrees
    if case["kind"] == "dec":
        monkeypatch.setattr(ops, "WGRAD_PQ", "0")
        us = torch.zeros(cin * T, bp, device=dev)
        us[:, :B] = U.permute(1, 2, 0).reshape(-1, B).float().to(dev)
        dw2 = torch.zeros(cout, cin, K, device=dev)
        op.backward_weight(dys, us, dw2, bp, torch.empty(op.slab_floats(bp), device=dev), src=srcs)
        assert rel_err(dw2.cpu().numpy(), got) < 4e-6


@pytest.mark.parametrize("cfg", [
    dict(rows=28, ch=12, neigh=[[0, 1, 2, 3]] * 4, kt=15, stride=1, pad=7, mode="reflect", t_in=8, B=70),      # 112 rows: mt 7
    dict(rows=24, ch=12, neigh=[[0, 1, 2, 3]] * 4, kt=15, stride=2, pad=7, mode="reflect", t_in=16, B=64),     # 96 rows: mt 6
    dict(rows=20, ch=6, neigh=[[0, 1, 2, 3]] * 4, kt=15, stride=1, pad=7, mode="zeros", t_in=9, B=33),         # 80 rows: mt 5
    dict(rows=28, ch=12, neigh=[[0, 1, 2, 3], [0, 1, 2, 3], [0, 1, 2, 3], [2, 3]], kt=15, stride=2, pad=7,
         mode="reflect", t_in=12, B=130),                                               # blocks read by 84 / 112 rows: dead row tiles
])
def test_wgrad_wide_tiles_match_float64(dev, monkeypatch, cfg):
    """wg_mode 4 (wgrad_bf3w_kernel: up to 112 rows x 128 columns per 8-wave workgroup, three bf16 planes, six cross
    terms): every row-tile count 5..7, groups with dead row tiles, a group's unpaired last column tile, ragged batches,
    zero padding, the fused epilogue (split 1) and the slab + reduce route (split 2), accumulate -- against a float64
    evaluation of the convolution's weight gradient (skeleton.py:95-105) and against the 48 x 64 kernel (mode 2)."""
    from hm_vae_amd import ops
    nj = len(cfg["neigh"])
    cin, cout = nj * cfg["ch"], nj * cfg["rows"]
    op = ops.conv_gemm(dev, cfg["neigh"], cin, cout, cfg["kt"], cfg["stride"], cfg["pad"], cfg["mode"], cfg["t_in"], need_dx=True)
    w = op.wide()
    assert w is not None and w.g.mt in (5, 6, 7), None if w is None else w.g.mt
    B = cfg["B"]
    bp = ops.pad_batch(B)
    t_out = op.t_out
    x = torch.zeros(cin * cfg["t_in"], bp, device=dev)
    dy = torch.zeros(cout * t_out, bp, device=dev)
    x[:, :B] = torch.randn(cin * cfg["t_in"], B, device=dev)
    dy[:, :B] = torch.randn(cout * t_out, B, device=dev)
    slab = torch.empty(op.slab_floats(bp), device=dev)
    monkeypatch.setattr(ops, "AUTOTUNE", False)
    xb = x[:, :B].reshape(cin, cfg["t_in"], B).permute(2, 0, 1).double().cpu()
    gb = dy[:, :B].reshape(cout, t_out, B).permute(2, 0, 1).double().cpu()
    wt = torch.zeros(cout, cin, cfg["kt"], dtype=torch.float64, requires_grad=True)
    mask = O.conv_mask(cfg["neigh"], cin, cout, cfg["kt"]).double()
    (O.skel_conv(xb, wt, mask, None, cfg["stride"], cfg["pad"], O._pad_mode(cfg["mode"])) * gb).sum().backward()
    exact = wt.grad.numpy()
    res = {}
    for mode, splits in (("2", (1,)), ("4", (1, 2))):
        monkeypatch.setattr(ops, "WGRAD_LDS", mode)
        for split in splits:
            if split > 1 and split not in op.split_candidates(bp):
                continue
            op._split_tuned[bp] = split
            dw = torch.full((cout, cin, cfg["kt"]), 7.0, device=dev) * torch.from_numpy(mask.numpy()).float().to(dev)
            op.backward_weight(dy, x, dw, bp, slab)
            assert op.wg_mode(bp) == int(mode) and op.fusable(bp) == (split == 1)
            res[(mode, split)] = dw.cpu().numpy()
            assert rel_err(res[(mode, split)], exact) < 2e-6, (mode, split)
            assert float(np.abs(res[(mode, split)] * (1 - mask.numpy())).max()) == 0.0
    assert ("4", 1) in res
    for k, v in res.items():
        assert rel_err(v, res[("2", 1)]) < 2e-6, k
    # accumulate: dW += (fused epilogue)
    monkeypatch.setattr(ops, "WGRAD_LDS", "4")
    op._split_tuned[bp] = 1
    dw = torch.from_numpy(res[("4", 1)]).to(dev).clone()
    op.backward_weight(dy, x, dw, bp, slab, accumulate=True)
    assert rel_err(dw.cpu().numpy(), 2.0 * exact) < 2e-6
