"""CPU checks of the host-side planning (hm_vae_amd/planner.py, ops.py tables): the numpy emulation of the
kernel semantics, fed with the device tables, must reproduce the oracle's SkeletonConv / Linear fwd + bwd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hmvae_oracle as O
from hm_vae_amd import ops, planner, topology
from hm_vae_amd import planner as P
from emu import emu_pack, emu_gather, emu_wgrad, emu_wgrad_pq, emu_pack4, emu_gather4
from helpers import rel_err


def check_v4_tables(side, dense, src, ref_soa_fn, ref, n_rows, bias=None, leaky=False):
    """The v4 kernel's tables (planner.make_groups_v4, 32-channel K steps, bf16 planes in A-operand order, split K) driven
    through their numpy emulation must give the oracle's result: both group sizes, unsplit and split three ways."""
    n = 0
    for mtw in (1, 2):
        sd = side.v4_side(mtw)
        if sd is None:
            continue
        g = sd.g
        assert g.mt == 4 * mtw and (g.g_nc4 % 8 == 0).all()
        rows = g.row_id[g.row_id >= 0]
        assert len(set(rows.tolist())) == len(rows) == len(set(side.g.row_id[side.g.row_id >= 0].tolist()))   # every row once
        wp4 = emu_pack4(sd, dense)
        for split in (1, 3):
            out = np.zeros((n_rows, src.shape[1]), dtype=np.float32)
            emu_gather4(sd, wp4, src, out, nc=64, split=split, bias=bias, leaky=leaky)
            assert rel_err(ref_soa_fn(out), ref) < 1e-5, (mtw, split)
            n += 1
    assert n > 0


def to_soa(x):          # (B, C, T) -> [C*T, bp]
    B, Cn, T = x.shape
    bp = ops.pad_batch(B)
    out = np.zeros((Cn * T, bp), dtype=np.float32)
    out[:, :B] = x.permute(1, 2, 0).reshape(Cn * T, B).numpy()
    return out


def from_soa(a, B, Cn, T):
    return torch.from_numpy(a[:, :B].reshape(Cn, T, B)).permute(2, 0, 1).contiguous()


def test_topology_matches_oracle():
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    assert edges == O.skeleton_edges(O.SMPL24_PARENTS)
    cur = edges
    for lvl in range(4):
        assert topology.find_neighbor(cur, 2) == O.edge_neighbours(cur, 2)
        a = topology.derive_pooling(cur, last_pool=(lvl == 3))
        b = O.pool_plan(cur, last_pool=(lvl == 3))
        assert a[0] == b[0] and a[1] == b[1] and [list(e) for e in a[2]] == [list(e) for e in b[2]]
        if lvl < 3:
            cur = a[2]
    for T in (8, 16, 64):
        assert topology.stride_schedule(T, 4) == O.level_strides(T, 4)


@pytest.mark.parametrize("case", [
    dict(level=0, ci=2, co=4, K=5, stride=2, T=8, mode="reflect"),      # tiny, strided, reflect
    dict(level=1, ci=3, co=2, K=3, stride=1, T=4, mode="reflect"),
    dict(level=2, ci=4, co=8, K=5, stride=1, T=6, mode="zeros"),
    dict(level=3, ci=4, co=20, K=3, stride=2, T=4, mode="reflect"),     # all-neighbour level, rows > 16
    dict(level=3, ci=3, co=5, K=15, stride=1, T=8, mode="reflect"),     # fewer positions than taps: tap folding (dec0)
    dict(level=3, ci=2, co=6, K=15, stride=2, T=8, mode="reflect"),     # ... strided (enc3)
    dict(level=2, ci=2, co=3, K=7, stride=1, T=4, mode="zeros"),        # ... zero padding
])
def test_conv_tables_reproduce_oracle(case):
    torch.manual_seed(case["level"])
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(case["level"]):
        edges = topology.derive_pooling(edges)[2]
    neigh = topology.find_neighbor(edges, 2)
    n = len(neigh)
    cin, cout, K, s, T = case["ci"] * n, case["co"] * n, case["K"], case["stride"], case["T"]
    pad = (K - 1) // 2
    B = 3
    op = ops.conv_gemm("cpu", neigh, cin, cout, K, s, pad, case["mode"], T, need_dx=True)
    mask = O.conv_mask(neigh, cin, cout, K)
    w = (torch.randn(cout, cin, K) * mask).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    x = torch.randn(B, cin, T, requires_grad=True)
    y = O.skel_conv(x, w, mask, b, s, pad, "reflect" if case["mode"] == "reflect" else "constant")
    gy = torch.randn_like(y)
    y.backward(gy)
    T_out = y.shape[2]
    # forward
    wp = emu_pack(op.fwd, w.detach().numpy())
    ys = np.zeros((cout * T_out, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.fwd, wp, to_soa(x.detach()), ys, bias=b.detach().numpy())
    assert rel_err(from_soa(ys, B, cout, T_out).numpy(), y.detach().numpy()) < 1e-5
    # backward data
    wpb = emu_pack(op.bwd, w.detach().numpy())
    dxs = np.zeros((cin * T, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.bwd, wpb, to_soa(gy), dxs)
    assert rel_err(from_soa(dxs, B, cin, T).numpy(), x.grad.numpy()) < 1e-5
    check_v4_tables(op.fwd, w.detach().numpy(), to_soa(x.detach()), lambda o: from_soa(o, B, cout, T_out).numpy(),
                    y.detach().numpy(), cout * T_out, bias=b.detach().numpy())
    check_v4_tables(op.bwd, w.detach().numpy(), to_soa(gy), lambda o: from_soa(o, B, cin, T).numpy(), x.grad.numpy(), cin * T)
    # backward weight
    dw = np.zeros((cout, cin, K), dtype=np.float32)
    emu_wgrad(op, to_soa(gy), to_soa(x.detach()), dw)
    assert rel_err(dw, (w.grad * mask).numpy()) < 1e-5
    assert op.folded or not (T < K and case["mode"] == "reflect")     # zero padding has nothing to fold
    assert not (op.folded and T >= K)
    if op.folded:               # fewer (position, virtual tap) items than (position, tap) ones
        assert sum(len(l) for l in op.fwd.pairs_host) < T_out * K
    # grouping sanity: every masked (row, chan) pair is covered exactly once
    g = op.fwd.g
    cover = np.zeros((cout, cin), dtype=np.int32)
    mask_rows = mask[:, :, 0].numpy()
    rid = g.row_id.reshape(g.n_groups, -1)
    for gi in range(g.n_groups):
        rows = rid[gi][rid[gi] >= 0]
        ch = g.chan_id[g.g_choff[gi]:g.g_choff[gi] + g.g_nc4[gi] * 4]
        ch = ch[ch >= 0]
        cover[np.ix_(rows, ch)] += 1
    assert (cover[mask_rows > 0] == 1).all()
    assert cover.max() == 1


@pytest.mark.parametrize("case", [
    dict(level=2, ch=2, co=3, K=15, Tc=4),       # dec0-like: 4 coarse positions, no edge is duplicated
    dict(level=1, ch=2, co=2, K=15, Tc=8),       # dec1-like
    dict(level=0, ch=1, co=1, K=5, Tc=6),        # finest level: 24 edges from 14
])
def test_conv_behind_upsample_unpool_reproduces_oracle(case):
    """The composite gather-GEMM (Upsample x2 linear + SkeletonUnpool folded into the conv, planner.compose_conv /
    merge_channels) on the COARSE tensor == conv(upsample(unpool(coarse))) of the oracle, forward and both gradients."""
    torch.manual_seed(3 + case["level"])
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(case["level"]):
        edges = topology.derive_pooling(edges)[2]
    pooling = topology.derive_pooling(edges)[1]                  # coarse edge -> its fine edges
    neigh = topology.find_neighbor(edges, 2)
    n_f, n_c = len(neigh), len(pooling)
    src = np.zeros(n_f, dtype=np.int64)
    for i, grp in enumerate(pooling):
        for j in grp:
            src[j] = i
    ch, K, Tc = case["ch"], case["K"], case["Tc"]
    cin, cout, T = n_f * ch, n_f * case["co"], 2 * Tc
    pad, B = (K - 1) // 2, 3
    mask = O.conv_mask(neigh, cin, cout, K)
    w = (torch.randn(cout, cin, K) * mask).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    xc = torch.randn(B, n_c * ch, Tc, requires_grad=True)
    idx = torch.as_tensor((src[:, None] * ch + np.arange(ch)[None, :]).reshape(-1))
    U = F.interpolate(xc[:, idx, :], scale_factor=2, mode="linear", align_corners=False)
    U.retain_grad()
    y = O.skel_conv(U, w, mask, b, 1, pad, "reflect")
    gy = torch.randn_like(y)
    y.backward(gy)
    op = ops.conv_gemm("cpu", neigh, cin, cout, K, 1, pad, "reflect", T, need_dx=True, upsample=True, unpool_src=src)
    assert op.folded and op.fused_front
    assert op.fwd.g.mfma_cost() * op.fwd.g.kt * 0 + sum(len(l) for l in op.fwd.pairs_host) < T * K
    ys = np.zeros((cout * T, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.fwd, emu_pack(op.fwd, w.detach().numpy()), to_soa(xc.detach()), ys, bias=b.detach().numpy())
    assert rel_err(from_soa(ys, B, cout, T).numpy(), y.detach().numpy()) < 1e-5
    dxs = np.zeros((n_c * ch * Tc, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.bwd, emu_pack(op.bwd, w.detach().numpy()), to_soa(gy), dxs)
    assert rel_err(from_soa(dxs, B, n_c * ch, Tc).numpy(), xc.grad.numpy()) < 1e-5
    check_v4_tables(op.fwd, w.detach().numpy(), to_soa(xc.detach()), lambda o: from_soa(o, B, cout, T).numpy(),
                    y.detach().numpy(), cout * T, bias=b.detach().numpy())
    check_v4_tables(op.bwd, w.detach().numpy(), to_soa(gy), lambda o: from_soa(o, B, n_c * ch, Tc).numpy(),
                    xc.grad.numpy(), n_c * ch * Tc)
    dw = np.zeros((cout, cin, K), dtype=np.float32)
    emu_wgrad(op, to_soa(gy), to_soa(U.detach()), dw)              # the weight gradient keeps the conv's own input
    assert rel_err(dw, (w.grad * mask).numpy()) < 1e-5
    # ... or comes from the COARSE tensor through the position-pair products (csrc/wgrad_pq.hip; powers of two, few positions)
    assert (op.pq() is not None) == (case["Tc"] in (4, 8))
    if op.pq() is not None:
        dw2 = np.full((cout, cin, K), 5.0, dtype=np.float32) * mask.numpy()
        emu_wgrad_pq(op, to_soa(gy), to_soa(xc.detach()), dw2)
        assert rel_err(dw2, (w.grad * mask).numpy()) < 1e-5
        assert float(np.abs(dw2 * (1 - mask.numpy())).max()) == 0.0
        emu_wgrad_pq(op, to_soa(gy), to_soa(xc.detach()), dw2, accumulate=True)
        assert rel_err(dw2, 2.0 * (w.grad * mask).numpy()) < 1e-5


@pytest.mark.parametrize("case", [
    dict(level=0, ci=2, co=3, K=15, s=2, T=16),      # enc0-like: 24 edges pooled to 14
    dict(level=2, ci=2, co=2, K=15, s=2, T=8),       # short sequence: taps folded as well
    dict(level=1, ci=1, co=2, K=5, s=1, T=6),
])
def test_conv_followed_by_pool_reproduces_oracle(case):
    """SkeletonPool (mean of <= 2 edges) + LeakyReLU folded into the convolution (planner.merge_rows): forward gives
    the pooled activated tensor, backward-data takes the gradient w.r.t. the pooled pre-activation."""
    torch.manual_seed(7 + case["level"])
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    for _ in range(case["level"]):
        edges = topology.derive_pooling(edges)[2]
    pooling = topology.derive_pooling(edges)[1]
    neigh = topology.find_neighbor(edges, 2)
    n_f, n_p = len(neigh), len(pooling)
    K, st, T = case["K"], case["s"], case["T"]
    cin, cout, co = n_f * case["ci"], n_f * case["co"], case["co"]
    pad, B = (K - 1) // 2, 3
    mask = O.conv_mask(neigh, cin, cout, K)
    w = (torch.randn(cout, cin, K) * mask).requires_grad_(True)
    b = torch.randn(cout, requires_grad=True)
    x = torch.randn(B, cin, T, requires_grad=True)
    y = O.skel_conv(x, w, mask, b, st, pad, "reflect")
    y.retain_grad()
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
    op = ops.conv_gemm("cpu", neigh, cin, cout, K, st, pad, "reflect", T, need_dx=True, pool_groups=pooling)
    assert op.folded and op.fused_pool
    As = np.zeros((n_p * co * T_out, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.fwd, emu_pack(op.fwd, w.detach().numpy()), to_soa(x.detach()), As, bias=b.detach().numpy(), leaky=True)
    assert rel_err(from_soa(As, B, n_p * co, T_out).numpy(), a.detach().numpy()) < 1e-5
    dxs = np.zeros((cin * T, ops.pad_batch(B)), dtype=np.float32)
    emu_gather(op.bwd, emu_pack(op.bwd, w.detach().numpy()), to_soa(pre.grad), dxs)
    assert rel_err(from_soa(dxs, B, cin, T).numpy(), x.grad.numpy()) < 1e-5
    check_v4_tables(op.fwd, w.detach().numpy(), to_soa(x.detach()), lambda o: from_soa(o, B, n_p * co, T_out).numpy(),
                    a.detach().numpy(), n_p * co * T_out, bias=b.detach().numpy(), leaky=True)
    check_v4_tables(op.bwd, w.detach().numpy(), to_soa(pre.grad), lambda o: from_soa(o, B, cin, T).numpy(), x.grad.numpy(),
                    cin * T)
    dw = np.zeros((cout, cin, K), dtype=np.float32)
    # the rows a pool group averages share their gradient: the deduplicated weight gradient takes the pooled one
    emu_wgrad(op, to_soa(pre.grad if op.wg_pooled_dz else y.grad), to_soa(x.detach()), dw)
    assert op.wg_pooled_dz
    assert rel_err(dw, (w.grad * mask).numpy()) < 1e-5
    # the first encoder layer wants no data gradient: its weight gradient is still tiled by input block over the pooled rows
    op0 = ops.conv_gemm("cpu", neigh, cin, cout, K, st, pad, "reflect", T, need_dx=False, pool_groups=pooling)
    assert op0.bwd is None and op0.wg_pooled_dz
    dw0 = np.zeros((cout, cin, K), dtype=np.float32)
    emu_wgrad(op0, to_soa(pre.grad), to_soa(x.detach()), dw0)
    assert rel_err(dw0, (w.grad * mask).numpy()) < 1e-5
    if op.pq() is not None:          # pooled rows: one product, every child row written with the mean's factor
        dw2 = np.zeros((cout, cin, K), dtype=np.float32)
        emu_wgrad_pq(op, to_soa(pre.grad), to_soa(x.detach()), dw2)
        assert rel_err(dw2, (w.grad * mask).numpy()) < 1e-5
    assert (op.pq() is not None) == (case["T"] in (8, 16) and K == 15), (case, op.pq() is not None)


def test_linear_tables_reproduce_oracle():
    torch.manual_seed(5)
    B, k_edges, n_in, n_out = 5, 3, 12, 20
    op = ops.linear_gemm("cpu", n_out, n_in, k_edges, n_in, n_out, need_dx=True)
    w = torch.randn(n_out, n_in, requires_grad=True)
    b = torch.randn(n_out, requires_grad=True)
    x = torch.randn(B, k_edges, n_in, requires_grad=True)
    y = F.linear(x, w, b)
    gy = torch.randn_like(y)
    y.backward(gy)
    bp = ops.pad_batch(B)
    xs = np.zeros((k_edges * n_in, bp), np.float32)
    xs[:, :B] = x.detach().reshape(B, -1).t().numpy()
    ys = np.zeros((k_edges * n_out, bp), np.float32)
    emu_gather(op.fwd, emu_pack(op.fwd, w.detach().numpy()), xs, ys, bias=b.detach().numpy())
    assert rel_err(ys[:, :B].T.reshape(B, k_edges, n_out), y.detach().numpy()) < 1e-5
    gys = np.zeros_like(ys)
    gys[:, :B] = gy.reshape(B, -1).t().numpy()
    dxs = np.zeros_like(xs)
    emu_gather(op.bwd, emu_pack(op.bwd, w.detach().numpy()), gys, dxs)
    assert rel_err(dxs[:, :B].T.reshape(B, k_edges, n_in), x.grad.numpy()) < 1e-5
    dw = np.zeros((n_out, n_in, 1), np.float32)
    emu_wgrad(op, gys, xs, dw)
    assert rel_err(dw[:, :, 0], w.grad.numpy()) < 1e-5


def test_grouping_efficiency_len64():
    """Record (and bound) how much of the MFMA work is useful for the real len64 layers."""
    edges = topology.edges_from_parents(topology.SMPL24_PARENTS)
    topo, neighs = [edges], []
    for lvl in range(4):
        neighs.append(topology.find_neighbor(topo[-1], 2))
        topo.append(topology.derive_pooling(topo[-1], last_pool=(lvl == 3))[2])
    layers = [(0, 144, 288), (1, 168, 336), (2, 216, 432), (3, 336, 672), (3, 672, 336), (2, 432, 216), (1, 336, 168),
              (0, 576, 144)]
    for lvl, cin, cout in layers:
        rows, chans = planner.conv_joint_lists(neighs[lvl], cin, cout)
        gs = planner.make_groups(rows, chans, 15)
        eff = gs.useful() / (gs.mfma_cost())
        assert eff > 0.55, (lvl, cin, cout, gs.mt, eff)


@pytest.mark.parametrize("np_", [1, 2, 4])
@pytest.mark.parametrize("geom", [(64, 15, 2, "reflect"), (8, 15, 1, "reflect"), (8, 3, 2, "reflect"), (6, 5, 1, "zeros"),
                                  (2, 3, 1, "reflect")])
def test_items_expand_to_pairs(np_, geom):
    """The v2 kernel's item lists (source position + one tap per owned output position) must enumerate exactly
    the (q, tap) pairs of every output position, forward and backward-data."""
    t_in, kt, stride, mode = geom
    pad = (kt - 1) // 2
    t_out = (t_in + 2 * pad - kt) // stride + 1
    for pairs in (planner.conv_pairs_fwd(t_in, t_out, kt, stride, pad, mode),
                  planner.conv_pairs_bwd(t_in, t_out, kt, stride, pad, mode)):
        pb_off, pb_len, items = planner.build_items(pairs, np_)
        items = items.reshape(-1, 2)
        got = [[] for _ in pairs]
        for pb in range(len(pb_off)):
            for k in range(pb_off[pb], pb_off[pb] + pb_len[pb]):
                q, word = int(items[k, 0]), int(items[k, 1]) & 0xffffffff
                for pl in range(4):
                    tap = (word >> (8 * pl)) & 0xff
                    if tap != 0xff:
                        assert pl < np_ and pb * np_ + pl < len(pairs)
                        got[pb * np_ + pl].append((q, tap))
        for a, b in zip(got, pairs):
            assert sorted(a) == sorted(b)
    # heuristic returns a supported (np, ksplit)
    for mt in (1, 2, 3, 4):
        np2, ks = planner.pick_np_ksplit(mt, 7, t_out, 4, 20.0)
        assert np2 * mt <= 4 and ks in (1, 2, 4) and np2 <= max(1, t_out)


# ---------------------------------------------------------------------------------------------------------------------
# window plans + stream-K schedule of the v3 gather-GEMM
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t_in,t_out,stride", [(64, 64, 1), (8, 8, 1), (64, 32, 2), (8, 4, 2), (16, 16, 1)])
@pytest.mark.parametrize("chunk,ts", [(8, 1), (4, 2), (2, 4), (1, 8)])
def test_window_plan_expands_to_pairs(t_in, t_out, stride, chunk, ts):
    """Every wave's (window row, tap) items, mapped through win_rows, are exactly the (q, tap) pairs of its position."""
    kt = 15
    for pairs in (P.conv_pairs_fwd(t_in, t_out, kt, stride, 7, "reflect"), P.conv_pairs_bwd(t_in, t_out, kt, stride, 7, "reflect")):
        plan = P.pick_window_un(pairs, kt, chunk, ts)
        if plan is None:            # relation outside the kernel's limits (items per wave): v2 serves it
            continue
        win, un = plan["win"], plan["un"]
        wt = plan["wave_tab"].reshape(-1, 8, 4)
        items = plan["items"].reshape(-1, 8, 32)
        rows = plan["win_rows"].reshape(-1, win)
        assert 4 <= win <= 32 and rows.min() >= 0 and rows.max() < max(q for l in pairs for q, _ in l) + 1
        got = [[] for _ in pairs]
        for c in range(wt.shape[0]):
            for w in range(8):
                p, pat, n, _ = wt[c, w]
                assert n % un == 0 and n <= 20
                lst = [(int(v) & 0xffff, int(v) >> 16) for v in items[c, w, :n]]
                real = [(r, t) for r, t in lst if t != kt]
                assert all(t == kt for _, t in lst[len(real):])          # padding (zero fragment) only at the end
                if pat == 1:
                    assert real == [(real[0][0] + i, i) for i in range(kt)]
                if pat == 2:
                    assert real == [(real[0][0] + i, kt - 1 - i) for i in range(kt)]
                if p >= 0:
                    got[p].extend((int(rows[c, r]), t) for r, t in real)
                else:
                    assert not real
        for p, lst in enumerate(pairs):
            assert sorted(got[p]) == sorted(lst), p


def test_stream_schedule_covers_and_balances():
    rng = np.random.default_rng(0)
    for n_workers in (8, 256, 512):
        nc4 = rng.integers(5, 170, size=13)
        steps = [int(n) for n in nc4 for _ in range(32)]
        cost = [int(rng.choice([15, 15, 15, 18])) for _ in steps]
        segs, wseg, fix, n_parts = P.build_stream_schedule(steps, cost, n_workers)
        cov = np.zeros(len(steps), dtype=np.int64)
        prev = {}
        for u, k0, k1, part in segs:
            assert 0 <= k0 < k1 <= steps[u]
            assert prev.get(u, 0) == k0            # the parts of a unit are consecutive K ranges, in order
            prev[u] = k1
            cov[u] += k1 - k0
        assert (cov == np.asarray(steps)).all()
        assert wseg[:, 1].sum() == len(segs) and (wseg[1:, 0] == wseg[:-1, 0] + wseg[:-1, 1]).all()
        load = np.array([sum((segs[i][2] - segs[i][1]) * cost[segs[i][0]] for i in range(a, a + n)) for a, n in wseg])
        assert load.max() <= 1.15 * load.mean() + 3 * 18
        # cut units: every part slot used exactly once, in K order
        parts = sorted(int(p) for _, _, _, p in segs if p >= 0)
        assert parts == list(range(n_parts))
        for u, p0, n in fix:
            mine = [(k0, p) for uu, k0, _, p in segs if uu == u]
            assert [p for _, p in sorted(mine)] == list(range(p0, p0 + n)) and n >= 2
        whole = [u for u, _, _, p in segs if p < 0]
        assert len(set(whole)) == len(whole)
