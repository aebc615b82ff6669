"""Numpy emulation of the MFMA kernels' *semantics* (include/hmvae.h), driven by the very same tables the
device kernels read.  Lets the CPU suite validate planner.py / ops.py table construction against the oracle
without a GPU.  Test-only."""
import numpy as np


def _np(t):
    return t.detach().cpu().numpy()


def emu_pack(side, dense):
    g = side.g
    rs, cs, ts = side.strides
    dense = dense.reshape(-1)
    wp = np.zeros(max(g.packed_size, 1), dtype=np.float32)
    row_id = g.row_id.reshape(g.n_groups, g.mt * 16)
    fold = getattr(side, "fold", None)
    if fold is not None:        # composite op (hmvae_pack_op): virtual taps over merged dense pairs
        nf, kr = fold["nf"], fold["kt_real"]
        taps, coefs = _np(fold["taps"]).reshape(-1, nf), _np(fold["coefs"]).reshape(-1, nf)
        ctab = None if fold["child_tab"] is None else _np(fold["child_tab"]).reshape(-1, 4)
    for gi in range(g.n_groups):
        for c4 in range(int(g.g_nc4[gi])):
            for mt in range(g.mt):
                for kk in range(4):
                    cid = g.chan_id[g.g_choff[gi] + c4 * 4 + kk]
                    for m in range(16):
                        rid = row_id[gi, mt * 16 + m]
                        bits = int(g.pair_ok[g.g_okoff[gi] + (mt * 16 + m) * (int(g.g_nc4[gi]) * 4) + c4 * 4 + kk])
                        if rid < 0 or cid < 0 or not bits:
                            continue
                        if fold is None:
                            vals = [dense[rid * rs + cid * cs + tap * ts] for tap in range(g.kt)]
                        else:
                            real = np.zeros(kr, dtype=np.float32)
                            for ci in range(4):
                                if not (bits >> ci) & 1:
                                    continue
                                dr, dc = rid, cid
                                if ctab is not None:
                                    if fold["child_side"]:
                                        dr = ctab[rid, ci]
                                    else:
                                        dc = ctab[cid, ci]
                                real += np.array([dense[dr * rs + dc * cs + k * ts] for k in range(kr)], dtype=np.float32)
                            if fold.get("child_scale") is not None:
                                real *= _np(fold["child_scale"])[rid if fold["child_side"] else cid]
                            vals = [np.float32(sum(coefs[vt, j] * real[taps[vt, j]] for j in range(nf) if taps[vt, j] >= 0))
                                    for vt in range(g.kt)]
                        for tap in range(g.kt):
                            wp[g.g_woff[gi] + ((c4 * g.kt + tap) * g.mt + mt) * 64 + kk * 16 + m] = vals[tap]
    return wp


def emu_gather(side, wp, src, dst, bias=None, leaky=False, accum=False, slope=0.2):
    """src, dst: numpy [rows, bp]."""
    g = side.g
    chan_rows, dst_rows = _np(side.chan_rows), _np(side.dst_rows)
    bias_idx = _np(side.bias_idx)
    bias_idx2 = None if getattr(side, "bias_idx2", None) is None else _np(side.bias_idx2)
    p_off, p_len, pairs = _np(side.p_off), _np(side.p_len), _np(side.pairs)
    R = g.mt * 16
    for gi in range(g.n_groups):
        nc4 = int(g.g_nc4[gi])
        W = wp[g.g_woff[gi]:g.g_woff[gi] + nc4 * g.kt * g.mt * 64].reshape(nc4, g.kt, g.mt, 4, 16)
        rows_src = chan_rows[g.g_choff[gi]:g.g_choff[gi] + nc4 * 4].reshape(nc4, 4)
        for p in range(side.n_pos):
            acc = np.zeros((R, src.shape[1]), dtype=np.float64)
            for i in range(int(p_len[p])):
                pr = int(pairs[p_off[p] + i])
                q, tap = pr & 0xffff, pr >> 16
                for c4 in range(nc4):
                    for kk in range(4):
                        w = W[c4, tap, :, kk, :].reshape(R)           # [mt*16 + m]
                        acc += w[:, None].astype(np.float64) * src[rows_src[c4, kk] + q * side.src_ts][None, :]
            for r in range(R):
                dr = dst_rows[gi * R + r]
                if dr < 0:
                    continue
                v = acc[r]
                if bias is not None and bias_idx[gi * R + r] >= 0:
                    bv = bias[bias_idx[gi * R + r]]
                    if bias_idx2 is not None and bias_idx2[gi * R + r] >= 0:      # mean of two dense rows (pool folded in)
                        bv = 0.5 * (bv + bias[bias_idx2[gi * R + r]])
                    v = v + bv
                if leaky:
                    v = np.where(v > 0, v, v * slope)
                row = dr + p * side.dst_ts
                dst[row] = (dst[row] + v) if accum else v
    return dst


def emu_wgrad(op, dy, x, dw):
    """dy, x: numpy [rows, bp]; writes dense dw (n_rows, n_chan, kt) at masked positions."""
    g = op.wg
    ge = op.geom
    R = g.mt * 16
    chan_rows, dz_rows = _np(op.wg_chan_rows), _np(op.dz_rows)
    row_id = g.row_id.reshape(g.n_groups, R)
    n_rows, n_chan, kt = op.w_shape
    dwf = dw.reshape(n_rows, n_chan, kt)
    for gi in range(g.n_groups):
        for ci in range(int(g.g_nch[gi])):
            cid = g.chan_id[g.g_choff[gi] + ci]
            xr = chan_rows[g.g_choff[gi] + ci]
            for tap in range(kt):
                acc = np.zeros(R, dtype=np.float64)
                for t in range(ge["t_dst"]):
                    u = t * ge["stride"] + tap - ge["pad"]
                    if ge["pad_mode"] == 0:
                        if u < 0:
                            u = -u
                        if u >= ge["t_src"]:
                            u = 2 * (ge["t_src"] - 1) - u
                    elif u < 0 or u >= ge["t_src"]:
                        continue
                    xv = x[xr + u * op.x_ts].astype(np.float64)
                    for r in range(R):
                        if dz_rows[gi * R + r] >= 0:
                            acc[r] += float(np.dot(dy[dz_rows[gi * R + r] + t * op.y_ts].astype(np.float64), xv))
                for r in range(R):
                    bits = int(g.pair_ok[g.g_okoff[gi] + r * (int(g.g_nc4[gi]) * 4) + ci])
                    if row_id[gi, r] < 0 or not bits:
                        continue
                    child = getattr(op, "wg_child", None)
                    if child is None:
                        dwf[row_id[gi, r], cid, tap] = acc[r]
                        continue
                    ctab, side, scale = _np(child[0]).reshape(-1, 4), child[1], child[2]
                    rid = row_id[gi, r]
                    v = acc[r] * (float(_np(scale)[rid if side else cid]) if scale is not None else 1.0)
                    for k in range(4):          # one computed gradient, several dense owners (hmvae_wreduce_desc)
                        if (bits >> k) & 1:
                            dwf[ctab[rid, k] if side else rid, cid if side else ctab[cid, k], tap] = v
    return dw


def emu_wgrad_pq(op, dy, src, dw, accumulate=False):
    """The position-pair weight gradient (csrc/wgrad_pq.hip) walked over the device tables of ops.MaskedGemm.pq(): per tile
    G = dy rows x source rows over the batch, contraction with m[pq][k], masked scatter to the children.  dy, src: numpy
    [rows, bp]; dw dense (n_rows, n_chan, kt)."""
    t = op.pq()
    assert t is not None
    T, S = t.T, t.S
    RT, CT = 128 // T, 128 // S
    m = t.m_host.reshape(T, S, 16).astype(np.float64)
    ok = t.ok_host
    dzb, xb = _np(t.dz_base), _np(t.x_base)
    ctab = None if t.child_tab is None else _np(t.child_tab).reshape(-1, 4)
    scale = None if t.child_scale is None else _np(t.child_scale)
    n_rows, n_chan, kt = op.w_shape
    dwf = dw.reshape(n_rows, n_chan, kt)
    seen = set()
    for r0, c0 in zip(_np(t.tile_r0), _np(t.tile_c0)):
        assert r0 % RT == 0 and c0 % CT == 0 and (r0, c0) not in seen
        seen.add((r0, c0))
        rows = list(range(r0, min(r0 + RT, t.n_rows)))
        chans = list(range(c0, min(c0 + CT, t.n_chan)))
        A = np.stack([dy[dzb[r] + p * t.dz_ts] for r in rows for p in range(T)]).astype(np.float64)
        Bm = np.stack([src[xb[c] + q * t.x_ts] for c in chans for q in range(S)]).astype(np.float64)
        G = (A @ Bm.T).reshape(len(rows), T, len(chans), S)
        Sk = np.einsum("rpcq,pqk->rck", G, m)
        for ri, r in enumerate(rows):
            for ci, c in enumerate(chans):
                bits = int(ok[r, c])
                if not bits:
                    continue
                v = Sk[ri, ci, :kt]
                if ctab is None:
                    dwf[r, c, :] = (dwf[r, c, :] if accumulate else 0) + v
                    continue
                v = v * (float(scale[r if t.child_side else c]) if scale is not None else 1.0)
                for k in range(4):
                    if (bits >> k) & 1:
                        dr, dc = (ctab[r, k], c) if t.child_side else (r, ctab[c, k])
                        dwf[dr, dc, :] = (dwf[dr, dc, :] if accumulate else 0) + v
    # tiles that are not listed hold no live pair
    for r0 in range(0, t.n_rows, RT):
        for c0 in range(0, t.n_chan, CT):
            assert (r0, c0) in seen or not ok[r0:r0 + RT, c0:c0 + CT].any()
    return dw


# ---- fourth-generation gather-GEMM (csrc/gemm_v4.hip): bf16 planes in A-operand order, K steps of 32 channels ----
def bf16_split3(x):
    """x (float32 array) -> three uint16 arrays of bf16 bit patterns with x == p0 + p1 + p2 exactly (round to nearest
    even at each level), as split2() does on the device."""
    planes = []
    r = np.ascontiguousarray(x, dtype=np.float32)
    for _ in range(3):
        u = r.view(np.uint32).astype(np.uint64)
        h = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint32)
        planes.append(h.astype(np.uint16))
        r = (r - (h << 16).astype(np.uint32).view(np.float32)).astype(np.float32)
    assert not r.any()
    return planes


def bf16_value(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def emu_pack4(side, dense):
    """hmvae_pack_weights_v4: float32 [fragment][h][lane (kk*16 + m)][4] (channel 8kk + 4h + j), fragment = (c32*kt + vt)*mtg
    + tile at the group's offset g_woff4 (in KB; a fragment is 2 KB).  `side` is a layout-4 companion (ops._Side.v4_side)."""
    g = side.g
    assert side.layout == 4
    # the effective matrix per (group, row, listed channel, vt) comes from the v2 emulation's own arithmetic: pack the
    # SAME tables in the v2 order and re-address
    wp2 = emu_pack(side, dense)
    woff4 = _np(side.g_woff4)
    mtg = g.mt
    out = np.zeros(max(int(side.wp.numel()) // 2, 1), dtype=np.float32).reshape(-1, 2, 64, 4)
    for gi in range(g.n_groups):
        nc4 = int(g.g_nc4[gi])
        W = wp2[g.g_woff[gi]:g.g_woff[gi] + nc4 * g.kt * mtg * 64].reshape(nc4, g.kt, mtg, 4, 16)   # [c4][vt][tile][kk4][m]
        for c32 in range(nc4 // 8):
            for vt in range(g.kt):
                for tile in range(mtg):
                    blk = np.zeros((4, 16, 8), dtype=np.float32)              # [kk][m][j]: channel c32*32 + kk*8 + j
                    for kk in range(4):
                        for j in range(8):
                            c = c32 * 32 + kk * 8 + j
                            blk[kk, :, j] = W[c // 4, vt, tile, c % 4, :]
                    out[int(woff4[gi]) // 2 + (c32 * g.kt + vt) * mtg + tile] = blk.reshape(64, 2, 4).transpose(1, 0, 2)
    return out


def emu_gather4(side, wp4, src, dst, nc=64, split=1, bias=None, leaky=False, accum=False, slope=0.2):
    """hmvae_gather_gemm_v4 on numpy [rows, bp] operands (wp4 from emu_pack4): units (group, position, nc columns), K steps (c32, item) with the
    item fastest, cut into `split` parts that are summed in part order."""
    g = side.g
    mtg = g.mt
    R = mtg * 16
    chan_rows, dst_rows = _np(side.chan_rows), _np(side.dst_rows)
    bias_idx = _np(side.bias_idx)
    bias_idx2 = None if getattr(side, "bias_idx2", None) is None else _np(side.bias_idx2)
    pb_off, pb_len, items, n_pblk, _ = side.items_for(1)
    pb_off, pb_len, items = _np(pb_off), _np(pb_len), _np(items).reshape(-1, 2)
    woff4, ntiles = _np(side.g_woff4), _np(side.g_ntiles)
    bp = src.shape[1]
    assert bp % nc == 0 and n_pblk == side.n_pos
    for gi in range(g.n_groups):
        nc32 = int(g.g_nc4[gi]) // 8
        rows_src = chan_rows[g.g_choff[gi]:g.g_choff[gi] + nc32 * 32]
        for p in range(side.n_pos):
            its = items[pb_off[p]:pb_off[p] + pb_len[p]]
            nks = nc32 * len(its)
            for cb in range(bp // nc):
                cols = slice(cb * nc, (cb + 1) * nc)
                total = np.zeros((R, nc), dtype=np.float64)
                for sp in range(split):
                    part = np.zeros((R, nc), dtype=np.float64)
                    for ks in range(nks * sp // split, nks * (sp + 1) // split):
                        c32, it = divmod(ks, len(its))
                        q, vt = int(its[it, 0]), int(its[it, 1]) & 0xff
                        B = src[rows_src[c32 * 32:(c32 + 1) * 32] + q * side.src_ts][:, cols].astype(np.float64)   # [32 k][nc]
                        for tile in range(int(ntiles[gi])):
                            frag = wp4[int(woff4[gi]) // 2 + (c32 * g.kt + vt) * mtg + tile]      # [h][lane][4] fp32
                            frag = np.ascontiguousarray(frag.transpose(1, 0, 2)).reshape(64, 8)
                            planes = bf16_split3(frag)                # what the owning wave does in registers: exact
                            A = sum(bf16_value(pl) for pl in planes)
                            A = A.reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)            # [m][k = kk*8 + j]
                            part[tile * 16:(tile + 1) * 16] += A @ B
                    total += part
                for r in range(int(ntiles[gi]) * 16):
                    dr = dst_rows[gi * R + r]
                    if dr < 0:
                        continue
                    v = total[r]
                    if bias is not None and bias_idx[gi * R + r] >= 0:
                        bv = bias[bias_idx[gi * R + r]]
                        if bias_idx2 is not None and bias_idx2[gi * R + r] >= 0:
                            bv = 0.5 * (bv + bias[bias_idx2[gi * R + r]])
                        v = v + bv
                    if leaky:
                        v = np.where(v > 0, v, v * slope)
                    row = dr + p * side.dst_ts
                    dst[row, cols] = (dst[row, cols] + v) if accum else v
    return dst
