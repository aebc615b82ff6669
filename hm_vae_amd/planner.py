"""Host-side planning for the MFMA kernels: row groups, channel lists, position/tap tables.

Everything here runs once per model (numpy, CPU); the results are small int32 tables that live on the
device next to the weights.  The kernels that consume them are in csrc/gemm_kernels.hip; the layout is
specified in include/hmvae.h (hmvae_gemm_desc / hmvae_wgrad_desc).

A *group* is up to 16*mt output rows that share one list of input channels (the union of the neighbour
channels of the joints whose rows it holds — skeleton.py:34-39).  Grouping decides how much of every
16x16x4 MFMA is useful work:  cost = (16*mt) * padded_channels summed over groups.
"""
import numpy as np


class GroupSet:
    """Row groups + channel lists for one masked dense matrix W[row, chan, tap]."""

    def __init__(self, mt, groups, kt, row_chans=None, pair_bits=None, chan_pad=4):
        """groups: [(rows, chans)]; row_chans: dense row id -> set of channels that row may read (its mask).
        When rows of several blocks share a group, the union list holds channels that are masked out for some
        rows: pair_ok marks the (row, channel) pairs that are really in the mask.  pair_bits(row, chan) -> uint8
        replaces the 0/1 entry by a bit mask over the dense (row, channel) pairs merged into this pair (composite
        relations, see compose_conv)."""
        self.mt = int(mt)
        self.kt = int(kt)
        self.n_groups = len(groups)
        rows_per = 16 * self.mt
        row_id = np.full((self.n_groups, rows_per), -1, dtype=np.int32)
        chan_id, g_choff, g_nc4, g_woff, g_nch = [], [], [], [], []
        g_okoff, ok_parts = [], []
        okoff = 0
        woff = 0
        for gi, (rows, chans) in enumerate(groups):
            assert 0 < len(rows) <= rows_per, (len(rows), rows_per)
            row_id[gi, :len(rows)] = rows
            nch = len(chans)
            pad = (-nch) % int(chan_pad)
            g_choff.append(len(chan_id))
            chan_id.extend(list(chans) + [-1] * pad)
            nc4 = (nch + pad) // 4
            g_nc4.append(nc4)
            g_nch.append(nch)
            g_woff.append(woff)
            woff += nc4 * self.kt * self.mt * 64
            ok = np.zeros((rows_per, nc4 * 4), dtype=np.uint8)
            for ri, r in enumerate(rows):
                if row_chans is None:
                    ok[ri, :nch] = 1
                elif pair_bits is not None:
                    allowed = row_chans[r]
                    ok[ri, :nch] = [pair_bits(r, c) if c in allowed else 0 for c in chans]
                else:
                    allowed = row_chans[r]
                    ok[ri, :nch] = [1 if c in allowed else 0 for c in chans]
            g_okoff.append(okoff)
            okoff += ok.size
            ok_parts.append(ok.reshape(-1))
        self.row_id = row_id.reshape(-1)
        self.chan_id = np.asarray(chan_id, dtype=np.int32)
        self.g_choff = np.asarray(g_choff, dtype=np.int32)
        self.g_nc4 = np.asarray(g_nc4, dtype=np.int32)
        self.g_nch = np.asarray(g_nch, dtype=np.int32)
        self.g_woff = np.asarray(g_woff, dtype=np.int32)
        self.packed_size = int(woff)
        self.g_okoff = np.asarray(g_okoff, dtype=np.int32)
        self.pair_ok = np.concatenate(ok_parts) if ok_parts else np.zeros(1, np.uint8)
        assert woff < 2 ** 31

    def mfma_cost(self):
        return int((self.g_nc4 * 4).sum()) * 16 * self.mt

    def useful(self):
        """(row, chan) pairs that are real work."""
        return int((self.pair_ok != 0).sum())

    # wgrad tiling: 64 (channel,tap) columns per tile
    def wgrad_tiles(self):
        tile_g, tile_col0 = [], []
        for gi in range(self.n_groups):
            ncols = int(self.g_nch[gi]) * self.kt
            for c0 in range(0, ncols, 64):
                tile_g.append(gi)
                tile_col0.append(c0)
        return (np.asarray(tile_g, dtype=np.int32), np.asarray(tile_col0, dtype=np.int32),
                (self.g_nch * self.kt).astype(np.int32))


def _merge_units(units, cap, chan_pad=4):
    """Greedy agglomeration: merge two units when rows fit and the summed (padded) channel lists shrink."""
    units = [(list(r), set(c)) for r, c in units]
    while True:
        best = None
        for i in range(len(units)):
            for j in range(i + 1, len(units)):
                if len(units[i][0]) + len(units[j][0]) > cap:
                    continue
                u = units[i][1] | units[j][1]
                pad4 = lambda n: -(-n // chan_pad) * chan_pad
                gain = pad4(len(units[i][1])) + pad4(len(units[j][1])) - pad4(len(u))
                if gain > 0 and (best is None or gain > best[0]):
                    best = (gain, i, j)
        if best is None:
            break
        _, i, j = best
        merged = (units[i][0] + units[j][0], units[i][1] | units[j][1])
        units = [u for k, u in enumerate(units) if k not in (i, j)] + [merged]
    return units


def make_groups(joint_rows, joint_chans, kt, mt_choices=(1,), pair_bits=None):
    """joint_rows[j]: dense row ids owned by block j; joint_chans[j]: dense channel ids block j reads.

    Default: one 16-row tile per group.  Measured on MI355X at B=256 (tools/profile_layers.py with HMVAE_MT=1..4):
    16-row groups win on every layer for the pipelined gather kernel — they give 3-4x more workgroups (the small-T
    layers otherwise put ~1 workgroup on a CU) and never pay for a union channel list."""
    import os
    if os.environ.get("HMVAE_MT"):
        mt_choices = tuple(int(c) for c in os.environ["HMVAE_MT"].split(","))
    best = None
    row_chans = {}
    for rows, chans in zip(joint_rows, joint_chans):
        fs = frozenset(chans)
        for r in rows:
            row_chans[r] = fs
    # larger tiles reuse the activation operand more; bias the choice slightly towards them
    eff = {1: 1.10, 2: 1.0, 3: 0.98, 4: 0.97}
    for mt in mt_choices:
        cap = 16 * mt
        units = []
        for rows, chans in zip(joint_rows, joint_chans):
            rows = list(rows)
            n_chunks = -(-len(rows) // cap)
            for k in range(n_chunks):
                units.append((rows[k * cap:(k + 1) * cap], list(chans)))
        units = _merge_units(units, cap)
        groups = [(r, sorted(c)) for r, c in units]
        groups.sort(key=lambda g: min(g[0]))
        if os.environ.get("HMVAE_GROUP_ORDER", "rows") == "len":
            # longest reduction first: the dispatcher hands out workgroups in index order, so the long groups of a layer
            # (K lengths differ up to 4x) start early and the short ones fill the tail
            groups.sort(key=lambda g: -len(g[1]))
        gs = GroupSet(mt, groups, kt, row_chans, pair_bits)
        score = gs.mfma_cost() * eff[mt]
        if best is None or score < best[0]:
            best = (score, gs)
    return best[1]


def make_groups_v4(joint_rows, joint_chans, kt, cap, pair_bits=None):
    """Row groups of the bf16-plane gather-GEMM (hmvae_gather_gemm_v4): up to `cap` rows (64 or 128 = four waves x one or
    two 16-row tiles) sharing one channel list padded to 32.  A workgroup spends mtw x (32-channel steps) matrix time on a
    group however many of its tiles are live, so rows with IDENTICAL channel lists are pooled across blocks before they are
    cut (a dense layer becomes ceil(rows / cap) full groups), and groups are merged while the padded lists shrink."""
    pools, order = {}, []
    row_chans = {}
    for rows, chans in zip(joint_rows, joint_chans):
        fs = frozenset(chans)
        if fs not in pools:
            pools[fs] = []
            order.append(fs)
        pools[fs].extend(rows)
        for r in rows:
            row_chans[r] = fs
    units = []
    for fs in order:
        rows = pools[fs]
        n_chunks = -(-len(rows) // cap)
        per = -(-len(rows) // n_chunks)
        per = -(-per // 16) * 16                      # whole tiles, evenly filled groups
        for k in range(0, len(rows), per):
            units.append((rows[k:k + per], list(fs)))
    units = _merge_units(units, cap, chan_pad=32)
    groups = [(r, sorted(c)) for r, c in units]
    groups.sort(key=lambda g: min(g[0]))
    return GroupSet(cap // 16, groups, kt, row_chans, pair_bits, chan_pad=32)


# ---------------------------------------------------------------------------------------------
# masked conv (SkeletonConv) structure
# ---------------------------------------------------------------------------------------------
def conv_joint_lists(neigh, cin, cout):
    """Per output block i: rows [co*i, co*(i+1)), channels = expanded neighbour list (skeleton.py:34-39)."""
    n = len(neigh)
    ci, co = cin // n, cout // n
    rows = [list(range(co * i, co * (i + 1))) for i in range(n)]
    chans = [[k * ci + c for k in nb for c in range(ci)] for nb in neigh]
    return rows, chans


def conv_joint_lists_transposed(neigh, cin, cout):
    """For backward-data: rows are INPUT channels of block k, channels are OUTPUT channels of every block i
    that lists k as a neighbour."""
    n = len(neigh)
    ci, co = cin // n, cout // n
    rows = [list(range(ci * k, ci * (k + 1))) for k in range(n)]
    chans = [[] for _ in range(n)]
    for i, nb in enumerate(neigh):
        for k in nb:
            chans[k].extend(range(co * i, co * (i + 1)))
    return rows, [sorted(c) for c in chans]


def make_wgrad_groups(rows_b, chans_b, kt, mt_choices=(4, 3, 2, 1), pair_bits=None):
    """Weight-gradient tiling keyed by INPUT block: columns = the block's channels x taps, rows = every output row
    that reads the block (all of them read all of its channels, so there is no union waste), cut into 16*mt-row
    pieces.  rows_b / chans_b are the transposed lists (conv_joint_lists_transposed): rows_b[k] = channels of input
    block k, chans_b[k] = output rows that read it.  Returns the GroupSet with the fewest MFMA tile-rows."""
    import os
    if os.environ.get("HMVAE_WGRAD_MT"):
        mt_choices = tuple(int(c) for c in os.environ["HMVAE_WGRAD_MT"].split(","))
    best = None
    for mt in mt_choices:
        cap = 16 * mt
        groups = []
        for in_ch, out_rows in zip(rows_b, chans_b):
            out_rows = list(out_rows)
            n_chunks = -(-len(out_rows) // cap)
            # even split keeps the last piece from being nearly empty
            per = -(-len(out_rows) // n_chunks)
            per = -(-per // 16) * 16 if per % 16 else per
            for k in range(0, len(out_rows), per):
                groups.append((out_rows[k:k + per], list(in_ch)))
        row_chans = None
        if pair_bits is not None:           # merged ids: a row reads a listed channel through some of its children only
            row_chans = {}
            for out_rows, in_ch in groups:
                for r in out_rows:
                    row_chans[r] = row_chans.get(r, frozenset()) | frozenset(c for c in in_ch if pair_bits(r, c))
        gs = GroupSet(mt, groups, kt, row_chans, pair_bits)
        tiles = sum(-(-int(n) * kt // 64) for n in gs.g_nch)
        cost = tiles * mt * (1.0 + 0.35 / mt)          # fewer, fatter tiles reuse the activation operand more
        if best is None or cost < best[0]:
            best = (cost, gs)
    return best[1]


def reflect(u, n):
    if u < 0:
        u = -u
    if u >= n:
        u = 2 * (n - 1) - u
    return u


def conv_pairs_fwd(t_in, t_out, kt, stride, pad, pad_mode):
    """pairs[p] = [(q, tap)]: source position q feeding output position p through tap."""
    out = []
    for p in range(t_out):
        lst = []
        for tap in range(kt):
            u = p * stride + tap - pad
            if 0 <= u < t_in:
                lst.append((u, tap))
            elif pad_mode == "reflect":
                lst.append((reflect(u, t_in), tap))
        out.append(lst)
    return out


def conv_pairs_bwd(t_in, t_out, kt, stride, pad, pad_mode):
    """pairs[tau] = [(t, tap)]: output position t whose tap reads input position tau (reflections folded in)."""
    out = [[] for _ in range(t_in)]
    for t in range(t_out):
        for tap in range(kt):
            u = t * stride + tap - pad
            if 0 <= u < t_in:
                out[u].append((t, tap))
            elif pad_mode == "reflect":
                out[reflect(u, t_in)].append((t, tap))
    return out


def flatten_pairs(pairs):
    p_off, p_len, flat = [], [], []
    for lst in pairs:
        p_off.append(len(flat))
        p_len.append(len(lst))
        flat.extend(q | (tap << 16) for q, tap in lst)
    if not flat:
        flat = [0]
    return (np.asarray(p_off, dtype=np.int32), np.asarray(p_len, dtype=np.int32), np.asarray(flat, dtype=np.int32))


def pick_split(n_tiles, k_chunks, target_blocks=1024, max_split=64):
    """Split of the (position, batch) reduction of the weight-gradient GEMM across workgroups."""
    s = max(1, min(max_split, -(-target_blocks // max(n_tiles, 1))))
    return int(max(1, min(s, max(1, k_chunks // 4))))


def build_items(pairs, np_):
    """Group consecutive output positions in blocks of np_ and merge their (q, tap) pairs by source position q.

    Returns (pb_off, pb_len, items) with items as int32 pairs [q, taps]; taps packs one tap per owned position,
    8 bits each, 0xff = unused.  A q that a position reads through two taps (reflection) yields two items.
    """
    n_pos = len(pairs)
    pb_off, pb_len, flat = [], [], []
    for p0 in range(0, n_pos, np_):
        by_q = {}
        for pl in range(np_):
            if p0 + pl >= n_pos:
                break
            for q, tap in pairs[p0 + pl]:
                by_q.setdefault(q, [[] for _ in range(np_)])[pl].append(tap)
        pb_off.append(len(flat) // 2)
        cnt = 0
        for q in sorted(by_q):
            lists = by_q[q]
            while any(lists):
                word = 0
                for pl in range(np_):
                    tap = lists[pl].pop(0) if lists[pl] else 0xff
                    word |= (tap & 0xff) << (8 * pl)
                for pl in range(np_, 4):
                    word |= 0xff << (8 * pl)
                if word >= 2 ** 31:
                    word -= 2 ** 32
                flat.extend((q, word))
                cnt += 1
        pb_len.append(cnt)
    if not flat:
        flat = [0, -1]
    return (np.asarray(pb_off, dtype=np.int32), np.asarray(pb_len, dtype=np.int32), np.asarray(flat, dtype=np.int32))


def _pp_blocks(n_pos, pp, pstep):
    """Blocks of `pp` output positions `pstep` apart (residue classes of pstep, cut into runs of pp)."""
    blocks = []
    for r in range(min(pstep, n_pos)):
        cls = list(range(r, n_pos, pstep))
        for i in range(0, len(cls), pp):
            blocks.append(cls[i:i + pp])
    return blocks


def _pp_merge(pairs, block):
    """Steps of one block: one per (tap, occurrence) of the union of the positions' tap multisets, tap ascending; each
    step lists the source position of every slot (None = the position does not use this tap occurrence)."""
    per = []
    for p in block:
        by_tap = {}
        for q, tap in sorted(pairs[p], key=lambda it: (it[1], it[0])):
            by_tap.setdefault(tap, []).append(q)
        per.append(by_tap)
    steps = []
    for tap in sorted(set(t for bt in per for t in bt)):
        for k in range(max(len(bt.get(tap, ())) for bt in per)):
            steps.append((tap, [bt[tap][k] if k < len(bt.get(tap, ())) else None for bt in per]))
    return steps


def _pp_slots(steps, pp, n_img):
    """Schedule of one block for the v4 kernel with position blocks: the activation rows of a 32-channel step live in
    n_img LDS images; every emitted step stages AT MOST ONE image (the register ring of the kernel carries one image per
    step) and multiplies the block's positions with the images that hold their rows.  Returns the emitted steps
    [(tap | 0xff, load_q | -1, load_slot, [slot per position | 0xf])] or None when n_img images are not enough.

    Rules: an image staged in step e is readable from step e on (stash -> barrier -> multiply) and must not be one that step
    e - 1 multiplies with (other waves may still be in e - 1) nor one a later step still needs; a step whose rows are not all
    present after its own load becomes a bubble (tap 0xff: it only stages)."""
    uses = {}                                    # row -> compute steps that read it
    for i, (_, qs) in enumerate(steps):
        for q in qs:
            if q is not None:
                uses.setdefault(q, []).append(i)
    loaded, slot_row = {}, [None] * n_img
    out, i, prev_read = [], 0, set()

    def next_use(r, i_now):
        for u in uses[r]:
            if u >= i_now:
                return u
        return None

    def free_slot(i_now, required, before=None):
        """An image for a new row at compute step i_now: an empty one, else one whose row is not read again, else --
        for a row the step cannot do without -- the one whose row is read again latest (it is staged again then); a
        look-ahead load only displaces rows that are read again later than the row it brings (`before`)."""
        cur = set(q for q in steps[i_now][1] if q is not None)
        best, best_u = None, -1
        for s in range(n_img):
            r = slot_row[s]
            if s in prev_read or (r is not None and r in cur):
                continue
            if r is None:
                return s
            u = next_use(r, i_now)
            if u is None:
                return s
            if u > best_u:
                best, best_u = s, u
        if best is None:
            return None
        if required or (before is not None and best_u > before):
            return best
        return None

    guard = 0
    while i < len(steps):
        guard += 1
        if guard > 4 * len(steps) + 64:
            return None
        tap, qs = steps[i]
        need = []
        for q in qs:
            if q is not None and q not in loaded and q not in need:
                need.append(q)
        load_q, load_s = -1, 0
        if need:
            s = free_slot(i, True)
            if s is None:
                return None
            load_q, load_s = need[0], s
        else:                                    # look ahead: the row a coming step misses first, if an image is free
            for i2 in range(i + 1, min(len(steps), i + 1 + n_img)):
                miss = [q for q in steps[i2][1] if q is not None and q not in loaded]
                if miss:
                    s = free_slot(i, False, before=i2)
                    if s is not None:
                        load_q, load_s = miss[0], s
                    break
        if load_q >= 0:
            old = slot_row[load_s]
            if old is not None:
                del loaded[old]
            slot_row[load_s] = load_q
            loaded[load_q] = load_s
        if len(need) <= 1:                       # every row of the step is present now: multiply
            slots = [0xf if q is None else loaded[q] for q in qs]
            out.append((tap, load_q, load_s, slots + [0xf] * (4 - len(slots))))
            prev_read = set(x for x in slots if x != 0xf)
            i += 1
        else:
            out.append((0xff, load_q, load_s, [0xf] * 4))
            prev_read = set()
    return out


def pick_np_ksplit(mt, n_groups, n_pos, n_col_tiles, avg_nc4, target_waves=2048):
    """(64-column units per wave, K split over the waves of a workgroup) for the gather-GEMM.

    Measured on MI355X (round 1, B=256).  fp32 MFMA shares the SIMD's fp32 datapath with ordinary vector
    instructions, so the more MFMAs a wave issues per item (mt*cw*4) the smaller the share lost to addressing; but
    the grid must still put a couple of waves on each of the 1024 SIMDs.
    """
    ks = 4 if avg_nc4 >= 8 else (2 if avg_nc4 >= 4 else 1)
    cw = 1
    if mt == 1 and n_col_tiles % 2 == 0 and n_groups * n_pos * (n_col_tiles // 2) * ks >= 5000:
        cw = 2
    return cw, ks


def _materialized_windows(pairs, kt, chunk, max_win):
    """Forward-convolution form: window row pi*s + k of a chunk holds the source position that output position
    p0 + pi reads through tap k (padding already applied), so every wave's items are (pi*s + k, k), k = 0..kt-1.
    Returns (stride, win_rows[n_chunks][win]) or None when the relation is not of that form."""
    n_pos = len(pairs)
    if any(sorted(t for _, t in lst) != list(range(kt)) for lst in pairs):
        return None
    n_chunks = -(-n_pos // chunk)
    for s in (1, 2, 3, 4):
        win = (chunk - 1) * s + kt
        if win > max_win:
            return None
        rows = -np.ones((n_chunks, win), dtype=np.int64)
        ok = True
        for c in range(n_chunks):
            for pi in range(chunk):
                p = c * chunk + pi
                if p >= n_pos:
                    break
                for q, t in pairs[p]:
                    j = pi * s + t
                    if rows[c, j] >= 0 and rows[c, j] != q:
                        ok = False
                    rows[c, j] = q
            if not ok:
                break
        if ok:
            rows[rows < 0] = 0            # rows nobody reads: any valid position
            return s, rows
    return None


def build_window_plan(pairs, kt, chunk, ts, un, n_waves=8, max_items=16, max_win=32):
    """Tables of the window gather-GEMM (hmvae_gather_gemm_v3) for one position relation.

    pairs[p] = [(q, tap)] as for v2.  A workgroup takes `chunk` neighbouring output positions; `ts` waves share one
    position (contiguous slices of its item list), chunk * ts == 8.  win_rows[c][j] is the source position staged in
    window row j of chunk c.  Forward convolutions get windows with the padding materialised (every wave then walks
    rows r0..r0+kt-1 with taps 0..kt-1: pattern 1, static LDS offsets in the kernel); any other relation gets the
    contiguous range of source positions the chunk reads and explicit (row, tap) items; a wave whose items are
    (r0 + i, kt-1-i) is pattern 2 (the interior of a transposed stride-1 convolution).  Item counts are padded to a
    multiple of `un` with (row 0, tap kt) entries: tap kt is the all-zero weight fragment.  Returns None when the
    relation does not fit the kernel's limits (window rows, items per wave).
    """
    assert chunk * ts == n_waves
    n_pos = len(pairs)
    n_chunks = -(-n_pos // chunk)
    q_all = [q for lst in pairs for q, _ in lst]
    if not q_all or kt > 64:                     # the kernel stages <= 64 taps' weight fragments per channel step
        return None
    q_hi = max(q_all)
    mat = _materialized_windows(pairs, kt, chunk, max_win) if (ts == 1 and kt <= max_items) else None
    if mat is not None:
        stride, win_rows = mat
        win = win_rows.shape[1]
    else:
        spans, win = [], 4
        for c in range(n_chunks):
            qs = [q for p in range(c * chunk, min(n_pos, (c + 1) * chunk)) for q, _ in pairs[p]]
            lo, hi = (min(qs), max(qs)) if qs else (0, 0)
            spans.append(lo)
            win = max(win, hi - lo + 1)
        if win > max_win or win > q_hi + 1:
            return None
        win_rows = np.zeros((n_chunks, win), dtype=np.int64)
        for c in range(n_chunks):
            win_rows[c] = max(0, min(spans[c], q_hi + 1 - win)) + np.arange(win)
    wave_tab = np.zeros((n_chunks, n_waves, 4), dtype=np.int32)
    items = np.full((n_chunks, n_waves, 32), kt << 16, dtype=np.int32)
    useful, issued, max_n = 0, 0, un
    for c in range(n_chunks):
        q0 = int(win_rows[c, 0])
        for pi in range(chunk):
            p = c * chunk + pi
            lst = sorted(pairs[p], key=lambda it: (it[1], it[0])) if p < n_pos else []
            for ti in range(ts):
                piece = lst[len(lst) * ti // ts: len(lst) * (ti + 1) // ts]
                if len(piece) > max_items:
                    return None
                if mat is not None:
                    rel = [(pi * stride + t, t) for _, t in piece]
                else:
                    rel = [(q - q0, t) for q, t in piece]
                pat = 0
                if len(rel) == kt and un == 5 and kt == 15:
                    if all(r == rel[0][0] + i and t == i for i, (r, t) in enumerate(rel)):
                        pat = 1
                    else:
                        by_row = sorted(rel)
                        if all(r == by_row[0][0] + i and t == kt - 1 - i for i, (r, t) in enumerate(by_row)):
                            pat, rel = 2, by_row
                n_pad = -(-len(rel) // un) * un
                wave_tab[c, pi * ts + ti] = (p if p < n_pos else -1, pat, n_pad, 0)
                for i, (r, t) in enumerate(rel):
                    assert 0 <= r < win
                    items[c, pi * ts + ti, i] = r | (t << 16)
                useful += len(rel)
                issued += n_pad
                max_n = max(max_n, n_pad)
    chunk_cost = wave_tab[:, :, 2].max(axis=1).astype(np.int32)        # items of the slowest wave, per chunk
    # chunks whose waves all run the static-offset path are ~12 % cheaper per item (measured)
    all_static = ((wave_tab[:, :, 1] > 0) | (wave_tab[:, :, 0] < 0)).all(axis=1)
    cost = float((chunk_cost * np.where(all_static, 0.88, 1.0)).sum()) * n_waves
    return dict(wave_tab=wave_tab.reshape(-1), items=items.reshape(-1), win_rows=win_rows.astype(np.int32).reshape(-1),
                win=int(win), chunk_cost=chunk_cost, n_chunks=int(n_chunks), ts=int(ts), un=int(un), max_items=int(max_n),
                useful=useful, issued=issued, cost=cost, n_static=int((wave_tab[:, :, 1] > 0).sum()))


def pick_window_un(pairs, kt, chunk, ts):
    """Unroll (items per chunk of the v3 inner loop) with the least padding for this relation."""
    best = None
    for un in (5, 4, 3):
        plan = build_window_plan(pairs, kt, chunk, ts, un)
        if plan is None:
            return None
        if best is None or plan["cost"] < best["cost"]:
            best = plan
    return best


def build_stream_schedule(unit_steps, unit_cost, n_workers, min_steps=3):
    """Cut a list of units into `n_workers` runs of equal cost ("stream-K").

    unit_steps[u]: K steps of unit u (in launch order); unit_cost[u]: cost of one step of it (items of its slowest
    wave).  Returns (segs, worker_seg, fixups, n_parts): segs rows are [u, k0, k1, part] (part -1 = whole unit),
    worker_seg rows [first seg, n segs], fixups rows [u, first part, n parts].  A cut closer than `min_steps` to a
    unit's edge snaps to the edge.
    """
    unit_steps = np.asarray(unit_steps, dtype=np.int64)
    unit_cost = np.asarray(unit_cost, dtype=np.int64)
    total = int((unit_steps * unit_cost).sum())
    segs, worker_seg = [], []
    u, k = 0, 0                      # next unit / next step inside it
    n_units = len(unit_steps)
    done = 0
    for w in range(n_workers):
        target = total * (w + 1) // n_workers
        first = len(segs)
        while u < n_units and done < target:
            c = int(unit_cost[u])
            left = int(unit_steps[u]) - k
            if c == 0 or left <= 0:
                if left > 0:
                    segs.append([u, k, k + left, -1])
                u, k = u + 1, 0
                continue
            take = min(left, -(-(target - done) // c))
            if left - take < min_steps or w == n_workers - 1:
                take = left                                   # do not leave a sliver behind
            elif take < min_steps:
                if k == 0:
                    break                                     # a sliver at the head of a unit: leave it to the next worker
                take = min(left, min_steps)
            segs.append([u, k, k + take, -1])
            done += take * c
            k += take
            if k >= int(unit_steps[u]):
                u, k = u + 1, 0
        worker_seg.append([first, len(segs) - first])
    assert u == n_units, (u, n_units)
    # parts of cut units
    fixups, n_parts = [], 0
    by_unit = {}
    for i, sg in enumerate(segs):
        by_unit.setdefault(sg[0], []).append(i)
    for uu in sorted(by_unit):
        idx = by_unit[uu]
        if len(idx) > 1:
            fixups.append([uu, n_parts, len(idx)])
            for i in idx:
                segs[i][3] = n_parts
                n_parts += 1
    return (np.asarray(segs, dtype=np.int64).reshape(-1, 4), np.asarray(worker_seg, dtype=np.int64).reshape(-1, 2),
            np.asarray(fixups, dtype=np.int64).reshape(-1, 3), n_parts)




# ---------------------------------------------------------------------------------------------
# composite relations: convolution o [x2 linear upsample] o [SkeletonUnpool], tap folding
# ---------------------------------------------------------------------------------------------
def upsample_rows(t_coarse):
    """nn.Upsample(scale_factor=2, mode='linear', align_corners=False): fine position t -> [(coarse position, weight)]."""
    out = []
    for t in range(2 * t_coarse):
        i = t >> 1
        if t & 1:
            a, b, wa, wb = i, min(i + 1, t_coarse - 1), 0.75, 0.25
        else:
            a, b, wa, wb = max(i - 1, 0), i, 0.25, 0.75
        out.append([(a, 1.0)] if a == b else [(a, wa), (b, wb)])
    return out


def compose_conv(pairs_fine, up=None, max_nf=16, max_vt=254):
    """Fold a temporal convolution (and the x2 linear upsample in front of it) into VIRTUAL TAPS.

    pairs_fine[p] = [(t, k)]: output position p reads (upsampled) position t through tap k.  With `up`
    (upsample_rows) every t is a weighted pair of source positions, so per output position the layer is a map over few
    source positions q with effective weights  Weff[p][q] = sum_k c(p,q,k) W[..,k].  Output positions whose (tap,
    coefficient) lists coincide share a virtual tap.  Returns (pairs_eff, vt_taps, vt_coefs) with
    pairs_eff[p] = [(q, vt)], vt_taps/vt_coefs [n_vt][nf] (tap -1 = unused), or None when the lists are too long.
    """
    pat_id, pats, pairs_eff = {}, [], []
    for lst in pairs_fine:
        acc = {}
        for t, k in lst:
            for q, c in (up[t] if up is not None else [(t, 1.0)]):
                d = acc.setdefault(q, {})
                d[k] = d.get(k, 0.0) + c
        eff = []
        for q in sorted(acc):
            key = tuple(sorted((k, round(c, 9)) for k, c in acc[q].items()))
            if key not in pat_id:
                pat_id[key] = len(pats)
                pats.append(key)
            eff.append((q, pat_id[key]))
        pairs_eff.append(eff)
    nf = max(len(k) for k in pats) if pats else 1
    if nf > max_nf or len(pats) > max_vt:
        return None
    nf = max(4, -(-nf // 4) * 4)
    taps = -np.ones((len(pats), nf), dtype=np.int32)
    coefs = np.zeros((len(pats), nf), dtype=np.float32)
    for i, key in enumerate(pats):
        for j, (k, c) in enumerate(key):
            taps[i, j], coefs[i, j] = k, c
    return pairs_eff, taps, coefs


def transpose_pairs(pairs_eff, n_src):
    """pairs_b[q] = [(p, vt)] for every (q, vt) in pairs_eff[p]."""
    out = [[] for _ in range(n_src)]
    for p, lst in enumerate(pairs_eff):
        for q, vt in lst:
            out[q].append((p, vt))
    return out


def merge_channels(joint_rows, joint_chans, cmap):
    """Forward lists with the fine input channels replaced by the coarse channels they are copies of
    (cmap[fine] = coarse, SkeletonUnpool folded into the weights).  Returns (rows, chans_coarse, pair_bits) where
    pair_bits(row, coarse) has bit i set when child i of `coarse` (children[coarse][i]) is in the row's mask."""
    children = {}
    for f, cc in enumerate(cmap):
        children.setdefault(int(cc), []).append(f)
    row_fine = {}
    for rows, chans in zip(joint_rows, joint_chans):
        fs = frozenset(chans)
        for r in rows:
            row_fine[r] = fs
    chans_c = [sorted(set(int(cmap[f]) for f in chans)) for chans in joint_chans]

    def bits(r, cc):
        allowed = row_fine[r]
        return sum(1 << i for i, f in enumerate(children[cc]) if f in allowed)
    return joint_rows, chans_c, bits, children


def merge_channels_transposed(rows_b, chans_b, cmap, n_coarse):
    """Transposed lists with the fine channels (the rows of this side) merged into coarse ones: rows = coarse channels of
    one coarse block, channels = every output row that reads one of their children."""
    children = {}
    for f, cc in enumerate(cmap):
        children.setdefault(int(cc), []).append(f)
    reads = {}                       # fine channel -> set of output rows reading it
    for rows, chans in zip(rows_b, chans_b):
        fs = frozenset(chans)
        for f in rows:
            reads[f] = fs
    # blocks: coarse channels whose children sit in the same fine blocks are grouped like their first child
    blk_of_fine = {}
    for bi, rows in enumerate(rows_b):
        for f in rows:
            blk_of_fine[f] = bi
    blocks = {}
    for cc in range(n_coarse):
        if cc in children:
            key = tuple(sorted(set(blk_of_fine[f] for f in children[cc])))
            blocks.setdefault(key, []).append(cc)
    rows_c, chans_c = [], []
    for key in sorted(blocks):
        rows_c.append(blocks[key])
        chans_c.append(sorted(set().union(*[reads[f] for cc in blocks[key] for f in children[cc]])))

    def bits(cc, r):
        return sum(1 << i for i, f in enumerate(children[cc]) if r in reads[f])
    return rows_c, chans_c, bits, children


def merge_rows(joint_rows, joint_chans, rmap):
    """Forward lists with the output rows replaced by the pooled rows they are averaged into (rmap[fine row] = pooled
    row, SkeletonPool folded into the weights).  Blocks: pooled rows whose children sit in the same fine blocks.
    Returns (rows_p, chans_p, pair_bits, children) -- pair_bits(pooled row, chan): bit i set when child i of the pooled
    row reads that channel."""
    children = {}
    for r, pr in enumerate(rmap):
        children.setdefault(int(pr), []).append(r)
    blk_of_row, reads = {}, {}
    for bi, (rows, chans) in enumerate(zip(joint_rows, joint_chans)):
        fs = frozenset(chans)
        for r in rows:
            blk_of_row[r] = bi
            reads[r] = fs
    blocks = {}
    for pr in sorted(children):
        key = tuple(sorted(set(blk_of_row[r] for r in children[pr])))
        blocks.setdefault(key, []).append(pr)
    rows_p, chans_p = [], []
    for key in sorted(blocks):
        rows_p.append(blocks[key])
        chans_p.append(sorted(set().union(*[reads[r] for pr in blocks[key] for r in children[pr]])))

    def bits(pr, c):
        return sum(1 << i for i, r in enumerate(children[pr]) if c in reads[r])
    return rows_p, chans_p, bits, children


def merge_rows_transposed(rows_b, chans_b, rmap):
    """Transposed lists (rows = input channels, channels = output rows reading them) with the output rows merged into
    pooled rows: pair_bits(input channel, pooled row) = which children of the pooled row read the channel."""
    children = {}
    for r, pr in enumerate(rmap):
        children.setdefault(int(pr), []).append(r)
    readers = {}
    for rows, chans in zip(rows_b, chans_b):
        fs = frozenset(chans)
        for f in rows:
            readers[f] = fs
    chans_p = [sorted(set(int(rmap[r]) for r in chans)) for chans in chans_b]

    def bits(f, pr):
        return sum(1 << i for i, r in enumerate(children[pr]) if r in readers[f])
    return rows_b, chans_p, bits, children
