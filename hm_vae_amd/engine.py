"""The fused TwoHierSAVAE step: a fixed sequence of HIP kernel launches over preallocated SoA buffers.

This is the hot path named by BASELINE.json: encoder -> latent heads -> reparametrise/KL -> decoder ->
rot-6D lift -> FK -> losses, and the hand-ordered backward of all of it, i.e. what the reference does in
`TwoHierSAVAEModel.forward` (seq_two_hier_sa_vae.py:335-417) including the `l_total.backward()` at :414.
There is no autograd graph: every backward kernel is launched explicitly, gradients land in views of one flat
fp32 buffer (the DDP bucket), and the whole sequence is capturable in a hipGraph (torch.cuda.CUDAGraph).

Buffers are "SoA": [rows][Bp], row = channel*T + t, batch innermost, Bp = batch rounded up to 64.
"""
import contextlib
import os as _os0

import numpy as np
import torch

from . import _lib, ops, topology

# 1: all slab->gradient reductions and bias sums as two launches after the backward chain (fewer launches, but they
# then sit on the critical path; measured slower than overlapping them on the side streams)
MULTI_REDUCE = _os0.environ.get("HMVAE_MULTI_REDUCE", "0") == "1"
# 1: targets (transposes + ground-truth FK), noise and the non-encoder weight layouts run on a side stream beside the encoder
EARLY_SIDE = _os0.environ.get("HMVAE_EARLY_SIDE", "1") != "0"
# 1: only the first encoder layer's weight layout is packed on the main stream; layers 1.. beside it on the early stream
PACK_SPLIT = _os0.environ.get("HMVAE_PACK_SPLIT", "1") != "0"
# the decoder's / transposed layouts are packed on the early stream BEFORE the targets and the noise are staged there
LATE_PACK_FIRST = _os0.environ.get("HMVAE_LATE_PACK_FIRST", "1") != "0"
SHALLOW_SIDE = _os0.environ.get("HMVAE_SHALLOW_SIDE", "1") != "0"
# LeakyReLU backward of layer i-1 inside the data-gradient GEMM of layer i (HMVAE_FLAG_LEAKY_GRAD) instead of its own launch
# the latent bridge of a level (head -> reparametrisation + KL -> head) as one launch per direction (latent_bridge.hip)
# the reparametrisation's noise from hmvae_randn (one launch, device-side launch counter) instead of torch.randn per level: a
# graph that holds torch.randn makes torch fill two RNG scalars in front of every replay ("torch": the torch generator)
NOISE_HIP = _os0.environ.get("HMVAE_NOISE", "hip") != "torch"
FUSE_BRIDGE = _os0.environ.get("HMVAE_FUSE_BRIDGE", "1") != "0"
FUSE_LEAKY_BWD = _os0.environ.get("HMVAE_FUSE_LEAKY_BWD", "1") != "0"
# weight-gradient launches on wide tiles (ops wg_mode 4) are issued when their segment's data-gradient chain is through
# (measured 1.34 -> 1.47 ms at B = 256: the chain's launches leave CUs idle that the weight gradients fill; off by default)
DEFER_WIDE = _os0.environ.get("HMVAE_DEFER_WIDE", "0") == "1"
# Weight gradients whose launch waits for the layer's OWN data-gradient launch (instead of running beside it): layer tags
# "dec0,dec1,enc3".  The position-pair launch (ops.MaskedGemm.wgrad_pq) is as dense as the data-gradient gather-GEMMs, so the two
# gain nothing from running side by side, while the launches behind the first decoder layer (latent heads, bridge: 16-48
# workgroups) leave the chip empty.
# Default: the operator's tuned choice (ops.MaskedGemm.wg_delay, tune file); HMVAE_WG_DELAY names the layers instead ("none": no layer).
# ("dec1:2": behind the NEXT layer's data-gradient launch.)
WG_DELAY = ({t.split(":")[0]: int(t.split(":")[1]) if ":" in t else 1 for t in _os0.environ["HMVAE_WG_DELAY"].split(",") if t}
            if "HMVAE_WG_DELAY" in _os0.environ else None)


class VaePlan:
    """Static structure for one config (mirrors seq_two_hier_sa_vae.py:53-258)."""

    def __init__(self, hp, parents=topology.SMPL24_PARENTS):
        L, T = int(hp["num_layers"]), int(hp["train_seq_len"])
        self.hp, self.L, self.T = hp, L, T
        self.K = int(hp["kernel_size"])
        self.pad = (self.K - 1) // 2
        self.pad_mode = {"zeros": "zeros", "constant": "zeros", "reflection": "reflect", "reflect": "reflect"}[hp["padding_mode"]]
        self.strides, self.tsteps = topology.stride_schedule(T, L)
        self.ch_base = [int(hp.get("input_dim", 6)) * (2 ** i) for i in range(L + 1)]
        self.topo = [topology.edges_from_parents(parents)]
        self.neigh, self.pooling, self.edge_num = [], [], [len(self.topo[0])]
        self.enc_cin, self.enc_cout = [], []
        for i in range(L):
            self.neigh.append(topology.find_neighbor(self.topo[i], hp["skeleton_dist"]))
            self.enc_cin.append(self.ch_base[i] * self.edge_num[i])
            self.enc_cout.append(self.ch_base[i + 1] * self.edge_num[i])
            _, pl, ne = topology.derive_pooling(self.topo[i], last_pool=(i == L - 1))
            self.pooling.append(pl)
            self.topo.append(ne)
            self.edge_num.append(len(pl))
        self.channel_list = [self.enc_cin[0]] + self.enc_cout
        self.dec_ts = list(reversed(self.tsteps))
        self.dec_cin, self.dec_cout, self.dec_bias, self.dec_up = [], [], [], []
        for i in range(L):
            cin = self.channel_list[L - i] * (2 if i == L - 1 else 1)
            self.dec_cin.append(cin)
            self.dec_cout.append(cin // 4 if i == L - 1 else cin // 2)
            self.dec_bias.append(i == 0 or i == L - 1)
            if T == 8:
                up = (i != L - 1 and i != 0)
            elif T == 16:
                up = (i != L - 1)
            else:
                up = True
            self.dec_up.append(up)
        self.lat_dim = [int(hp["shallow_latent_d"])] + [int(hp["latent_d"])] * (L - 1)
        self.feat = [self.ch_base[i + 1] * self.tsteps[i + 1] for i in range(L)]       # Linear in-features, level i
        self.dec_feat = [self.ch_base[L - i] * self.dec_ts[i] for i in range(L)]        # Linear out-features, dec idx i


class _StepBase:
    """What the fixed kernel sequences of both models share: cached index tables, views into the flat gradient bucket,
    and the side streams the weight-gradient GEMMs run on."""

    def _init_step_base(self, device):
        import os as _os
        self.device = device
        self._bias_rows = {}
        self._flat = None
        self._side = None
        self.overlap = _os.environ.get("HMVAE_OVERLAP", "1") != "0"
        # One side stream per weight-gradient chain of a step (12-13 forks): measured A/B (shared autotune cache, 100
        # steps): 2 streams 134.4 k seq/s, 4 / 8 / 12 / 13 / 16 streams 136-138 k, but 3 / 5 / 6 / 7 only 122-134 k -- with
        # fewer streams than chains the round-robin makes late layers queue behind unrelated early chains.
        self.n_side = int(_os.environ.get("HMVAE_SIDE_STREAMS", "16"))
        self._turn = 0
        self._forked = set()              # side streams forked since the last join
        # HMVAE_JOIN_ALL=1: round-1 behaviour, kept for the A/B of tools/repro -- the join also records and waits events on
        # side streams that were never forked in this step (inside a stream capture that is an eager record on a stream
        # outside the capture, waited on by the capturing stream)
        self._join_all = _os.environ.get("HMVAE_JOIN_ALL", "0") == "1"
        self._pending_red, self._pending_rs = [], []
        self._deferred = []
        self._events = {}
        self._skip_wgrad = False          # test-time optimisation of the latents alone: data gradients only

    def _scales(self, key, scale):
        k = ("scale", key)
        if k not in self._bias_rows:
            self._bias_rows[k] = torch.as_tensor(np.asarray(scale, dtype=np.float32)).to(self.device)
        return self._bias_rows[k]

    def _rows(self, key, base, count):
        k = (key, count)
        if k not in self._bias_rows:
            self._bias_rows[k] = torch.as_tensor(np.asarray(base, dtype=np.int32)).to(self.device)
        return self._bias_rows[k]

    def grad_view(self, name):
        order, offs, *_ = self.grad_layout()
        t = self.params[name]
        return self.flat_grad[offs[name]:offs[name] + t.numel()].view(t.shape)

    # ---- weight / bias gradients run on a side stream ----
    # They only READ dY and the forward activations and write gradient slices nobody else touches, so they overlap
    # with the backward-data chain of the following layers (two kernels that each leave SIMDs idle in their tails
    # fill the chip together).  Fork = event on the main stream once dY is final; join = end of _backward.
    # Every cross-stream dependency goes through an event this object owns for its lifetime (no event is created or
    # destroyed inside a step or a stream capture: ~40 event objects per step less for the host to make and drop).
    def _event(self, key):
        ev = self._events.get(key)
        if ev is None:
            ev = self._events[key] = torch.cuda.Event()
        return ev

    def _stream_wait(self, waiter, waited, key):
        """`waiter` waits for everything queued on `waited` so far."""
        ev = self._event(key)
        ev.record(waited)
        waiter.wait_event(ev)

    def _fork(self):
        if not self.overlap:
            return None
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(self.n_side)]
        self._turn = (self._turn + 1) % self.n_side
        side = self._side[self._turn]
        self._stream_wait(side, main, ("fork", self._turn))
        self._forked.add(self._turn)
        return torch.cuda.stream(side)

    def _flush_delayed(self, force=False):
        """Queue the weight gradients that waited for the data-gradient launch just issued (WG_DELAY)."""
        lst, self._delayed = getattr(self, "_delayed", []), []
        self._flushing = True
        try:
            for args, kw, n in lst:
                if n > 1 and not force:
                    self._delayed.append((args, kw, n - 1))
                else:
                    self._wgrad(*args, **kw)
        finally:
            self._flushing = False

    def _flush_deferred(self):
        self._flush_delayed(force=True)
        lst, self._deferred = self._deferred, []
        self._flushing = True
        try:
            for args in lst:
                self._wgrad(*args)
        finally:
            self._flushing = False

    def _join(self):
        """The main stream waits for every side stream forked since the last join (and only those)."""
        self._flush_deferred()
        if self.overlap and self._side is not None:
            main = torch.cuda.current_stream()
            for i in (range(self.n_side) if self._join_all else sorted(self._forked)):
                self._stream_wait(main, self._side[i], ("join", i))
        self._forked.clear()

    def _wgrad(self, op, dy, x, name, bp, w, bias=None, before=None, src=None, tag=None):
        """bias: optional (src, grad_name, rows_key, base, n_out, rows_per_out, row_step) for the row-sum kernel.
        before: launches that produce `x`, issued on the same side stream first.
        src: the tensor the layer's gather sides read (in front of the folded Upsample / Unpool): where the operator's tuned
        choice is the position-pair launch (ops.MaskedGemm.use_pq) the gradient comes from it and `x` is never built."""
        if self._skip_wgrad:
            return
        if (tag is not None and not MULTI_REDUCE and not getattr(self, "_flushing", False)
                and ((tag in WG_DELAY) if WG_DELAY is not None else (op.wg_decided(bp) and op.wg_delay(bp) > 0))):
            if not hasattr(self, "_delayed"):
                self._delayed = []
            self._delayed.append(((op, dy, x, name, bp, w), dict(bias=bias, before=before, src=src),
                                  WG_DELAY.get(tag, 1) if WG_DELAY is not None else op.wg_delay(bp)))
            return
        if src is not None and not MULTI_REDUCE:
            if not op.wg_decided(bp):           # first eager step: time the candidates (needs x)
                if before is not None:
                    before()
                op.autotune_split(dy, x, self.grad_view(name), bp, w["slabs"][id(op)], src=src)
            if op.use_pq(bp):
                x, before = None, None
            else:
                src = None
        if DEFER_WIDE and not MULTI_REDUCE and not getattr(self, "_flushing", False) and op.wg_mode(bp) == 4:
            # wide-tile launches (one 8-wave workgroup per CU holding 130 KB of LDS for the whole tile) starve the
            # data-gradient chain when they run beside it: they are queued when the segment's chain is through (_flush_deferred)
            self._deferred.append((op, dy, x, name, bp, w, bias, before))
            return
        slab = w["slabs"][id(op)]
        if not MULTI_REDUCE:
            # default: GEMM + reduction + bias sums all on the side stream, overlapped with the data-gradient chain
            # (Issuing the side chain AFTER the main stream's next launch -- so that under capture the data-gradient chain
            # keeps its hardware queue at every fork instead of hopping -- was measured 20 % SLOWER, 1.90 vs 1.58 ms: the
            # graph's queues then serialise the weight-gradient chains among themselves.  Issue order is kept as is.)
            ctx = self._fork() or contextlib.nullcontext()
            with ctx:
                if before is not None:
                    before()
                op.backward_weight(dy, x, self.grad_view(name), bp, slab, src=src)
                if bias is not None:
                    src, gname, key, base, n_out, rows_per, step = bias[:7]
                    if len(bias) > 7:       # factor per output: rows averaged by a folded SkeletonPool
                        ops.row_sums_scaled(src, self.grad_view(gname), self._rows(key, base, len(base)),
                                            self._scales(key, bias[7]), n_out, rows_per, step, bp)
                    else:
                        ops.row_sums(src, self.grad_view(gname), self._rows(key, base, len(base)), n_out, rows_per, step, bp)
            return
        self._pending_red.append((op, slab, self.grad_view(name)))
        if bias is not None:
            if len(bias) > 7:
                raise _lib.HmvaeError("HMVAE_MULTI_REDUCE=1 does not support folded pooling (set HMVAE_FOLD=0)")
            src, gname, key, base, n_out, rows_per, step = bias
            self._pending_rs.append((src, self.grad_view(gname), self._rows(key, base, len(base)), n_out, rows_per, step))
        ctx = self._fork() or contextlib.nullcontext()
        with ctx:
            if before is not None:
                before()
            op.wgrad_gemm(dy, x, bp, slab)


class VaeEngine(_StepBase):
    """Owns device tables + workspaces; borrows the parameter tensors from the nn.Modules (dict name -> Parameter)."""

    def __init__(self, hp, params, offsets, device, parents=topology.SMPL24_PARENTS):
        _lib.load()
        self.hp, self.device = hp, device
        self.parents = list(parents)
        self.p = p = VaePlan(hp, parents)
        self.params = params                       # name -> tensor (dense layout, reference names)
        self.offsets = torch.as_tensor(np.asarray(offsets), dtype=torch.float32).contiguous().to(device)
        L = p.L
        if L < 2:
            raise ValueError("num_layers >= 2 expected")
        self.nj = int(hp["n_joints"])
        if self.nj != 24 or int(hp["output_dim"]) != 6:
            raise NotImplementedError("the fused decode tail is specialised for 24 joints x 6D")
        # ---- encoder ops ----
        self.enc_conv, self.enc_pool, self.enc_lin = [], [], []
        for i in range(L):
            pool = ops.PoolOp(device, p.pooling[i], p.edge_num[i], p.enc_cout[i] // p.edge_num[i])
            # SkeletonPool + LeakyReLU are folded into the convolution's forward / backward-data GEMMs (ops.conv_gemm):
            # forward writes the pooled activated tensor A directly; the gradient w.r.t. the conv's own rows (dY) is only
            # built for the weight gradient, on its side stream
            fuse = ops.FOLD and p.K <= 16 and pool.covered and all(len(g) <= 2 for g in p.pooling[i])
            self.enc_conv.append(ops.conv_gemm(device, p.neigh[i], p.enc_cin[i], p.enc_cout[i], p.K, p.strides[i], p.pad,
                                               p.pad_mode, p.tsteps[i], need_dx=(i > 0),
                                               pool_groups=p.pooling[i] if fuse else None))
            self.enc_pool.append(pool)
            k = p.edge_num[i + 1]
            self.enc_lin.append(ops.linear_gemm(device, 2 * p.lat_dim[i], p.feat[i], k, p.feat[i], 2 * p.lat_dim[i]))
        # ---- decoder ops ----
        self.dec_conv, self.dec_unpool, self.dec_lin = [], [], {}
        self.cD = p.dec_cout[L - 2] // p.edge_num[1]          # running channels per edge before the concat
        # encoder layers >= enc_cut: the coarse half (gradient bucket / backward cut of the DDP step)
        self.enc_cut = min(max(int(_os0.environ.get("HMVAE_ENC_CUT", L // 2)), 1), L - 1) if L > 1 else 0
        self.cH = p.ch_base[1]                                # skip (hier_feats[L-1]) channels per edge
        assert p.dec_cin[L - 1] // p.edge_num[0] == self.cD + self.cH
        for i in range(L):
            t_in = p.dec_ts[i]
            t_conv = 2 * t_in if p.dec_up[i] else t_in
            assert t_conv == p.dec_ts[i + 1], (i, t_conv, p.dec_ts)
            lvl = L - 1 - i
            self.dec_unpool.append(ops.UnpoolOp(device, p.pooling[lvl], p.dec_cin[i] // p.edge_num[lvl], p.dec_up[i]))
            cmap = None
            if i == L - 2:      # output lands in the per-edge concat buffer (seq_two_hier_sa_vae.py:280-288)
                cmap = [(c // self.cD) * (self.cD + self.cH) + (c % self.cD) for c in range(p.dec_cout[i])]
            # Upsample + Unpool are folded into the convolution's forward and backward-data GEMMs (ops.conv_gemm): they
            # read / write the coarse tensor directly; the unpooled, upsampled tensor U is only built for the weight
            # gradient, on its side stream
            usrc = np.zeros(p.edge_num[lvl], dtype=np.int64)
            for ci, grp in enumerate(p.pooling[lvl]):
                for j in grp:
                    usrc[j] = ci
            fuse = ops.FOLD and p.K <= 16 and bool(p.dec_up[i])
            self.dec_conv.append(ops.conv_gemm(device, p.neigh[lvl], p.dec_cin[i], p.dec_cout[i], p.K, 1, p.pad, p.pad_mode,
                                               t_conv, need_dx=True, out_chan_map=cmap, upsample=fuse,
                                               unpool_src=usrc if fuse else None))
        kd = p.edge_num[L]
        self.dec_lin[0] = ops.linear_gemm(device, p.dec_feat[0], p.lat_dim[L - 1], kd, p.lat_dim[L - 1], p.dec_feat[0])
        t_cat = p.dec_ts[L - 1]
        assert p.dec_feat[L - 1] == self.cH * t_cat
        self.cat_edge_rows = (self.cD + self.cH) * t_cat
        self.dec_lin[L - 1] = ops.linear_gemm(device, p.dec_feat[L - 1], p.lat_dim[0], p.edge_num[1], p.lat_dim[0],
                                              self.cat_edge_rows, y_offset=self.cD * t_cat)
        # the dead middle heads (seq_two_hier_sa_vae.py:264-272 computes them, nothing consumes them) exist only for
        # Decoder.forward's API; they are built lazily
        self._init_step_base(device)
        self._ws = {}
        self._shallow = None
        self._noise_state = None
        self._cat_act = None
        self._dp_done = set()
        self._multipack = None
        self._multipack_late = None
        self._multipack_dec = None
        self._multipack_mid = None
        self._late_packs = []
        self._mid_packs = []
        self._mid_event = None
        self._early = None
        self._multired = None
        self.flops_per_seq = self._count_flops()

    # ------------------------------------------------------------------------------------------
    def _count_flops(self):
        """Mask-effective conv FLOPs per sequence, forward (2*MAC), as in SURVEY.md §8(d)."""
        tot = 0
        for op in self.enc_conv + self.dec_conv:
            tot += 2 * op.useful_macs_per_col
        return tot

    def noise_state_dict(self):
        """Seed and launch counter of the in-kernel noise generator (hmvae_randn), or None before its first use: saved
        with the optimiser checkpoint so that a resumed run continues the noise stream instead of replaying it from 0."""
        if self._noise_state is None:
            return None
        return {"seed": int(self._noise_seed), "state": [int(v) for v in self._noise_state.tolist()]}

    def load_noise_state(self, d):
        if not d:
            return
        self._noise_seed = int(d["seed"]) & 0xFFFFFFFFFFFFFFFF
        st = torch.tensor([int(v) for v in d["state"]], dtype=torch.int64, device=self.device)
        if self._noise_state is None:
            self._noise_state = st
        else:
            self._noise_state.copy_(st)          # (a captured graph holds this tensor's address)

    def workspace(self, B):
        if B in self._ws:
            return self._ws[B]
        if torch.device(self.device).type != "cuda":
            raise _lib.HmvaeError("the engine computes on a HIP device only (tables may be planned on CPU for inspection)")
        p, dev = self.p, self.device
        bp = ops.pad_batch(B)
        L = p.L

        def buf(rows):
            return torch.zeros(int(rows), bp, dtype=torch.float32, device=dev)

        w = {"bp": bp, "B": B}
        T0 = p.T
        w["X0"], w["GR"], w["GP"] = buf(144 * T0), buf(216 * T0), buf(72 * T0)
        w["pos_aos"] = torch.zeros(B * T0, 24, 3, device=dev)
        for i in range(L):
            t = p.tsteps[i + 1]
            w["Y%d" % i], w["dY%d" % i] = buf(p.enc_cout[i] * t), buf(p.enc_cout[i] * t)
            pc = p.ch_base[i + 1] * p.edge_num[i + 1]
            w["A%d" % i], w["dA%d" % i] = buf(pc * t), buf(pc * t)
            w["dP%d" % i] = buf(pc * t)                   # gradient w.r.t. the pooled pre-activation (fused pool)
            k, d = p.edge_num[i + 1], p.lat_dim[i]
            w["ZD%d" % i], w["dZD%d" % i] = buf(k * 2 * d), buf(k * 2 * d)
            w["Z%d" % i], w["dZ%d" % i] = buf(k * d), buf(k * d)
            w["KLP%d" % i] = torch.zeros(ops.kl_partial_blocks(k, bp, d), device=dev)
            w["KLPB%d" % i] = torch.zeros(k * (bp // 16), device=dev)          # ... of the one-launch latent bridge
        w["H0"], w["dH0"] = buf(p.edge_num[L] * p.dec_feat[0]), buf(p.edge_num[L] * p.dec_feat[0])
        for i in range(L):
            t_conv = p.dec_ts[i + 1]
            w["U%d" % i], w["dU%d" % i] = buf(p.dec_cin[i] * t_conv), buf(p.dec_cin[i] * t_conv)
            if i == L - 2:
                rows = p.edge_num[1] * self.cat_edge_rows
            else:
                rows = p.dec_cout[i] * t_conv
            w["D%d" % i], w["dD%d" % i] = buf(rows), buf(rows)
        w["MSEP"] = torch.zeros(ops.decode_tail_loss_blocks(T0, B, bp), 3, device=dev)
        w["losses"] = torch.zeros(8, device=dev)
        # one slab per weight tensor: the weight-gradient GEMMs run on side streams and are all reduced together at the
        # end of the backward pass (one launch)
        all_ops = self.enc_conv + self.dec_conv + self.enc_lin + list(self.dec_lin.values())
        tot = sum(op.slab_floats(bp) for op in all_ops)
        big = torch.empty(max(tot, 1), device=dev)
        o = 0
        w["slabs"] = {}
        for op in all_ops:
            n = op.slab_floats(bp)
            w["slabs"][id(op)] = big[o:o + n]
            o += n
        w["slab"] = big
        self._ws[B] = w
        return w

    # ------------------------------------------------------------------------------------------
    # parameters / gradients
    # ------------------------------------------------------------------------------------------
    def names(self):
        p, L = self.p, self.p.L
        enc_w = ["enc.layers.%d.0.weight" % i for i in range(L)]
        enc_b = ["enc.layers.%d.0.bias" % i for i in range(L)]
        dconv = ["dec.layers.%d.%d" % (i, 2 if p.dec_up[i] else 1) for i in range(L)]
        return enc_w, enc_b, dconv

    def grad_layout(self):
        """Flat gradient bucket: [always active | active from iteration_interval on | never].  Returns
        (ordered names, offsets, n_phase1, n_phase2) in floats."""
        if self._flat is not None:
            return self._flat
        p, L = self.p, self.p.L
        enc_w, enc_b, dconv = self.names()
        # decoder first: its gradients are complete half-way through the backward pass, so under DDP their slice of the
        # bucket is all-reduced while the encoder's backward runs (graph_step.py)
        always = []
        for i in range(L):
            always.append(dconv[i] + ".weight")
            if p.dec_bias[i]:
                always.append(dconv[i] + ".bias")
        for i in (0, L - 1):
            always += ["dec.latent_dec_layers.%d.%s" % (i, s) for s in ("weight", "bias")]
        n_dec_names = len(always)
        # then the encoder's coarse half (layers L-1 .. L/2 and the deep latent head): most of the encoder's bytes, final
        # after the first, cheap part of its backward -- a second asynchronous all-reduce beside the fine layers' backward;
        # the fine layers (large T, small weights) come last and are the only exposed collective
        for i in range(L - 1, self.enc_cut - 1, -1):
            always += [enc_w[i], enc_b[i]]
        always += ["enc.latent_enc_layers.%d.%s" % (L - 1, s) for s in ("weight", "bias")]
        n_hi_names = len(always)
        for i in range(self.enc_cut - 1, -1, -1):
            always += [enc_w[i], enc_b[i]]
        phase2 = ["enc.latent_enc_layers.0.%s" % s for s in ("weight", "bias")]
        never = []
        for i in range(1, L - 1):
            never += ["enc.latent_enc_layers.%d.%s" % (i, s) for s in ("weight", "bias")]
            never += ["dec.latent_dec_layers.%d.%s" % (i, s) for s in ("weight", "bias")]
        order = always + phase2 + never
        offs, o = {}, 0
        for n in order:
            offs[n] = o
            o += self.params[n].numel()
            o = (o + 3) // 4 * 4
        n1 = offs[phase2[0]]
        n2 = offs[never[0]] if never else o
        self.n_dec_floats = offs[always[n_dec_names]]          # [0, n_dec_floats): every decoder gradient
        self.n_hi_floats = offs[always[n_hi_names]] if n_hi_names < len(always) else n1   # [n_dec, n_hi): coarse encoder half
        self.flat_grad = torch.zeros(o, dtype=torch.float32, device=self.device)
        self._flat = (order, offs, n1, n2, set(always), set(phase2), set(never))
        return self._flat

    def assign_grads(self, phase2):
        """Point .grad of the live parameters at their slice of the flat bucket; the rest get None
        (torch.optim.Adam skips None-grad params — SURVEY.md §3.1 structural fact 1)."""
        order, offs, n1, n2, always, ph2, never = self.grad_layout()
        for n in order:
            t = self.params[n]
            if n in always or (phase2 and n in ph2):
                t.grad = self.grad_view(n)
            else:
                t.grad = None

    def active_grad_floats(self, phase2):
        _, _, n1, n2, *_ = self.grad_layout()
        return n2 if phase2 else n1

    def _pack_list(self):
        """(op, dense weight) in the order the step consumes them."""
        P = self.params
        enc_w, _, dconv = self.names()
        L = self.p.L
        lst = [(self.enc_conv[i], P[enc_w[i]]) for i in range(L)]
        lst += [(self.enc_lin[i], P["enc.latent_enc_layers.%d.weight" % i]) for i in (0, L - 1)]
        lst += [(self.dec_lin[i], P["dec.latent_dec_layers.%d.weight" % i]) for i in (0, L - 1)]
        lst += [(self.dec_conv[i], P[dconv[i] + ".weight"]) for i in range(L)]
        return lst

    def pack_weights(self, side=True, defer_late=False):
        """Regather the dense weights into MFMA-fragment order (applies the neighbour mask, skeleton.py:96): every layout
        of the model in at most three launches (`side` is kept for callers of the older signature and ignored).  A variant
        with one launch and one event per layer on its own stream was measured slower (3.33 vs 3.03 ms/step) and removed."""
        lst = self._pack_list()
        if self._multipack is None:
            self._multipack = ops.MultiPack(self.device)
            self._multipack_late = ops.MultiPack(self.device)
            self._multipack_mid = ops.MultiPack(self.device)
        # two launches: what the encoder needs right away, and the rest (decoder layouts, every transposed layout),
        # which the early side stream packs while the encoder runs (train_step joins it before the decoder)
        first, late = [], []
        enc_ops = set(id(o) for o in self.enc_conv + self.enc_lin)
        for op, wt in lst:
            (first if (id(op) in enc_ops or not EARLY_SIDE) else late).append((op.fwd, wt))
            if op.bwd is not None:
                (late if EARLY_SIDE else first).append((op.bwd, wt))
        # three launches inside train_step: the first encoder layer's layout now (the pack kernel is a chain of
        # dependent table reads: ~60 us whatever it packs), the other encoder layouts on the early side stream under
        # the batch staging and the first convolution (encode() waits for them before layer 1), the rest later
        mid = []
        if defer_late and EARLY_SIDE and self.overlap and PACK_SPLIT:
            keep = id(self.enc_conv[0].fwd)
            mid = [(sd, wt) for sd, wt in first if id(sd) != keep]
            first = [(sd, wt) for sd, wt in first if id(sd) == keep]
        self._multipack.run(first)
        self._late_packs = late
        self._mid_packs = mid
        if not defer_late and late:          # callers outside train_step: everything now, on this stream
            self._multipack_late.run(late)
            self._late_packs = []

    def pack_decoder(self):
        """The decoder's layouts only (forward and transposed): what the test-time optimisation loop re-reads."""
        P, L = self.params, self.p.L
        _, _, dconv = self.names()
        if self._multipack_dec is None:
            self._multipack_dec = ops.MultiPack(self.device)
        lst = []
        for op, wt in [(self.dec_lin[i], P["dec.latent_dec_layers.%d.weight" % i]) for i in (0, L - 1)] + \
                [(self.dec_conv[i], P[dconv[i] + ".weight"]) for i in range(L)]:
            lst.append((op.fwd, wt))
            if op.bwd is not None:
                lst.append((op.bwd, wt))
        self._multipack_dec.run(lst)

    # ------------------------------------------------------------------------------------------
    # forward pieces
    # ------------------------------------------------------------------------------------------
    def _early_stream(self):
        main = torch.cuda.current_stream()
        if self._early is None:
            self._early = torch.cuda.Stream(device=self.device)
        self._stream_wait(self._early, main, "early_fork")
        return self._early

    def stage_batch(self, w, r6, rmat):
        """AoS (B,T,144)/(B,T,216) device tensors -> SoA inputs/targets; target joint positions by FK
        (seq_two_hier_sa_vae.py:343)."""
        B, T, bp = w["B"], self.p.T, w["bp"]
        ops.aos_to_soa(r6, w["X0"], B, T, 144, bp)
        self.stage_targets(w, rmat)

    def stage_targets(self, w, rmat):
        B, T, bp = w["B"], self.p.T, w["bp"]
        ops.aos_to_soa(rmat, w["GR"], B, T, 216, bp)
        ops.fk_fwd(rmat, 9, self.offsets, 0, self.parents, B * T, None, w["pos_aos"])
        ops.aos_to_soa(w["pos_aos"], w["GP"], B, T, 72, bp)

    def _bridge_geom(self, i):
        """(enc head, dec head, dec head's name index, k, d, n_in, n_out, y rows per edge, y offset, y buffer key) of level i's
        latent bridge, or None when the one-launch kernels do not take it (hmvae_latent_bridge_fwd: d <= 24, d % 4, rows % 64)."""
        p, L = self.p, self.p.L
        if not FUSE_BRIDGE or i not in (0, L - 1) or (i == 0 and L == 1):
            return None
        k, d, n_in = p.edge_num[i + 1], p.lat_dim[i], p.feat[i]
        n_out = p.dec_feat[0] if i == L - 1 else p.dec_feat[L - 1]
        if d > 24 or d % 4 or n_in % 64 or n_out % 64:
            return None
        if i == L - 1:
            return (0, k, d, n_in, n_out, p.dec_feat[0], 0, "H0")
        return (L - 1, k, d, n_in, n_out, self.cat_edge_rows, self.cD * p.dec_ts[L - 1], "D%d" % (L - 2))

    def _bridge_fwd(self, w, i, eps, sample):
        P, bp = self.params, w["bp"]
        di, k, d, n_in, n_out, y_es, y_off, ykey = self._bridge_geom(i)
        ops.latent_bridge_fwd(w["A%d" % i], P["enc.latent_enc_layers.%d.weight" % i], P["enc.latent_enc_layers.%d.bias" % i],
                              eps[i], P["dec.latent_dec_layers.%d.weight" % di], P["dec.latent_dec_layers.%d.bias" % di],
                              w["ZD%d" % i], w["Z%d" % i], w[ykey], w["KLPB%d" % i], k, n_in, d, n_out, n_in, y_es, y_off,
                              w["B"], bp, sample)

    def encode(self, w, heads, after_first=None):
        """after_first: called once layer 0 has been launched (the training step starts the shallow latent branch there)."""
        p, P, bp = self.p, self.params, w["bp"]
        enc_w, enc_b, _ = self.names()
        x = w["X0"]
        for i in range(p.L):
            if i == 1 and after_first is not None:
                after_first()
            if i == 1 and self._mid_event is not None:           # layouts of layers 1.. were packed beside layer 0
                torch.cuda.current_stream().wait_event(self._mid_event)
                self._mid_event = None
            if self.enc_conv[i].fused_pool:
                self.enc_conv[i].forward(x, w["A%d" % i], bp, bias=P[enc_b[i]], flags=_lib.FLAG_LEAKY)
            else:
                self.enc_conv[i].forward(x, w["Y%d" % i], bp, bias=P[enc_b[i]])
                self.enc_pool[i].forward(w["Y%d" % i], w["A%d" % i], p.tsteps[i + 1], bp)
            x = w["A%d" % i]
        for i in heads:
            self.enc_lin[i].forward(w["A%d" % i], w["ZD%d" % i], bp, bias=P["enc.latent_enc_layers.%d.bias" % i])

    def decode_feats(self, w, shallow_done=None, bridged=()):
        """Decoder.forward (seq_two_hier_sa_vae.py:260-294) from Z{L-1} and Z0; result in D{L-1}.
        shallow_done: an event after which the shallow head's part of the concat buffer is in place (the training step
        computes it on a side stream, see train_step); the last layer waits for it.
        bridged: levels whose decoder head has already run (the one-launch latent bridge of the training step)."""
        p, P, bp, L = self.p, self.params, w["bp"], self.p.L
        _, _, dconv = self.names()
        if (L - 1) not in bridged:
            self.dec_lin[0].forward(w["Z%d" % (L - 1)], w["H0"], bp, bias=P["dec.latent_dec_layers.0.bias"])
        if shallow_done is None and 0 not in bridged:
            self.dec_lin[L - 1].forward(w["Z0"], w["D%d" % (L - 2)], bp, bias=P["dec.latent_dec_layers.%d.bias" % (L - 1)])
        src = w["H0"]
        for i in range(L):
            if i == L - 1 and shallow_done is not None:
                torch.cuda.current_stream().wait_event(shallow_done)
            bias = P[dconv[i] + ".bias"] if p.dec_bias[i] else None
            if self.dec_conv[i].fused_front:
                x = src
            else:
                self.dec_unpool[i].forward(src, w["U%d" % i], p.dec_ts[i], bp)
                x = w["U%d" % i]
            self.dec_conv[i].forward(x, w["D%d" % i], bp, bias=bias, flags=_lib.FLAG_LEAKY if i < L - 1 else 0)
            src = w["D%d" % i]

    # ------------------------------------------------------------------------------------------
    # the training / validation step
    # ------------------------------------------------------------------------------------------
    def train_step(self, r6, rmat, iterations, eps_list=None, backward=True, repack=True, stage="all", slice_hook=None):
        """r6 (B,T,144), rmat (B,T,216): fp32 device tensors.  Returns the `losses` device tensor:
        [total, kl, rec_6d, rec_rot, rec_pose, kl_level0, kl_deep, 0].
        slice_hook (stage "all" only): called as slice_hook("dec") once every launch that writes the decoder's slice of
        the gradient bucket has been QUEUED (main + side streams, not joined) and slice_hook("enc_hi") likewise for the
        coarse encoder half: the in-graph DDP step starts each slice's all-reduce there, beside the rest of the backward."""
        hp, p, P = self.hp, self.p, self.params
        L = p.L
        B, T = r6.shape[0], r6.shape[1]
        if T != p.T:
            raise ValueError("sequence length %d != train_seq_len %d" % (T, p.T))
        w = self.workspace(B)
        bp = w["bp"]
        phase2 = iterations >= hp["iteration_interval"]
        sample = hp["kl_w"] != 0
        # levels whose latent bridge is one launch per direction; the noise must exist for the sampling variant
        bridge = tuple(i for i in sorted(set((0, L - 1))) if self._bridge_geom(i) is not None)
        self._bridge_levels = bridge
        self._late_packs = []
        self._mid_packs = []
        self._mid_event = None
        # Off the critical path, on the early side stream while the encoder runs: the layouts of encoder layers 1.. (forked
        # BEFORE the first pack, which only covers layer 0), the targets (transposes + FK of the ground truth,
        # seq_two_hier_sa_vae.py:343), the noise of every level -- drawn in level order 0..L-1 so the generator is
        # consumed like the reference (:359-391) -- and the weight layouts the encoder does not read.
        early = self._early_stream() if (EARLY_SIDE and self.overlap) else None
        x0_event = None
        if early is not None and repack and PACK_SPLIT:
            # the input transpose runs beside the first pack (both only depend on the previous step)
            with torch.cuda.stream(early):
                ops.aos_to_soa(r6, w["X0"], B, T, 144, bp)
                x0_event = self._event("x0")
                x0_event.record(early)
        if repack:
            self.pack_weights(defer_late=True)
            if self._mid_packs:
                with (torch.cuda.stream(early) if early is not None else contextlib.nullcontext()):
                    self._multipack_mid.run(self._mid_packs)
                    self._mid_packs = []
                    if early is not None:
                        self._mid_event = self._event("mid")
                        self._mid_event.record(early)
        if x0_event is not None:
            torch.cuda.current_stream().wait_event(x0_event)
        else:
            ops.aos_to_soa(r6, w["X0"], B, T, 144, bp)
        eps = []
        with (torch.cuda.stream(early) if early is not None else contextlib.nullcontext()):
            # the late layouts first: they are the longest piece of this stream and the decoder waits for them
            if self._late_packs and LATE_PACK_FIRST:
                self._multipack_late.run(self._late_packs)
                self._late_packs = []
            self.stage_targets(w, rmat)
            if eps_list is None and sample and NOISE_HIP:
                # the two levels that are sampled (the middle heads feed nothing), one launch, the launch counter on the device
                if "EPS" not in w:
                    n0, nd = B * p.edge_num[1] * p.lat_dim[0], B * p.edge_num[L] * p.lat_dim[L - 1]
                    w["EPS"] = torch.empty(n0 + (nd if L > 1 else 0), device=self.device)
                    w["EPS_views"] = {0: w["EPS"][:n0].view(B * p.edge_num[1], p.lat_dim[0])}
                    if L > 1:
                        w["EPS_views"][L - 1] = w["EPS"][n0:].view(B * p.edge_num[L], p.lat_dim[L - 1])
                if self._noise_state is None:
                    self._noise_state = torch.zeros(2, dtype=torch.int64, device=self.device)
                    # (one stream per rank: the ranks of a data-parallel job are usually seeded alike)
                    from . import ddp as _ddp
                    self._noise_seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * _ddp.rank()) & 0xFFFFFFFFFFFFFFFF
                ops.randn_(w["EPS"], self._noise_seed, self._noise_state)
            for i in range(L):
                k, d = p.edge_num[i + 1], p.lat_dim[i]
                if eps_list is not None:
                    e = eps_list[i]
                elif sample and NOISE_HIP:
                    e = w["EPS_views"].get(i)
                elif sample:
                    e = torch.randn(B * k, d, device=self.device)
                else:
                    e = None
                eps.append(e)
            if self._late_packs:
                self._multipack_late.run(self._late_packs)
            eps_ready = None
            if early is not None and SHALLOW_SIDE:
                eps_ready = self._event("eps_ready")
                eps_ready.record(early)
        shallow_done = None
        if eps_ready is not None:
            # The shallow latent branch (level-0 head -> reparametrisation -> its decoder head into the last layer's concat
            # buffer) only needs the first encoder layer's output: it runs on a stream of its own beside the encoder instead
            # of as three launches between the encoder's last and the decoder's first convolution (1.510 -> 1.497 ms,
            # interleaved A/B; the same move for the branch's BACKWARD -- its two data-gradient launches beside the decoder's
            # backward instead of between decoder and encoder -- gave the gain back, 1.509 ms, and is not done).
            mid_event = self._mid_event

            def shallow():
                main = torch.cuda.current_stream()
                if self._shallow is None:
                    self._shallow = torch.cuda.Stream(device=self.device)
                sb = self._shallow
                self._stream_wait(sb, main, "shallow_fork")
                if mid_event is not None:
                    sb.wait_event(mid_event)               # the head's layout is packed with layers 1..
                sb.wait_event(eps_ready)                   # noise + the decoder heads' layouts (early stream)
                with torch.cuda.stream(sb):
                    k, d = p.edge_num[1], p.lat_dim[0]
                    if 0 in bridge:
                        self._bridge_fwd(w, 0, eps, sample)
                    else:
                        self.enc_lin[0].forward(w["A0"], w["ZD0"], bp, bias=P["enc.latent_enc_layers.0.bias"])
                        ops.reparam_kl_fwd(w["ZD0"], eps[0], w["Z0"], w["KLP0"], k, d, B, bp, sample)
                        self.dec_lin[L - 1].forward(w["Z0"], w["D%d" % (L - 2)], bp,
                                                    bias=P["dec.latent_dec_layers.%d.bias" % (L - 1)])
                    ev = self._event("shallow_done")
                    ev.record(sb)
                return ev
            holder = []
            self.encode(w, tuple(i for i in (L - 1,) if i not in bridge), after_first=lambda: holder.append(shallow()))
            shallow_done = holder[0]
        else:
            self.encode(w, tuple(i for i in (0, L - 1) if i not in bridge))
        if early is not None:
            self._stream_wait(torch.cuda.current_stream(), early, "early_join")
        for i in ((L - 1,) if shallow_done is not None else (0, L - 1)):
            if i in bridge:
                self._bridge_fwd(w, i, eps, sample)
                continue
            k, d = p.edge_num[i + 1], p.lat_dim[i]
            ops.reparam_kl_fwd(w["ZD%d" % i], eps[i], w["Z%d" % i], w["KLP%d" % i], k, d, B, bp, sample)
        self.decode_feats(w, shallow_done, bridged=bridge)
        n6, nR, nP = B * T * 144, B * T * 216, B * T * 72
        ops.decode_tail_loss(w["D%d" % (L - 1)], w["X0"], w["GR"], w["GP"], self.offsets, T, B, bp,
                             hp["rec_6d_w"] / n6, hp["rec_rot_w"] / nR, hp["rec_pose_w"] / nP, w["dD%d" % (L - 1)], w["MSEP"])
        ls = w["losses"]
        ops.vae_loss_finalize(w["MSEP"], w["KLPB0" if 0 in bridge else "KLP0"],
                              w[("KLPB%d" if (L - 1) in bridge else "KLP%d") % (L - 1)], 1.0 / n6, 1.0 / nR, 1.0 / nP,
                              1.0 / (B * p.edge_num[1]), 1.0 / (B * p.edge_num[L]), hp["rec_6d_w"], hp["rec_rot_w"],
                              hp["rec_pose_w"], hp["kl_w"], hp["shallow_kl_w"], ls)
        if backward:
            # stage "dec": stop after the decoder's backward (all its gradients final, side streams joined);
            # backward_rest() then runs the latent + encoder part.  DDP uses the cut to overlap the first all-reduce.
            self._bw_ctx = (w, phase2, eps, B, bp)
            self._slice_hook = slice_hook if stage == "all" else None
            self._backward(w, phase2, eps, B, bp, part="dec" if stage == "dec" else "all")
            self._slice_hook = None
            if stage != "dec":
                self.assign_grads(phase2)
        return ls

    def backward_rest(self, part="enc"):
        """part: "enc" (all of it), or "enc_hi" (latents + coarse layers, their gradients final afterwards) then "enc_lo"."""
        w, phase2, eps, B, bp = self._bw_ctx
        self._backward(w, phase2, eps, B, bp, part=part)
        if part != "enc_hi":
            self.assign_grads(phase2)

    def _backward(self, w, phase2, eps, B, bp, part="all"):
        hp, p, L = self.hp, self.p, self.p.L
        enc_w, enc_b, dconv = self.names()
        if part in ("all", "dec"):
            self._turn = 0                      # the same chain -> stream assignment every step
            self._pending_red, self._pending_rs = [], []
            self._backward_decoder(w, phase2, bp, dconv, bridge=getattr(self, "_bridge_levels", ()))
            if part == "dec":
                self._join()
                return
            if getattr(self, "_slice_hook", None) is not None:
                self._flush_deferred()          # every launch that writes the slice must be queued
                self._slice_hook("dec")
        self._backward_encoder(w, phase2, eps, B, bp, enc_w, enc_b, part if part in ("enc_hi", "enc_lo") else "all")

    def _concat_act_base(self):
        """Row map of the last decoder layer's input (the per-edge concat buffer, seq_two_hier_sa_vae.py:280-288) for the fused
        LeakyReLU backward: the convolution's rows map to themselves, the skip (latent head) rows to -1."""
        if self._cat_act is None:
            p, L = self.p, self.p.L
            t = p.dec_ts[L - 1]
            n_src = self.cat_edge_rows // t * p.edge_num[1]                 # channels of the concat buffer
            ch_per_edge = self.cat_edge_rows // t
            base = np.arange(n_src, dtype=np.int64) * t
            skip = (np.arange(n_src) % ch_per_edge) >= self.cD
            self._cat_act = np.where(skip, -1, base).astype(np.int64)
        return self._cat_act

    def _backward_decoder(self, w, phase2, bp, dconv, bridge=()):
        """bridge: levels whose latent heads' DATA gradients are left to the one-launch bridge of _backward_encoder_head (the
        training step passes them; callers that stop at the latents' gradients -- the refinement loop -- pass none)."""
        hp, p, L = self.hp, self.p, self.p.L
        # ---------------- decoder ----------------
        fused_prev = False          # the previous iteration's data-gradient GEMM already applied this layer's LeakyReLU backward
        for i in range(L - 1, -1, -1):
            t_conv = p.dec_ts[i + 1]
            dy = w["dD%d" % i]
            if i < L - 1 and not fused_prev:
                if i == L - 2:      # only the conv half of each edge's concat block went through LeakyReLU
                    ops.leaky_bwd(dy, w["D%d" % i], dy, self.cD * t_conv * bp, n_blocks=p.edge_num[1],
                                  block_stride=self.cat_edge_rows * bp)
                else:
                    ops.leaky_bwd(dy, w["D%d" % i], dy, dy.numel())
            conv = self.dec_conv[i]
            bias = None
            if p.dec_bias[i]:
                base = (np.arange(p.dec_cout[i]) * t_conv).astype(np.int32)
                bias = (dy, dconv[i] + ".bias", ("dconvb", i), base, p.dec_cout[i], t_conv, 1)
            dst = w["dH0"] if i == 0 else w["dD%d" % (i - 1)]
            fused_prev = False
            if conv.fused_front:
                src = w["H0"] if i == 0 else w["D%d" % (i - 1)]
                self._wgrad(conv, dy, w["U%d" % i], dconv[i] + ".weight", bp, w, bias,
                            before=lambda i=i, src=src: self.dec_unpool[i].forward(src, w["U%d" % i], p.dec_ts[i], bp),
                            src=src if i < L - 1 else None,     # (the last layer reads the per-edge concat buffer)
                            tag="dec%d" % i)
                if i >= 1 and FUSE_LEAKY_BWD:
                    # layer i-1's LeakyReLU backward rides in this launch's epilogue (its output D[i-1] is laid out like dst;
                    # in the concat buffer only the convolution's half of every edge block went through the activation)
                    fused_prev = conv.backward_data(dy, dst, bp, act=w["D%d" % (i - 1)],
                                                    act_base=self._concat_act_base() if i - 1 == L - 2 else None)
                else:
                    conv.backward_data(dy, dst, bp)
                self._flush_delayed()
            else:
                self._wgrad(conv, dy, w["U%d" % i], dconv[i] + ".weight", bp, w, bias)
                conv.backward_data(dy, w["dU%d" % i], bp)
                self.dec_unpool[i].backward(w["dU%d" % i], dst, p.dec_ts[i], bp)
        # latent decode layers
        t_cat = p.dec_ts[L - 1]
        lin = self.dec_lin[L - 1]
        dcat = w["dD%d" % (L - 2)]
        base = (np.arange(p.dec_feat[L - 1]) + self.cD * t_cat).astype(np.int32)
        self._wgrad(lin, dcat, w["Z0"], "dec.latent_dec_layers.%d.weight" % (L - 1), bp, w,
                    (dcat, "dec.latent_dec_layers.%d.bias" % (L - 1), "dlinb_last", base, p.dec_feat[L - 1], p.edge_num[1],
                     self.cat_edge_rows))
        if phase2 and 0 not in bridge:
            lin.backward_data(dcat, w["dZ0"], bp)
        lin = self.dec_lin[0]
        base = np.arange(p.dec_feat[0]).astype(np.int32)
        self._wgrad(lin, w["dH0"], w["Z%d" % (L - 1)], "dec.latent_dec_layers.0.weight", bp, w,
                    (w["dH0"], "dec.latent_dec_layers.0.bias", "dlinb0", base, p.dec_feat[0], p.edge_num[L], p.dec_feat[0]))
        if (L - 1) not in bridge:
            lin.backward_data(w["dH0"], w["dZ%d" % (L - 1)], bp)

    def _backward_encoder(self, w, phase2, eps, B, bp, enc_w, enc_b, part="all"):
        hp, p, L = self.hp, self.p, self.p.L
        if part != "enc_lo":
            self._backward_encoder_head(w, phase2, eps, B, bp)
            hook = getattr(self, "_slice_hook", None) if part == "all" else None
            lo_layer = self.enc_cut if (part == "enc_hi" or hook is not None) else 0
            self._backward_encoder_layers(w, phase2, bp, enc_w, enc_b, L - 1, lo_layer)
            if part == "enc_hi":
                self._join()
                return
            if hook is not None:
                self._flush_deferred()
                hook("enc_hi")
                self._backward_encoder_layers(w, phase2, bp, enc_w, enc_b, self.enc_cut - 1, 0)
        else:
            self._backward_encoder_layers(w, phase2, bp, enc_w, enc_b, self.enc_cut - 1, 0)
        self._join()
        if MULTI_REDUCE:
            if self._multired is None:
                self._multired = ops.MultiReduce(self.device)
            self._multired.run(self._pending_red, self._pending_rs, bp)

    def _backward_encoder_head(self, w, phase2, eps, B, bp):
        hp, p, L = self.hp, self.p, self.p.L
        # ---------------- reparametrisation + KL ----------------
        k, d = p.edge_num[L], p.lat_dim[L - 1]
        P = self.params
        bridge = getattr(self, "_bridge_levels", ())
        self._dp_done = set()        # layers whose dP (gradient behind the LeakyReLU) is already in place
        i = L - 1
        lin = self.enc_lin[i]
        base = np.arange(2 * d).astype(np.int32)
        if i in bridge:
            # decoder head's data gradient -> reparametrisation + KL backward -> encoder head's data gradient (and the last
            # encoder layer's LeakyReLU backward where its consumer takes the gradient behind the activation): one launch
            di, _, _, n_in, n_out, y_es, y_off, ykey = self._bridge_geom(i)
            fus = self._enc_fusable(i)
            ops.latent_bridge_bwd(w["d" + ykey], P["dec.latent_dec_layers.%d.weight" % di], w["ZD%d" % i], eps[i],
                                  P["enc.latent_enc_layers.%d.weight" % i], w["dZD%d" % i], w[("dP%d" if fus else "dA%d") % i],
                                  k, n_in, d, n_out, n_in, y_es, y_off, B, bp, hp["kl_w"] / (B * k),
                                  act=w["A%d" % i] if fus else None)
            if fus:
                self._dp_done.add(i)
        else:
            ops.reparam_kl_bwd(w["ZD%d" % i], eps[i], w["dZ%d" % i], w["dZD%d" % i], k, d, B, bp, True, hp["kl_w"] / (B * k))
        if phase2:
            k0, d0 = p.edge_num[1], p.lat_dim[0]
            if 0 in bridge:              # the shallow level: down to the head's pre-activation gradient, layer 0 adds it later
                di, _, _, n_in, n_out, y_es, y_off, ykey = self._bridge_geom(0)
                ops.latent_bridge_bwd(w["d" + ykey], P["dec.latent_dec_layers.%d.weight" % di], w["ZD0"], eps[0], None,
                                      w["dZD0"], None, k0, n_in, d0, n_out, n_in, y_es, y_off, B, bp,
                                      hp["shallow_kl_w"] / (B * k0))
            else:
                ops.reparam_kl_bwd(w["ZD0"], eps[0], w["dZ0"], w["dZD0"], k0, d0, B, bp, True, hp["shallow_kl_w"] / (B * k0))
        # ---------------- encoder ----------------
        self._wgrad(lin, w["dZD%d" % i], w["A%d" % i], "enc.latent_enc_layers.%d.weight" % i, bp, w,
                    (w["dZD%d" % i], "enc.latent_enc_layers.%d.bias" % i, ("elinb", i), base, 2 * d, k, 2 * d))
        if i not in bridge:
            self._enc_grad_into(i, lambda dst, **kw: lin.backward_data(w["dZD%d" % i], dst, bp, **kw), w)

    def _enc_fusable(self, i):
        """Layer i's LeakyReLU backward can ride in the launch that produces its output gradient (the consumer must not
        need the gradient in front of the activation: pool folded in, weight gradient on the pooled rows)."""
        c = self.enc_conv[i]
        return FUSE_LEAKY_BWD and c.fused_pool and c.wg_pooled_dz

    def _enc_grad_into(self, i, launch, w, final=True, accumulate=False):
        """Run `launch(dst, **kw)` -- a data-gradient GEMM whose result is dL/dA[i] -- so that dP[i] = dL/dA[i] * leaky'(A[i])
        ends up in w["dP%d" % i] without a launch of its own where that is possible: the GEMM writes dP[i] and applies the
        factor in its epilogue (`final`: no other launch adds to it later; `accumulate`: this one adds to what is there)."""
        if not self._enc_fusable(i):
            launch(w["dA%d" % i], **(dict(accumulate=True) if accumulate else {}))
            return
        kw = dict(accumulate=True) if accumulate else {}
        if not final:
            launch(w["dP%d" % i], **kw)                       # a later launch accumulates and applies the factor
            return
        if not launch(w["dP%d" % i], act=w["A%d" % i], **kw):
            ops.leaky_bwd(w["dP%d" % i], w["A%d" % i], w["dP%d" % i], w["dP%d" % i].numel())      # (v3 pick: no such epilogue)
        self._dp_done.add(i)

    def _backward_encoder_layers(self, w, phase2, bp, enc_w, enc_b, first, last):
        """Encoder layers first, first-1, .., last (backward order)."""
        hp, p, L = self.hp, self.p, self.p.L
        for i in range(first, last - 1, -1):
            if i == 0 and phase2:
                lin = self.enc_lin[0]
                k0, d0 = p.edge_num[1], p.lat_dim[0]
                base = np.arange(2 * d0).astype(np.int32)
                self._wgrad(lin, w["dZD0"], w["A0"], "enc.latent_enc_layers.0.weight", bp, w,
                            (w["dZD0"], "enc.latent_enc_layers.0.bias", ("elinb", 0), base, 2 * d0, k0, 2 * d0))
                self._enc_grad_into(0, lambda dst, **kw: lin.backward_data(w["dZD0"], dst, bp, **kw), w, accumulate=True)
            t = p.tsteps[i + 1]
            conv = self.enc_conv[i]
            x = w["X0"] if i == 0 else w["A%d" % (i - 1)]
            base = (np.arange(p.enc_cout[i]) * t).astype(np.int32)
            bias = (w["dY%d" % i], enc_b[i], ("econvb", i), base, p.enc_cout[i], t, 1)
            if conv.fused_pool:
                # gradient w.r.t. the pooled pre-activation: feeds the backward-data GEMM and -- the rows a pool group
                # averages share their gradient -- the deduplicated weight gradient and the bias sums
                # (layer 0 without the pooled weight gradient: nothing reads dP -- no data gradient is wanted and the weight
                # gradient takes dY from the pool's own backward)
                if i not in self._dp_done and (i > 0 or conv.wg_pooled_dz):
                    ops.leaky_bwd(w["dA%d" % i], w["A%d" % i], w["dP%d" % i], w["dA%d" % i].numel())
                if conv.wg_pooled_dz:
                    rows_p = conv.pool_dst_base[conv.pool_rmap].astype(np.int32)
                    bias_p = (w["dP%d" % i], enc_b[i], ("econvbp", i), rows_p, p.enc_cout[i], t, 1,
                              conv.pool_scale[conv.pool_rmap])
                    self._wgrad(conv, w["dP%d" % i], x, enc_w[i], bp, w, bias_p, src=x, tag="enc%d" % i)
                else:
                    self._wgrad(conv, w["dY%d" % i], x, enc_w[i], bp, w, bias,
                                before=lambda i=i, t=t: self.enc_pool[i].backward(w["dA%d" % i], w["A%d" % i],
                                                                                  w["dY%d" % i], t, bp))
                if i > 0:
                    # (layer 0 in phase 2: the shallow head's gradient is added afterwards, by the launch that applies the factor)
                    self._enc_grad_into(i - 1, lambda dst, i=i, **kw: conv.backward_data(w["dP%d" % i], dst, bp, **kw), w,
                                        final=not (i - 1 == 0 and phase2))
                self._flush_delayed()
            else:
                self.enc_pool[i].backward(w["dA%d" % i], w["A%d" % i], w["dY%d" % i], t, bp)
                self._wgrad(conv, w["dY%d" % i], x, enc_w[i], bp, w, bias)
                if i > 0:
                    conv.backward_data(w["dY%d" % i], w["dA%d" % (i - 1)], bp)

    # ------------------------------------------------------------------------------------------
    # inference pieces (test()/gen_seq()/_decode)
    # ------------------------------------------------------------------------------------------
    def load_latents(self, w, z0, z_deep):
        """z0 (B,k1,d0), z_deep (B,kL,d) device tensors -> Z0 / Z{L-1} SoA."""
        B, bp, L = w["B"], w["bp"], self.p.L
        ops.aos_to_soa(z0.contiguous(), w["Z0"], B, 1, z0.shape[1] * z0.shape[2], bp)
        ops.aos_to_soa(z_deep.contiguous(), w["Z%d" % (L - 1)], B, 1, z_deep.shape[1] * z_deep.shape[2], bp)

    def decode(self, z0, z_deep, repack=True, need_c6=True):
        """_decode (seq_two_hier_sa_vae.py:436-474): returns AoS (c6 (B,T,24,6), R (B,T,24,3,3), pos (B,T,24,3)).
        need_c6=False skips the frame-major copy of the 6D decoder output (c6 is None): test()/gen_seq only use pos."""
        B = z0.shape[0]
        w = self.workspace(B)
        if repack:
            self.pack_weights(side=False)
        self.load_latents(w, z0, z_deep)
        self.decode_feats(w)
        T = self.p.T
        c6 = torch.empty(B, T, 24, 6, device=self.device) if need_c6 else None
        R = torch.empty(B, T, 24, 3, 3, device=self.device)
        pos = torch.empty(B, T, 24, 3, device=self.device)
        ops.decode_tail_fwd(w["D%d" % (self.p.L - 1)], self.offsets, T, B, w["bp"], c6, R, pos)
        return c6, R, pos

    def encode_batch(self, r6, repack=True):
        """Encoder.forward on (B,T,144): returns latent (B, C, T') and the list of z distributions (B,k,2d)."""
        B, T = r6.shape[0], r6.shape[1]
        w = self.workspace(B)
        bp, L, p = w["bp"], self.p.L, self.p
        if repack:
            self.pack_weights(side=False)
            for i in range(1, L - 1):
                self.enc_lin[i].pack(self.params["enc.latent_enc_layers.%d.weight" % i])
        ops.aos_to_soa(r6.contiguous(), w["X0"], B, T, 144, bp)
        self.encode(w, range(L))
        zs = []
        for i in range(L):
            k, d = p.edge_num[i + 1], p.lat_dim[i]
            out = torch.empty(B, k * 2 * d, device=self.device)
            ops.soa_to_aos(w["ZD%d" % i], out, B, 1, k * 2 * d, bp)
            zs.append(out.reshape(B, k, 2 * d))
        t = p.tsteps[L]
        rows = w["A%d" % (L - 1)].shape[0]
        lat = torch.empty(B, rows, device=self.device)
        ops.soa_to_aos(w["A%d" % (L - 1)], lat, B, 1, rows, bp)
        return lat.reshape(B, rows // t, t), zs
