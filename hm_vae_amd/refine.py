"""Test-time optimisation of latent codes (and then of a copy of the decoder) against partially observed motion.

The inner loop of the reference's interpolation / completion routines (seq_two_hier_sa_vae.py:1270-1541; the loop body is
:1356-1429, its configuration configs/len_64_test_interpolation.yaml:68-84):

    out = _decode(z_list)                                              # decoder + rot6D lift + FK
    l   = rec_6d_w * masked_mse(6D) + rec_rot_w * masked_mse(R) + rec_pose_w * masked_mse(pos)     (:717-735, :1363-1365)
        + reg_w * (mse(z_shallow, 0) + mse(z_deep, 0))                                             (:1366-1367)
        + reg_w_decoder * sum_p mse(p, p_original)                  [optimize_decoder only]        (:1369-1373)
    iterations 0..50 step Adam on the latents, later ones Adam (lr * 0.001) on the decoder copy     (:1395-1417)

Here every iteration is a fixed launch sequence on the training step's buffers: latents -> SoA, decoder GEMMs, the masked
decode-tail loss kernel (forward + FK/lift backward in one launch), the decoder's backward-data chain down to the two
latents (weight-gradient GEMMs only in the decoder phase), the l2-anchor kernel for the regularisers, one finalize launch,
HipAdam.  The reference optimises a deep copy of the decoder; this class snapshots the decoder's tensors (the snapshot is also
the regulariser's anchor) and `restore()` puts them back, so the model is unchanged afterwards.

The released routine cannot run (stale call signatures), so its parity is pinned against the CPU oracle's
restatement of the same loop only (`OracleVAE.refine_forward`), not against reference outputs: "parity unpinned".
The trajectory term (`optimize_trajectory`, :1376-1389): the decoded window's joint positions go through the frozen
TrajectoryModel (standardise -> encoder -> root velocity), the velocities are accumulated into a root trajectory and its
differences between "key frames" are held against the ground truth's (cal_key_frame_root_loss :1031-1045, restated AS
RELEASED: see hmvae_keyframe_root_loss in include/hmvae.h); its gradient reaches the latents through the trajectory encoder's
data-gradient chain (traj_engine.backward_input), the standardisation, the FK / lift backward and the decoder.
"""
import os

import numpy as np
import torch

from . import _lib, graph_step, ops
from .optim import HipAdam
from .seq_two_hier_sa_vae import get_opt_scheduler

DECODER_PHASE_AFTER = 50           # seq_two_hier_sa_vae.py:1393 `prev_epochs`


class LatentRefiner:
    """One batch of windows.  target_rot_mat (B,T,24,3,3), mask (B,T,24) -- 1 where the frame/joint is observed."""

    def __init__(self, model, hp, target_rot_mat, mask, z_shallow=None, z_deep=None, use_graph=None,
                 trajectory_model=None, root_trans=None):
        """trajectory_model: the trained TrajectoryModel (hm_vae_amd.trajectory_pred_model, on the device) and root_trans
        (B,T,3) the windows' ground-truth root translation -- both needed when hp['optimize_trajectory'] is set."""
        self.model, self.hp = model, hp
        # an iteration is ~35 (latent phase) / ~75 (decoder phase) launches of a few microseconds each at the reference's
        # batch of one window: replayed from a hipGraph per phase (the phase's first iteration runs eagerly -- it creates
        # the optimiser's tables -- the second is captured).  Everything a graph addresses (latents, Adam state, targets,
        # mask, anchor) lives for the refiner's lifetime; begin() loads a new window into the same buffers, so ONE refiner
        # and its two graphs serve every window batch of the same shape (TwoHierSAVAEModel.refine_window keeps them).
        self.use_graph = (os.environ.get("HMVAE_REFINE_GRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        self._graphs, self._eager_done = {}, {False: 0, True: 0}
        self.eng = eng = model.engine()
        p, dev = eng.p, eng.device
        B, T = target_rot_mat.shape[0], target_rot_mat.shape[1]
        if T != p.T:
            raise ValueError("window length %d != train_seq_len %d" % (T, p.T))
        self.B, self.T = B, T
        self.w = w = eng.workspace(B)
        bp = w["bp"]
        self.traj = None
        if hp.get('optimize_trajectory', False):
            self._init_trajectory(trajectory_model, dev, B, T, bp)
        if "MK" not in w:
            w["MK"] = torch.zeros(24 * T, bp, device=dev)
            nb = ops.l2_anchor_blocks()
            w["REGP"] = torch.zeros(2 * nb, device=dev)
            w["DREGP"] = torch.zeros(max(1, len(self._decoder_names())) * nb, device=dev)
            w["ZERO1"] = torch.zeros(1, device=dev)
        L = p.L
        k0, d0, kd, dd = p.edge_num[1], p.lat_dim[0], p.edge_num[L], p.lat_dim[L - 1]
        self.z_shallow = torch.nn.Parameter(torch.zeros(B, k0, d0, device=dev))
        self.z_deep = torch.nn.Parameter(torch.zeros(B, kd, dd, device=dev))
        self.z_shallow.grad = torch.zeros_like(self.z_shallow)
        self.z_deep.grad = torch.zeros_like(self.z_deep)
        self.gen_opt = HipAdam([self.z_shallow, self.z_deep], lr=hp['opt_lr'], weight_decay=hp['weight_decay'])
        self.optimize_decoder = bool(hp.get('optimize_decoder', False))
        self.anchor = None
        self.gen_opt_for_decoder = None
        if self.optimize_decoder:
            names = self._decoder_names()
            self.anchor = {n: eng.params[n].detach().clone() for n in names}
            eng.grad_layout()
            self.gen_opt_for_decoder = HipAdam([eng.params[n] for n in names], lr=hp['opt_lr'] * 0.001,
                                               weight_decay=hp['weight_decay'])
        self.losses = torch.zeros(8, device=dev)
        self.begin(target_rot_mat, mask, z_shallow, z_deep, root_trans=root_trans)

    def _init_trajectory(self, tm, dev, B, T, bp):
        """Buffers and the frozen trajectory engine of the `optimize_trajectory` term (:1376-1389)."""
        from .traj_engine import TrajEngine
        from .trajectory_pred_model import JOINT_POS_SLICE, ROOT_V_SLICE
        tm = getattr(tm, "model", tm)                 # a Trainer is accepted like the reference's `trajectory_trainer`
        if tm is None:
            raise ValueError("optimize_trajectory needs the trained TrajectoryModel (trajectory_model=...)")
        if not tm.hp['trajectory_input_joint_pos']:
            raise NotImplementedError("optimize_trajectory with a rotation-input TrajectoryModel")
        teng = TrajEngine(dict(tm.hp, train_seq_len=T), dict(tm.named_parameters()),
                          tm.std_vals[0, ROOT_V_SLICE[0]:ROOT_V_SLICE[1]].cpu().numpy(), dev,
                          parents=tm.fk_layer.parent_list, input_grad=True)

        def f(*shape):
            return torch.zeros(*shape, device=dev)
        self.traj = dict(eng=teng, w=teng.workspace(B),
                         jm=tm.mean_vals[0, JOINT_POS_SLICE[0]:JOINT_POS_SLICE[1]].to(dev).contiguous(),
                         js=tm.std_vals[0, JOINT_POS_SLICE[0]:JOINT_POS_SLICE[1]].to(dev).contiguous(),
                         mean3=tm.mean_vals[0, ROOT_V_SLICE[0]:ROOT_V_SLICE[1]].to(dev).contiguous(),
                         std3=tm.std_vals[0, ROOT_V_SLICE[0]:ROOT_V_SLICE[1]].to(dev).contiguous(),
                         c6=f(B, T, 24, 6), R=f(B, T, 24, 3, 3), pos=f(B, T, 24, 3), xn=f(B, T, 72), dpos=f(B, T, 72),
                         d6=f(B, T, 144), d6s=f(144 * T, bp), gt=f(3 * T, bp),
                         part=f(ops.keyframe_root_loss_blocks(T, B, bp)), l=f(1), n=0)

    def begin(self, target_rot_mat, mask, z_shallow=None, z_deep=None, root_trans=None):
        """Load a window batch (same shape as the refiner was built for) and start over: iteration 0, fresh Adam state,
        the anchor = the model's current decoder.  No buffer a captured graph addresses is replaced."""
        eng, w, hp = self.eng, self.w, self.hp
        p, dev, B, T, bp = eng.p, eng.device, self.B, self.T, self.w["bp"]
        R = target_rot_mat.to(device=dev, dtype=torch.float32).contiguous()
        if tuple(R.shape[:2]) != (B, T) or tuple(mask.shape) != (B, T, 24):
            raise ValueError("this refiner serves windows of shape (%d, %d, 24, 3, 3) with a (B, T, 24) mask" % (B, T))
        # targets: 6D = the first two columns of R (:1279-1280), R, joint positions by FK (:1282)
        r6 = torch.stack((R[..., 0], R[..., 1]), dim=-2).reshape(B, T, 144).contiguous()
        eng.stage_batch(w, r6, R.reshape(B, T, 216))
        ops.aos_to_soa(mask.to(device=dev, dtype=torch.float32).contiguous(), w["MK"], B, T, 24, bp)
        if self.traj is not None:
            if root_trans is None or tuple(root_trans.shape) != (B, T, 3):
                raise ValueError("optimize_trajectory needs root_trans of shape (%d, %d, 3)" % (B, T))
            tj = self.traj
            ops.aos_to_soa(root_trans.to(device=dev, dtype=torch.float32).contiguous(), tj["gt"], B, T, 3, bp)
            # N of the mean (:1043): three components per pair of consecutive selected indices, all windows together; as
            # released the selection is row 0 of each window's (T, 24) mask, i.e. the joints observed in frame 0 (< T)
            sel = (mask[:, 0, :min(24, T)] != 0).sum(dim=1)
            n = 3 * int(torch.clamp(sel - 1, min=0).sum())
            if tj["n"] not in (0, n) and self._graphs:
                raise ValueError("the captured iteration bakes in the number of key-frame pairs (%d), this batch has %d"
                                 % (tj["n"] // 3, n // 3))
            tj["n"] = n
            tj["eng"].pack_weights()                  # frozen: once per window batch
        with torch.no_grad():
            if z_shallow is None:                     # :1317-1318: the deep code first, then the shallow one
                z_deep = torch.randn(self.z_deep.shape, device=dev)
                z_shallow = torch.randn(self.z_shallow.shape, device=dev)
            self.z_shallow.copy_(z_shallow.to(device=dev, dtype=torch.float32))
            self.z_deep.copy_(z_deep.to(device=dev, dtype=torch.float32))
        self.z_shallow.grad.zero_()
        self.z_deep.grad.zero_()
        self.gen_opt.reset_state()
        self.gen_scheduler = get_opt_scheduler(self.gen_opt, hp)
        if self.optimize_decoder:
            with torch.no_grad():
                for n, t in self.anchor.items():
                    t.copy_(eng.params[n])
            for n in self.anchor:
                eng.params[n].grad = eng.grad_view(n)
            self.gen_opt_for_decoder.reset_state()
            self.gen_scheduler_for_decoder = get_opt_scheduler(self.gen_opt_for_decoder, hp)
        self.it = 0
        eng.pack_decoder()                            # (the latent phase's captured iteration does not pack)
        self._packed = True
        self._decoder_moved = False
        w["DREGP"].zero_()
        self._restored = False

    def _decoder_names(self):
        """The trainable tensors of `dec` proper: the convolutions and ALL latent heads.  The two middle heads feed nothing
        (seq_two_hier_sa_vae.py:264-272) but the reference's decoder optimiser holds them (:1343-1344) and l_reg_decoder
        sums over them (:1371-1373): their gradient is that regulariser's (zero at first) plus Adam's weight decay, so
        they drift and enter the reported l_reg_decoder / l_total -- reproduced here.  NOT reproduced: the reference's
        deep copy of the decoder also carries a copy of the ENCODER (`dec.enc.*`, :180), 6.3 M more parameters that
        decay the same way and add to the reported regulariser without ever touching an output; with weight_decay != 0
        columns 0, 1 and 6 of the history therefore differ from what the released routine would print (it cannot run:
        parity unpinned either way)."""
        eng, L = self.eng, self.eng.p.L
        _, _, dconv = eng.names()
        names = []
        for i in range(L):
            names.append(dconv[i] + ".weight")
            if eng.p.dec_bias[i]:
                names.append(dconv[i] + ".bias")
        for i in range(L):
            names += ["dec.latent_dec_layers.%d.weight" % i, "dec.latent_dec_layers.%d.bias" % i]
        return names

    def _dead_heads(self):
        L = self.eng.p.L
        return ["dec.latent_dec_layers.%d.%s" % (i, s) for i in range(1, L - 1) for s in ("weight", "bias")]

    # ------------------------------------------------------------------------------------------
    def loss_and_grads(self, decoder_phase):
        """One forward + backward.  Leaves d l_total / d z in z_*.grad and (decoder_phase) d l_total / d decoder tensors in
        their gradient views; returns the device tensor [total, reg terms weighted, rec_6d, rec_rot, rec_pose, l_reg,
        l_reg_decoder, 0]."""
        eng, w, hp = self.eng, self.w, self.hp
        p, L, B, T, bp = eng.p, eng.p.L, self.B, self.T, self.w["bp"]
        if decoder_phase or not self._packed:
            eng.pack_decoder()
            self._packed = True
        eng.load_latents(w, self.z_shallow.data, self.z_deep.data)
        eng.decode_feats(w)
        n6, nR, nP = B * T * 144, B * T * 216, B * T * 72
        ops.decode_tail_loss(w["D%d" % (L - 1)], w["X0"], w["GR"], w["GP"], eng.offsets, T, B, bp, hp["rec_6d_w"] / n6,
                             hp["rec_rot_w"] / nR, hp["rec_pose_w"] / nP, w["dD%d" % (L - 1)], w["MSEP"], mask=w["MK"])
        if self.traj is not None and self.traj["n"] > 0:
            self._trajectory_term(w, L, B, T, bp)
        _, _, dconv = eng.names()
        eng._skip_wgrad = not decoder_phase
        try:
            eng._turn = 0
            eng._backward_decoder(w, True, bp, dconv)
            eng._join()
        finally:
            eng._skip_wgrad = False
        g0, gd = self.z_shallow.grad, self.z_deep.grad
        ops.soa_to_aos(w["dZ0"], g0.view(B, -1), B, 1, g0[0].numel(), bp)
        ops.soa_to_aos(w["dZ%d" % (L - 1)], gd.view(B, -1), B, 1, gd[0].numel(), bp)
        reg_w = float(hp.get("reg_w", 0.0))
        nb = ops.l2_anchor_blocks()
        for k, (z, g) in enumerate(((self.z_shallow, g0), (self.z_deep, gd))):
            n = z.numel()
            ops.l2_anchor(z.data, None, 2.0 * reg_w / n, 1.0 / n, g, w["REGP"][k * nb:(k + 1) * nb])
        reg_dec = float(hp.get("reg_w_decoder", 0.0)) if self.optimize_decoder else 0.0
        if self.optimize_decoder and (decoder_phase or self._decoder_moved):
            # (before the first decoder update every term is exactly zero -- p == p_original -- and DREGP still holds zeros)
            if decoder_phase:
                for name in self._dead_heads():       # no kernel writes their gradient: the regulariser's term is all of it
                    eng.grad_view(name).zero_()
            for k, name in enumerate(self._decoder_names()):
                t = eng.params[name]
                n = t.numel()
                ops.l2_anchor(t.data, self.anchor[name], 2.0 * reg_dec / n, 1.0 / n,
                              eng.grad_view(name) if decoder_phase else None, w["DREGP"][k * nb:(k + 1) * nb])
            self._decoder_moved = True
        dregp = w["DREGP"] if self.optimize_decoder else w["ZERO1"]
        # total = w6 l6 + wR lR + wP lP + reg_w * l_reg + reg_w_decoder * l_reg_decoder  (:1388-1389)
        ops.vae_loss_finalize(w["MSEP"], w["REGP"], dregp, 1.0 / n6, 1.0 / nR, 1.0 / nP, 1.0, 1.0, hp["rec_6d_w"],
                              hp["rec_rot_w"], hp["rec_pose_w"], reg_dec, reg_w, self.losses)
        if self.traj is not None and self.traj["n"] > 0:          # + reg_w_trajectory * l_rec_trajectory (:1388-1389)
            self.losses[7:8].copy_(self.traj["l"])
            self.losses[0:1].add_(self.traj["l"], alpha=float(hp.get("reg_w_trajectory", 0.0)))
        return self.losses

    def _trajectory_term(self, w, L, B, T, bp):
        """l_rec_trajectory (:1376-1389) and its gradient, ADDED to dL/d(decoder output) before the decoder's backward."""
        eng, hp, tj = self.eng, self.hp, self.traj
        teng, tw = tj["eng"], tj["w"]
        dec = w["D%d" % (L - 1)]
        # sampled_seq_test (trajectory_pred_model.py:397-439): FK of the decoded rotations, standardised, through the encoder
        ops.decode_tail_fwd(dec, eng.offsets, T, B, bp, tj["c6"], tj["R"], tj["pos"])
        torch.sub(tj["pos"].view(B, T, 72), tj["jm"], out=tj["xn"])
        tj["xn"].div_(tj["js"])
        teng.encode(tw, tj["xn"])
        w_t = float(hp.get("reg_w_trajectory", 0.0))
        nb = ops.keyframe_root_loss(tw["V"], tj["gt"], w["MK"], tj["mean3"], tj["std3"], T, B, bp, 2.0 * w_t / tj["n"],
                                    tw["dV"], tj["part"])
        ops.finalize_sum(tj["part"], nb, 1, 0, 1.0 / tj["n"], tj["l"])
        if w_t == 0.0:
            return
        # back: trajectory encoder (data gradients only) -> de-standardise -> FK / lift backward -> + dL/d(decoder output)
        teng.backward_input(tw, bp)
        ops.soa_to_aos(tw["dX0"], tj["dpos"], B, T, 72, bp)
        tj["dpos"].div_(tj["js"])
        ops.fk_bwd(tj["c6"].view(B * T, 24, 6), 6, eng.offsets, 0, eng.parents, B * T, None, tj["dpos"].view(B * T, 24, 3),
                   tj["d6"].view(B * T, 24, 6))
        ops.aos_to_soa(tj["d6"], tj["d6s"], B, T, 144, bp)
        w["dD%d" % (L - 1)].add_(tj["d6s"])

    def _graph_for(self, decoder_phase, opt):
        g = self._graphs.get(decoder_phase)
        if g is not None or not self.use_graph or self._eager_done[decoder_phase] < 1 or not graph_step.may_capture():
            return g
        torch.cuda.synchronize()
        host_before = dict(opt._host_step)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss_and_grads(decoder_phase)
            opt.step()
        opt._host_step = host_before          # capture records the counter increments without executing them
        opt.resync_device_steps()
        self._graphs[decoder_phase] = g
        return g

    def step(self):
        """One iteration of the loop (:1356-1417).  Returns the loss tensor of the iterate BEFORE the update."""
        decoder_phase = self.optimize_decoder and self.it > DECODER_PHASE_AFTER
        if decoder_phase:
            opt, sched = self.gen_opt_for_decoder, self.gen_scheduler_for_decoder
        else:
            opt, sched = self.gen_opt, self.gen_scheduler
        g = self._graph_for(decoder_phase, opt)
        if g is None:
            self.loss_and_grads(decoder_phase)
            opt.step()
            self._eager_done[decoder_phase] += 1
        else:
            g.replay()
            opt.advance_host_steps(1)
        if sched is not None:
            sched.step()
            opt.sync_lr()
        self.it += 1
        return self.losses

    def run(self, iterations=None):
        """The whole loop; returns the per-iteration losses as an (iterations, 8) host array (one read-back at the end)."""
        n = int(self.hp["opt_it"] if iterations is None else iterations)
        hist = torch.zeros(n, 8, device=self.eng.device)
        for i in range(n):
            hist[i].copy_(self.step())
        return hist.cpu().numpy()

    def decode(self):
        """The current estimate: (cont6d (B,T,24,6), R (B,T,24,3,3), pos (B,T,24,3))."""
        with torch.no_grad():
            self.eng.pack_decoder()
            return self.eng.decode(self.z_shallow.data, self.z_deep.data, repack=False)

    def __del__(self):
        try:
            graph_step.retire(*self._graphs.values())
        except Exception:
            pass

    def restore(self):
        """Put the original decoder back (the reference optimised a deep copy, :1342)."""
        if self.anchor is not None:
            with torch.no_grad():
                for n, t in self.anchor.items():
                    self.eng.params[n].copy_(t)
            self.eng.pack_decoder()
        self._restored = True                          # the graphs stay: begin() reuses them for the next window


def temporal_mask(T, window):
    """seq_two_hier_sa_vae.py:1300-1304: every `window`-th frame and the last one are observed."""
    m = np.zeros(T, dtype=np.float32)
    m[::window] = 1
    m[-1] = 1
    return m
