"""The TrajectoryModel step as a fixed sequence of HIP kernel launches (trajectory_pred_model.py:196-265).

joint positions (B,T,72) -> four stride-1 skeleton convolutions (K = 31, reflect pad) each followed by SkeletonPool(mean) +
LeakyReLU(0.2) -> Linear(7*48 -> 3) per frame -> root velocity; losses: MSE on the velocity and MSE on the accumulated
trajectory (gen_motion_w_trajectory, :292-305); then the hand-ordered backward of all of it (the reference calls
`l_total.backward()` inside forward, :262).  Same conventions as engine.py: SoA buffers [rows][Bp] with row = channel*T + t,
gradients in views of one flat fp32 bucket, weight-gradient GEMMs on side streams.

The convolutions reuse the VAE's masked gather-GEMMs un-fused: with 31 taps the composite (pool-folded) relation of
ops.conv_gemm does not apply (its pack kernel holds <= 16 real taps in registers).
"""
import numpy as np
import torch

from . import _lib, ops, topology
from .engine import _StepBase


class TrajPlan:
    """Static structure of the TrajectoryModel encoder (trajectory_pred_model.py:45-104)."""

    def __init__(self, hp, parents=topology.SMPL24_PARENTS):
        L = int(hp["num_layers"])
        self.hp, self.L, self.T = hp, L, int(hp["train_seq_len"])
        self.K = int(hp["kernel_size"])
        self.pad = (self.K - 1) // 2
        self.pad_mode = {"zeros": "zeros", "constant": "zeros", "reflection": "reflect", "reflect": "reflect"}[hp["padding_mode"]]
        c0 = 3 if hp["trajectory_input_joint_pos"] else 6
        self.ch_base = [c0 * (2 ** i) for i in range(L + 1)]
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
        self.d_model = self.ch_base[-1]
        self.latent_ch = self.d_model * self.edge_num[L]
        self.fc_in = self.d_model * 7                 # trajectory_pred_model.py:192 hard-codes the 7 coarse edges
        if self.latent_ch != self.fc_in:
            raise ValueError("fc_mapping expects 7 coarse edges, the skeleton pools to %d" % self.edge_num[L])


class TrajEngine(_StepBase):
    """Owns device tables + workspaces; borrows the parameter tensors (dict reference-name -> Parameter)."""

    def __init__(self, hp, params, std_root, device, parents=topology.SMPL24_PARENTS, input_grad=False):
        """input_grad: also build the first convolution's backward-data side (the test-time optimisation loop's trajectory
        term differentiates the frozen model with respect to its INPUT, seq_two_hier_sa_vae.py:1376-1389)."""
        _lib.load()
        self._init_step_base(device)
        self.input_grad = bool(input_grad)
        self.hp = hp
        self.p = p = TrajPlan(hp, parents)
        self.params = params
        self.std_root = torch.as_tensor(np.asarray(std_root, dtype=np.float32)).reshape(3).contiguous().to(device)
        self.conv, self.pool = [], []
        T = p.T
        for i in range(p.L):
            self.conv.append(ops.conv_gemm(device, p.neigh[i], p.enc_cin[i], p.enc_cout[i], p.K, 1, p.pad, p.pad_mode, T,
                                           need_dx=(i > 0 or input_grad)))
            self.pool.append(ops.PoolOp(device, p.pooling[i], p.edge_num[i], p.enc_cout[i] // p.edge_num[i]))
        # fc_mapping over the frames: one "edge" that owns every latent channel, one tap, T positions
        self.head = ops.conv_gemm(device, [[0]], p.latent_ch, 3, 1, 1, 0, p.pad_mode, T, need_dx=True)
        self._ws = {}
        self._multipack = None
        self.flops_per_seq = 2 * sum(op.useful_macs_per_col for op in self.conv + [self.head])

    # ------------------------------------------------------------------------------------------
    def workspace(self, B):
        if B in self._ws:
            return self._ws[B]
        if torch.device(self.device).type != "cuda":
            raise _lib.HmvaeError("the engine computes on a HIP device only (tables may be planned on CPU for inspection)")
        p, dev, T = self.p, self.device, self.p.T
        bp = ops.pad_batch(B)

        def buf(rows):
            return torch.zeros(int(rows), bp, dtype=torch.float32, device=dev)

        w = {"bp": bp, "B": B}
        w["X0"], w["GV"] = buf(p.enc_cin[0] * T), buf(3 * T)
        for i in range(p.L):
            w["Y%d" % i], w["dY%d" % i] = buf(p.enc_cout[i] * T), buf(p.enc_cout[i] * T)
            pc = p.ch_base[i + 1] * p.edge_num[i + 1]
            w["A%d" % i], w["dA%d" % i] = buf(pc * T), buf(pc * T)
        w["V"], w["dV"] = buf(3 * T), buf(3 * T)
        if self.input_grad:
            w["dX0"] = buf(p.enc_cin[0] * T)
        nb = ops.traj_loss_blocks(T, B, bp)
        w["LP"] = torch.zeros(nb, 2, device=dev)
        w["losses"] = torch.zeros(9, device=dev)
        all_ops = self.conv + [self.head]
        tot = sum(op.slab_floats(bp) for op in all_ops)
        big = torch.empty(max(tot, 1), device=dev)
        o = 0
        w["slabs"] = {}
        for op in all_ops:
            n = op.slab_floats(bp)
            w["slabs"][id(op)] = big[o:o + n]
            o += n
        self._ws[B] = w
        return w

    # ------------------------------------------------------------------------------------------
    def names(self):
        L = self.p.L
        return (["enc.layers.%d.0.weight" % i for i in range(L)], ["enc.layers.%d.0.bias" % i for i in range(L)])

    def grad_layout(self):
        """Flat gradient bucket in the reference's parameter order; every tensor is active from the first step."""
        if self._flat is not None:
            return self._flat
        enc_w, enc_b = self.names()
        order = []
        for wn, bn in zip(enc_w, enc_b):
            order += [wn, bn]
        order += ["fc_mapping.weight", "fc_mapping.bias"]
        offs, o = {}, 0
        for n in order:
            offs[n] = o
            o += self.params[n].numel()
            o = (o + 3) // 4 * 4
        self.flat_grad = torch.zeros(o, dtype=torch.float32, device=self.device)
        self._flat = (order, offs, o, o)
        return self._flat

    def assign_grads(self, phase2=False):
        order = self.grad_layout()[0]
        for n in order:
            self.params[n].grad = self.grad_view(n)

    def active_grad_floats(self, phase2=False):
        return self.grad_layout()[2]

    def pack_weights(self):
        """Regather the dense weights into MFMA-fragment order (applies the neighbour mask, skeleton.py:96)."""
        enc_w, _ = self.names()
        if self._multipack is None:
            self._multipack = ops.MultiPack(self.device)          # every layout of the model in one launch
        lst = []
        for op, wt in list(zip(self.conv, (self.params[n] for n in enc_w))) + [(self.head, self.params["fc_mapping.weight"])]:
            lst.append((op.fwd, wt))
            if op.bwd is not None:
                lst.append((op.bwd, wt))
        self._multipack.run(lst)

    # ------------------------------------------------------------------------------------------
    def encode(self, w, joint_pos):
        """Encoder.forward + fc_mapping (:220-235); the velocity lands in w["V"] (rows c*T + t)."""
        p, P, bp = self.p, self.params, w["bp"]
        B, T = joint_pos.shape[0], joint_pos.shape[1]
        enc_w, enc_b = self.names()
        ops.aos_to_soa(joint_pos, w["X0"], B, T, p.enc_cin[0], bp)
        x = w["X0"]
        for i in range(p.L):
            self.conv[i].forward(x, w["Y%d" % i], bp, bias=P[enc_b[i]])
            self.pool[i].forward(w["Y%d" % i], w["A%d" % i], T, bp)
            x = w["A%d" % i]
        self.head.forward(x, w["V"], bp, bias=P["fc_mapping.bias"])

    def predict(self, joint_pos, repack=True):
        """(B,T,72) normalised joint positions -> (B,T,3) normalised root velocity (the forward half of test(), :369-383)."""
        B, T = joint_pos.shape[0], joint_pos.shape[1]
        self._check_len(T)
        w = self.workspace(B)
        if repack:
            self.pack_weights()
        self.encode(w, joint_pos)
        out = torch.empty(B, T, 3, device=self.device)
        ops.soa_to_aos(w["V"], out, B, T, 3, w["bp"])
        return out

    def _check_len(self, T):
        if T != self.p.T:
            raise ValueError("sequence length %d != train_seq_len %d (the gather tables are built per length)" % (T, self.p.T))

    def train_step(self, joint_pos, root_v, backward=True, repack=True):
        """joint_pos (B,T,72), root_v (B,T,3): fp32 device tensors.  Returns the `losses` device tensor in the order
        TrajectoryModel.forward returns them: [total, 0, 0, 0, 0, 0, rec_root_v, 0, rec_root_trans]."""
        hp, p = self.hp, self.p
        B, T = joint_pos.shape[0], joint_pos.shape[1]
        self._check_len(T)
        w = self.workspace(B)
        bp = w["bp"]
        if repack:
            self.pack_weights()
        ops.aos_to_soa(root_v, w["GV"], B, T, 3, bp)
        self.encode(w, joint_pos)
        n = 3.0 * T * B
        wv = float(hp["rec_root_v_w"])
        wt = float(hp["rec_root_trans_w"]) if hp["use_accumulation_root_v"] else 0.0
        nb = ops.traj_loss(w["V"], w["GV"], self.std_root, T, B, bp, 2.0 * wv / n, 2.0 * wt / n, w["dV"], w["LP"])
        ops.traj_loss_finalize(w["LP"], nb, 1.0 / n, wv, wt, w["losses"])
        if backward:
            self._backward(w, bp)
            self.assign_grads()
        return w["losses"]

    def backward_input(self, w, bp):
        """dL/d(input) from w["dV"]: the data-gradient chain alone, down to w["dX0"] (no weight gradients: the model is frozen)."""
        if not self.input_grad:
            raise _lib.HmvaeError("this engine was built without the first layer's backward-data side (input_grad=True)")
        p, T, L = self.p, self.p.T, self.p.L
        self.head.backward_data(w["dV"], w["dA%d" % (L - 1)], bp)
        for i in range(L - 1, -1, -1):
            self.pool[i].backward(w["dA%d" % i], w["A%d" % i], w["dY%d" % i], T, bp)
            self.conv[i].backward_data(w["dY%d" % i], w["dA%d" % (i - 1)] if i > 0 else w["dX0"], bp)

    def _backward(self, w, bp):
        p, T = self.p, self.p.T
        enc_w, enc_b = self.names()
        L = p.L
        self._turn = 0                          # the same chain -> stream assignment every step
        base = (np.arange(3) * T).astype(np.int32)
        self._wgrad(self.head, w["dV"], w["A%d" % (L - 1)], "fc_mapping.weight", bp, w,
                    (w["dV"], "fc_mapping.bias", "headb", base, 3, T, 1))
        self.head.backward_data(w["dV"], w["dA%d" % (L - 1)], bp)
        for i in range(L - 1, -1, -1):
            self.pool[i].backward(w["dA%d" % i], w["A%d" % i], w["dY%d" % i], T, bp)
            x = w["X0"] if i == 0 else w["A%d" % (i - 1)]
            base = (np.arange(p.enc_cout[i]) * T).astype(np.int32)
            self._wgrad(self.conv[i], w["dY%d" % i], x, enc_w[i], bp, w,
                        (w["dY%d" % i], enc_b[i], ("convb", i), base, p.enc_cout[i], T, 1), tag="enc%d" % i)
            if i > 0:
                self.conv[i].backward_data(w["dY%d" % i], w["dA%d" % (i - 1)], bp)
            self._flush_delayed()               # weight gradients placed behind this data-gradient launch (ops.MaskedGemm.wg_delay)
        self._join()
