"""TrajectoryModel with the reference's surface (trajectory_pred_model.py:45-115, 176-440).

`Encoder` keeps the reference module tree (18 state_dict keys: conv mask/weight/bias + the pool matrices + fc_mapping) and
can be called on its own, module by module.  `TrajectoryModel.forward` hands the batch to the fixed HIP kernel sequence
of traj_engine.py, which runs forward, both losses and the whole backward and leaves the gradients in `.grad` — the same
contract as the reference, whose forward calls `l_total.backward()` itself (:262).
"""
import numpy as np
import torch
import torch.nn as nn

from . import topology
from .fk_layer import ForwardKinematicsLayer
from .seq_two_hier_sa_vae import get_scheduler, _parents_of  # noqa: F401  (get_scheduler is re-exported like the reference's)
from .skeleton import SkeletonUnpool, SkeletonPool, SkeletonConv, find_neighbor, SkeletonLinear, get_edges  # noqa: F401
from .traj_engine import TrajEngine, TrajPlan

ROOT_V_SLICE = (576, 579)        # columns of the mean/std table that hold the root velocity (:241, :294)
JOINT_POS_SLICE = (24 * 6 + 24 * 9, 24 * 6 + 24 * 9 + 24 * 3)


class Encoder(nn.Module):
    """trajectory_pred_model.py:45-115 — four stride-1 SkeletonConv + SkeletonPool + LeakyReLU blocks."""

    def __init__(self, args, topology_edges):
        super().__init__()
        plan = TrajPlan(args, _parents_of(topology_edges))
        self.plan = plan
        self.args = args
        self.topologies = plan.topo
        self.channel_base = plan.ch_base
        self.channel_list = plan.channel_list
        self.edge_num = plan.edge_num[:-1] + [len(plan.topo[-1])]      # reference quirk: last entry is len([]) = 0
        self.pooling_list = plan.pooling
        self.layers = nn.ModuleList()
        self.convs = []
        if args['extra_conv'] != 0:
            raise NotImplementedError("extra_conv > 0 is not used by any shipped config")
        for i in range(plan.L):
            conv = SkeletonConv(plan.neigh[i], in_channels=plan.enc_cin[i], out_channels=plan.enc_cout[i],
                                joint_num=plan.edge_num[i], kernel_size=plan.K, stride=1, padding=plan.pad,
                                padding_mode=args['padding_mode'], bias=True)
            pool = SkeletonPool(edges=plan.topo[i], pooling_mode=args['skeleton_pool'],
                                channels_per_edge=plan.enc_cout[i] // plan.edge_num[i], last_pool=(i == plan.L - 1))
            self.convs.append(conv)
            self.layers.append(nn.Sequential(conv, pool, nn.LeakyReLU(negative_slope=0.2)))

    def forward(self, input, offset=None):
        for block in self.layers:
            input = block(input)
        return input


class Decoder(nn.Module):
    """trajectory_pred_model.py:117-174.  TrajectoryModel never builds it (the file carries it over from the VAE);
    kept constructible for API parity."""

    def __init__(self, args, enc: Encoder):
        super().__init__()
        self.layers = nn.ModuleList()
        self.unpools = nn.ModuleList()
        self.args = args
        self.enc = enc
        self.convs = []
        plan = enc.plan
        L = plan.L
        for i in range(L):
            lvl = L - 1 - i
            cin = plan.channel_list[L - i]
            unpool = SkeletonUnpool(plan.pooling[lvl], cin // plan.edge_num[lvl])
            self.unpools.append(unpool)
            conv = SkeletonConv(plan.neigh[lvl], in_channels=cin, out_channels=cin // 2, joint_num=plan.edge_num[lvl],
                                kernel_size=plan.K, stride=1 if i % 2 == 0 else 2, padding=plan.pad,
                                padding_mode=args['padding_mode'], bias=(i == 0 or i == L - 1))
            self.convs.append(conv)
            seq = [nn.Upsample(scale_factor=2, mode=args['upsampling'], align_corners=False), unpool, conv]
            if i != L - 1:
                seq.append(nn.LeakyReLU(negative_slope=0.2))
            self.layers.append(nn.Sequential(*seq))

    def forward(self, input, offset=None):
        for block in self.layers:
            input = block(input)
        return input


class TrajectoryModel(nn.Module):
    """trajectory_pred_model.py:176-440"""

    def __init__(self, hp):
        super().__init__()
        self.latent_d = hp['latent_d']
        self.n_joints = hp['n_joints']
        self.input_dim = hp['input_dim']
        self.output_dim = hp['output_dim']
        self.max_timesteps = hp['train_seq_len']
        self.hp = hp
        edges = get_edges("./utils/data/joint24_parents.json")
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.fk_layer = ForwardKinematicsLayer(device=dev)
        self.enc = Encoder(hp, edges)
        self.d_model = self.enc.channel_base[-1]
        self.fc_mapping = nn.Linear(self.d_model * 7, 3)
        ms = np.load(topology.resolve_asset("./utils/data/for_all_data_motion_model/all_amass_data_mean_std.npy"))
        ms[1, ms[1, :] == 0] = 1.0
        self.mean_vals = torch.from_numpy(ms[0, :]).float()[None, :].to(dev)
        self.std_vals = torch.from_numpy(ms[1, :]).float()[None, :].to(dev)
        self._engines = {}
        self.last_engine = None

    # -- engine plumbing --
    def engine(self, timesteps=None):
        """The kernel sequence for one sequence length (gather tables are per length; train_seq_len by default)."""
        T = int(self.max_timesteps if timesteps is None else timesteps)
        dev = next(self.parameters()).device
        key = (T, str(dev))
        if key not in self._engines:
            hp = dict(self.hp, train_seq_len=T)
            self._engines[key] = TrajEngine(hp, dict(self.named_parameters()),
                                            self.std_vals[0, ROOT_V_SLICE[0]:ROOT_V_SLICE[1]].cpu().numpy(), dev,
                                            parents=self.fk_layer.parent_list)
        return self._engines[key]

    def _dev(self, t):
        dev = next(self.parameters()).device
        return t.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()

    def _encoder_input(self, seq_rot_6d, seq_joint_pos):
        return seq_joint_pos if self.hp['trajectory_input_joint_pos'] else seq_rot_6d

    def forward(self, data, hp, iterations, multigpus=False, validation_flag=False):
        """Returns the reference's 9-tuple (total, kl, 0, 0, 0, 0, rec_root_v, 0, rec_root_trans); unless
        validation_flag, parameter .grad is populated (backward inside)."""
        x = self._dev(self._encoder_input(data[0], data[3]))
        root_v = self._dev(data[6])
        eng = self.last_engine = self.engine(x.shape[1])
        ls = eng.train_step(x, root_v, backward=not validation_flag)
        self.last_losses = ls
        return tuple(ls[i] for i in range(9))

    def l2_criterion(self, pred, gt):
        assert pred.size() == gt.size()
        return ((pred - gt) ** 2).mean()

    def l2_criterion_for_save(self, pred, gt):
        assert pred.size() == gt.size()
        loss = (pred - gt) ** 2
        bs, timesteps = loss.shape[0], loss.shape[1]
        return loss.mean(), loss.reshape(bs, -1).sum(dim=1).cpu().numpy(), loss.reshape(bs, timesteps, -1).sum(dim=-1)

    def de_standardize(self, output_data, start_idx, end_idx):
        m, s = self.mean_vals[:, start_idx:end_idx], self.std_vals[:, start_idx:end_idx]
        if output_data.dim() == 3:
            m, s = m[None], s[None]
        return m + s * output_data

    def standardize_data_specify_dim(self, ori_data, start_idx, end_idx):
        return (ori_data - self.mean_vals[:, start_idx:end_idx]) / self.std_vals[:, start_idx:end_idx]

    def gen_motion_w_trajectory(self, pose_data, root_v_data, need_destandardize=True):
        """pose (T,bs,24,3) root-relative, root_v (T,bs,3): frame t >= 1 is moved by the sum of the velocities of
        frames 1..t (:292-305)."""
        v = root_v_data
        if need_destandardize:
            v = self.de_standardize(v, *ROOT_V_SLICE)
        v = v.clone()
        v[0] = 0.0
        return pose_data + torch.cumsum(v, dim=0)[:, :, None, :]

    def adjust_root_rot(self, ori_seq_data):
        """ori (bs,T,24,3,3): re-express the root so that frame 0 has identity root rotation (:307-329)."""
        bs, T = ori_seq_data.shape[0], ori_seq_data.shape[1]
        rel = ori_seq_data[:, 0, 0].transpose(1, 2)[:, None].expand(bs, T, 3, 3).contiguous()
        out = ori_seq_data.clone()
        out[:, :, 0] = torch.matmul(rel, ori_seq_data[:, :, 0])
        return out, rel

    def apply_root_rot_to_translation(self, relative_root, ori_root_v):
        return torch.matmul(relative_root, ori_root_v[:, :, :, None]).squeeze(-1)

    def predict_root_v(self, encoder_input):
        """(bs,T,24*c) or (bs,T,24,c) normalised encoder input -> (bs,T,3) normalised root velocity."""
        x = self._dev(encoder_input)
        x = x.reshape(x.shape[0], x.shape[1], -1)
        with torch.no_grad():
            return self.engine(x.shape[1]).predict(x)

    def test(self, data, hp, iterations, gen_seq_len=None):
        """:339-395 — ground-truth and predicted absolute joint trajectories, each (T,bs,24,3)."""
        with torch.no_grad():
            rot_pos, root_v = self._dev(data[2]), self._dev(data[6])
            x = self._dev(self._encoder_input(data[0], data[3]))
            bs, T = x.shape[0], x.shape[1]
            pose = rot_pos.transpose(0, 1).contiguous().reshape(T, bs, 24, 3)
            gt = self.gen_motion_w_trajectory(pose, root_v.transpose(0, 1))
            pred = self.gen_motion_w_trajectory(pose, self.predict_root_v(x).transpose(0, 1))
        return gt, pred, pred, None

    def sampled_seq_test(self, data_input):
        """:397-439 — data_input (bs,T,24,6) rotations or (bs,T,24,3) root-relative joint positions."""
        bs, T = data_input.shape[0], data_input.shape[1]
        data_input = self._dev(data_input)
        with torch.no_grad():
            if data_input.shape[-1] == 6:
                pose = self.fk_layer(data_input.reshape(bs * T, -1, 6)).reshape(bs, T, 24, 3)
            else:
                pose = data_input
            if self.hp['trajectory_input_joint_pos']:
                x = self.standardize_data_specify_dim(pose.reshape(-1, 72), *JOINT_POS_SLICE).reshape(bs, T, 72)
            else:
                x = data_input
            out_v = self.predict_root_v(x).transpose(0, 1)
            return self.gen_motion_w_trajectory(pose.transpose(0, 1), out_v)

    def gen_seq(self, data, hp, iterations):
        return self.test(data, hp, iterations, hp['max_input_timesteps'])
