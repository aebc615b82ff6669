"""HM-VAE model classes with the reference's surface (seq_two_hier_sa_vae.py:53-475, 560-642, 773-796).

`Encoder` / `Decoder` keep the reference module tree (so the 74 state_dict keys, incl. the `dec.enc.*` aliases and
the twice-registered unpools, come out identical), and can be called on their own (differentiable, module by
module).  `TwoHierSAVAEModel.forward` does not walk that tree: it hands the batch to the fused HIP engine
(engine.py), which runs forward, losses and the whole backward as one fixed kernel sequence and leaves the
gradients in `.grad` — the same contract as the reference, whose forward calls `l_total.backward()` itself (:414).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from . import my_tools, topology
from .engine import VaeEngine, VaePlan
from .fk_layer import ForwardKinematicsLayer
from .skeleton import SkeletonUnpool, SkeletonPool, SkeletonConv, find_neighbor, SkeletonLinear, get_edges  # noqa: F401


def get_scheduler(optimizer, hp, it=-1):
    """seq_two_hier_sa_vae.py:27-38 / trainer_motion_vae.py:251-262"""
    policy = hp.get('lr_policy', 'constant')
    if policy == 'constant':
        return None
    if policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=hp['step_size'], gamma=hp['gamma'], last_epoch=it)
    if policy == 'mstep':
        return lr_scheduler.MultiStepLR(optimizer, milestones=hp['step_size'], gamma=hp['gamma'], last_epoch=it)
    return NotImplementedError('%s not implemented', policy)


def get_opt_scheduler(optimizer, hp, it=-1):
    """seq_two_hier_sa_vae.py:40-51 (test-time optimisation schedule; kept for API completeness)."""
    sub = {'lr_policy': hp.get('opt_lr_policy', 'constant'), 'step_size': hp.get('opt_step_size'),
           'gamma': hp.get('opt_gamma')}
    return get_scheduler(optimizer, sub, it)


class Encoder(nn.Module):
    """seq_two_hier_sa_vae.py:53-166"""

    def __init__(self, args, topology_edges):
        super().__init__()
        plan = VaePlan(args, _parents_of(topology_edges))
        self.plan = plan
        self.args = args
        self.topologies = plan.topo
        self.latent_d, self.shallow_latent_d = args['latent_d'], args['shallow_latent_d']
        self.channel_base = plan.ch_base
        self.timestep_list = plan.tsteps
        self.channel_list = plan.channel_list
        self.edge_num = plan.edge_num[:-1] + [len(plan.topo[-1])]      # reference quirk: last entry is len([]) = 0
        self.pooling_list = plan.pooling
        self.layers = nn.ModuleList()
        self.latent_enc_layers = nn.ModuleList()
        self.convs = []
        if args['extra_conv'] != 0:
            raise NotImplementedError("extra_conv > 0 is not used by any shipped config")
        for i in range(plan.L):
            conv = SkeletonConv(plan.neigh[i], in_channels=plan.enc_cin[i], out_channels=plan.enc_cout[i],
                                joint_num=plan.edge_num[i], kernel_size=plan.K, stride=plan.strides[i], padding=plan.pad,
                                padding_mode=args['padding_mode'], bias=True)
            pool = SkeletonPool(edges=plan.topo[i], pooling_mode=args['skeleton_pool'],
                                channels_per_edge=plan.enc_cout[i] // plan.edge_num[i], last_pool=(i == plan.L - 1))
            self.convs.append(conv)
            self.layers.append(nn.Sequential(conv, pool, nn.LeakyReLU(negative_slope=0.2)))
            self.latent_enc_layers.append(nn.Linear(plan.feat[i], 2 * plan.lat_dim[i]))

    def forward(self, input, offset=None):
        """input (B, 24*6, T) on a HIP device -> (latent, [z_dist per level (B, k, 2d)])"""
        zs = []
        x = input
        for lvl, block in enumerate(self.layers):
            x = block(x)
            k = x.shape[1] // self.channel_base[lvl + 1]
            zs.append(self.latent_enc_layers[lvl](x.reshape(x.shape[0], k, -1)))
        return x, zs


def _parents_of(edges):
    """edges as produced by get_edges -> parent list (the virtual root edge is edges[0])."""
    n = len(edges)
    parents = [-1] * n
    for a, b in edges[1:]:
        parents[b] = a
    return parents


class Decoder(nn.Module):
    """seq_two_hier_sa_vae.py:169-294"""

    def __init__(self, args, enc: Encoder):
        super().__init__()
        plan = enc.plan
        self.plan = plan
        self.args = self.hp = args
        self.latent_d, self.shallow_latent_d = args['latent_d'], args['shallow_latent_d']
        self.timestep_list = plan.dec_ts
        # registration order matters for state_dict key order: layers, unpools, latent_dec_layers, then the
        # encoder alias (seq_two_hier_sa_vae.py:172-180)
        self.layers = nn.ModuleList()
        self.unpools = nn.ModuleList()
        self.latent_dec_layers = nn.ModuleList()
        self.enc = enc
        self.convs = []
        L = plan.L
        for i in range(L):
            lvl = L - 1 - i
            self.latent_dec_layers.append(nn.Linear(plan.lat_dim[lvl], plan.dec_feat[i]))
            unpool = SkeletonUnpool(plan.pooling[lvl], plan.dec_cin[i] // plan.edge_num[lvl])
            self.unpools.append(unpool)
            seq = []
            if plan.dec_up[i]:
                seq.append(nn.Upsample(scale_factor=2, mode=args['upsampling'], align_corners=False))
            seq.append(unpool)
            conv = SkeletonConv(plan.neigh[lvl], in_channels=plan.dec_cin[i], out_channels=plan.dec_cout[i],
                                joint_num=plan.edge_num[lvl], kernel_size=plan.K, stride=1, padding=plan.pad,
                                padding_mode=args['padding_mode'], bias=plan.dec_bias[i])
            seq.append(conv)
            self.convs.append(conv)
            if i != L - 1:
                seq.append(nn.LeakyReLU(negative_slope=0.2))
            self.layers.append(nn.Sequential(*seq))

    def forward(self, z_vec_list, offset=None):
        """[z per level, shallow..deep, each (B, k, d)] -> (B, 24*6, T)"""
        L = self.plan.L
        # only the deepest and the shallowest latent feed the network; the two middle heads are computed by the
        # reference (:264-272) and never consumed
        deep = self.latent_dec_layers[0](z_vec_list[L - 1])
        x = deep.reshape(deep.shape[0], -1, self.timestep_list[0])
        for i, block in enumerate(self.layers):
            if i == L - 1:
                skip = self.latent_dec_layers[L - 1](z_vec_list[0])
                b, _, t = x.shape
                k = self.plan.edge_num[1]
                x = torch.cat((x.reshape(b, k, -1, t), skip.reshape(b, k, -1, t)), dim=2).reshape(b, -1, t)
            x = block(x)
        return x


class TwoHierSAVAEModel(nn.Module):
    """seq_two_hier_sa_vae.py:296-475 (+ test/gen_seq/rot_mat_to_6d/sample_single_seq)"""

    def __init__(self, hp):
        super().__init__()
        self.latent_d = hp['latent_d']
        self.shallow_latent_d = hp['shallow_latent_d']
        self.n_joints = hp['n_joints']
        self.input_dim = hp['input_dim']
        self.output_dim = hp['output_dim']
        self.max_timesteps = hp['train_seq_len']
        self.hp = hp
        edges = get_edges("./utils/data/joint24_parents.json")
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        self.fk_layer = ForwardKinematicsLayer(device=dev)
        self.enc = Encoder(hp, edges)
        self.dec = Decoder(hp, self.enc)
        self.iteration_interval = hp['iteration_interval']
        ms = np.load(topology.resolve_asset("./utils/data/for_all_data_motion_model/all_amass_data_mean_std.npy"))
        ms[1, ms[1, :] == 0] = 1.0
        self.mean_vals = torch.from_numpy(ms[0, :]).float()[None, :].to(dev)
        self.std_vals = torch.from_numpy(ms[1, :]).float()[None, :].to(dev)
        self._engine = None
        self.eps_override = None          # parity hook: list of per-level eps tensors (B*k, d); None -> torch.randn

    # -- engine plumbing --
    def engine(self):
        dev = next(self.parameters()).device
        if self._engine is None or self._engine.device != dev:
            params = dict(self.named_parameters())
            self._engine = VaeEngine(self.hp, params, self.fk_layer.positions[0].cpu().numpy(), dev,
                                     parents=self.fk_layer.parent_list)
        return self._engine

    def get_hier_level(self, step):
        return 1 if step < self.iteration_interval else 4

    def _dev(self, t):
        dev = next(self.parameters()).device
        return t.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()

    def forward(self, data, hp, iterations, multigpus=False, validation_flag=False):
        """Returns the reference's 10-tuple; unless validation_flag, parameter .grad is populated (backward inside)."""
        seq_rot_6d, seq_rot_mat = self._dev(data[0]), self._dev(data[1])
        eng = self.engine()
        eps = None
        if self.eps_override is not None:
            eps = [self._dev(e) for e in self.eps_override]
        ls = eng.train_step(seq_rot_6d, seq_rot_mat, iterations, eps_list=eps, backward=not validation_flag)
        self.last_losses = ls
        zero = lambda: ls[7:8]          # a device zero without a fill kernel
        l_kl_list = [ls[5]] + [zero() for _ in range(self.hp['num_layers'] - 2)] + [ls[6]]
        return ls[0], ls[1], ls[2], ls[3], ls[4], zero(), zero(), zero(), zero(), l_kl_list

    def reparametrize(self, pred_mean, pred_logvar):
        return torch.randn_like(pred_mean) * torch.exp(0.5 * pred_logvar) + pred_mean

    def kl_loss(self, logvar, mu):
        return (-0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1)).mean()

    def l2_criterion(self, pred, gt):
        assert pred.size() == gt.size()
        return ((pred - gt) ** 2).mean()

    def _decode(self, z_list, adjust_root_rot_flag=False, relative_root_rot=None, need_c6=True):
        """z_list: per-level latents (shallow..deep).  Returns the reference 7-tuple (last four are None).
        need_c6=False (not in the reference): the first element is None and the kernel skips writing it."""
        eng = self.engine()
        z0, zd = self._dev(z_list[0]), self._dev(z_list[-1])
        with torch.no_grad():
            c6, R, pos = eng.decode(z0, zd, need_c6=need_c6)
            if adjust_root_rot_flag:       # visualisation-only branch of the reference (:457-465)
                bs, T = R.shape[0], R.shape[1]
                if relative_root_rot is not None:
                    R = R.clone()
                    R[:, :, 0] = torch.matmul(relative_root_rot, R[:, :, 0])
                else:
                    R, relative_root_rot = self.adjust_root_rot(R)
                pos = self.fk_layer(R.reshape(bs * T, self.n_joints, 3, 3)).reshape(bs, T, self.n_joints, 3)
        return c6, R, pos, None, None, None, None

    def de_standardize(self, output_data, start_idx, end_idx):
        m, s = self.mean_vals[:, start_idx:end_idx], self.std_vals[:, start_idx:end_idx]
        if output_data.dim() == 3:
            m, s = m[None], s[None]
        return m + s * output_data

    def adjust_root_rot(self, ori_seq_data):
        """ori (bs,T,24,3,3): re-express the root so that frame 0 has identity root rotation (:532-551)."""
        bs, T = ori_seq_data.shape[0], ori_seq_data.shape[1]
        rel = ori_seq_data[:, 0, 0].transpose(1, 2)[:, None].expand(bs, T, 3, 3).contiguous()
        out = ori_seq_data.clone()
        out[:, :, 0] = torch.matmul(rel, ori_seq_data[:, :, 0])
        return out, rel

    def test(self, data, hp, iterations, gen_seq_len=None):
        """seq_two_hier_sa_vae.py:560-636: ground truth, mean-latent and sampled-latent joint positions, (T,B,24,3)."""
        with torch.no_grad():
            r6, rmat = self._dev(data[0]), self._dev(data[1])
            bs, T, _ = r6.shape
            rel = None
            if hp['random_root_rot_flag']:
                rm, rel = self.adjust_root_rot(rmat.reshape(bs, T, 24, 3, 3))
                rmat = rm.reshape(bs, T, -1)
            gt = self.fk_layer(rmat.reshape(bs * T, self.n_joints, 3, 3)).reshape(bs, T, self.n_joints, 3)
            gt_seq_res = gt.transpose(0, 1).contiguous()
            _, zs = self.engine().encode_batch(r6)
            mean_z, sampled_z = [], []
            for lvl, dist in enumerate(zs):
                d = self.shallow_latent_d if lvl == 0 else self.latent_d
                mean_z.append(dist[:, :, :d].contiguous())
                sampled_z.append(torch.randn_like(mean_z[-1]))
            if hp['random_root_rot_flag']:
                mean_pos = self._decode(mean_z, adjust_root_rot_flag=True, relative_root_rot=rel, need_c6=False)[2]
                samp_pos = self._decode(sampled_z, adjust_root_rot_flag=True, need_c6=False)[2]
            else:
                mean_pos = self._decode(mean_z, need_c6=False)[2]
                samp_pos = self._decode(sampled_z, need_c6=False)[2]
        return gt_seq_res, mean_pos.transpose(0, 1), samp_pos.transpose(0, 1), None

    def gen_seq(self, data, hp, iterations):
        return self.test(data, hp, iterations, hp['max_input_timesteps'])

    def l2_masked_criterion(self, pred, gt, mask):
        """seq_two_hier_sa_vae.py:717-735 on (B,T,24,...) tensors: (mean of the masked squared error, its (B,T) map).
        Reporting helper on torch tensors; the optimisation loop itself uses the fused masked decode-tail kernel."""
        assert pred.size() == gt.size()
        m = mask.reshape(mask.shape + (1,) * (pred.dim() - 3))
        loss = (pred - gt) ** 2 * m
        return loss.mean(), loss.reshape(loss.shape[0], loss.shape[1], -1).mean(dim=-1)

    def refine_window(self, target_rot_mat, mask, hp=None, z_shallow=None, z_deep=None, iterations=None,
                      trajectory_trainer=None, root_trans=None):
        """The test-time optimisation loop of the interpolation / completion routines (seq_two_hier_sa_vae.py:1356-1429) for
        a batch of windows: target_rot_mat (B,T,24,3,3), mask (B,T,24).  Returns (cont6d, R, pos, loss history (it, 8));
        the decoder is restored afterwards, as the reference optimises a deep copy.  With hp['optimize_trajectory']:
        trajectory_trainer (the trained TrajectoryModel or its Trainer, :1378) and root_trans (B,T,3), :1381."""
        from .refine import LatentRefiner
        hp = self.hp if hp is None else hp
        # one refiner (buffers, optimisers, two captured graphs) per window shape and set of hyper-parameters, reused
        # window after window: the graphs bake the scalars in, so the key holds every scalar of hp
        key = (int(target_rot_mat.shape[0]), int(target_rot_mat.shape[1]),
               tuple(sorted((k, v) for k, v in hp.items() if isinstance(v, (int, float, bool, str)))))
        cache = self.__dict__.setdefault("_refiners", {})
        ref = cache.get(key)
        if ref is None or ref.eng is not self.engine():
            ref = cache[key] = LatentRefiner(self, hp, target_rot_mat, mask, z_shallow, z_deep,
                                             trajectory_model=trajectory_trainer, root_trans=root_trans)
        else:
            ref.begin(target_rot_mat, mask, z_shallow, z_deep, root_trans=root_trans)
        try:
            hist = ref.run(iterations)
            c6, R, pos = ref.decode()
        finally:
            ref.restore()
        return c6, R, pos, hist

    def rot_mat_to_6d(self, rotMatrices):
        """(bs,24,3,3) -> (bs,24,6) = [col0 | col1] (:773-779)"""
        c = torch.stack((rotMatrices[:, :, :, 0], rotMatrices[:, :, :, 1]), dim=-2)
        return c.reshape(rotMatrices.shape[0], rotMatrices.shape[1], 6)

    def sample_single_seq(self, bs=256):
        """Decoder-only sampling (:781-796).  The reference passes a bare tensor where `_decode` needs the 4-list and
        cannot run as released; this is the working equivalent (the sampled branch of `test`)."""
        plan = self.enc.plan
        dev = next(self.parameters()).device
        z_list = [torch.randn(bs, plan.edge_num[i + 1], plan.lat_dim[i], device=dev) for i in range(plan.L)]
        c6 = self._decode(z_list)[0]
        clean = my_tools.rotation_matrix_from_ortho6d(c6.reshape(-1, 24, 6))
        return self.rot_mat_to_6d(clean).reshape(bs, -1, 24, 6)
