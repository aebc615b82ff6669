"""CPU oracle for the hm-vae training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) restatement of the reference algorithm for the path
named in BASELINE.json: skeleton topology, skeleton-aware conv / pool / unpool,
rot-6D lift, forward kinematics, the TwoHierSAVAE encoder/decoder, losses and one Adam step.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
the product package (`hm_vae_amd/`) never does and fails loudly without its HIP library.

Parity pin: every function below is checked in `tests/test_oracle_golden.py` against fixtures in
`tests/golden/` that were produced by running the UNMODIFIED reference on CPU in the build
container (`tools/gen_golden.py`), plus the reference's own known-answer data
(`rest_pose_coord.npy`, the topology dump in `skeleton.py:464-477`).

Each function cites the reference lines it follows (paths relative to the reference root).
Third-party arithmetic (conv1d, reflect pad, linear interpolation, Adam) is PyTorch's, exactly as
in the reference (SURVEY.md §8c last row).
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

SMPL24_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]


# --------------------------------------------------------------------------------------
# topology (skeleton.py:306-315, 364-410, 159-231)
# --------------------------------------------------------------------------------------
def skeleton_edges(parents):
    """skeleton.py:306-315 — virtual root edge (0, n_joints) first, then (parent, child)."""
    n = len(parents)
    return [(0, n)] + [(parents[j], j) for j in range(1, n)]


def edge_distances(edges):
    """skeleton.py:364-387 — edges are adjacent when they share a joint id; all-pairs hops."""
    m = len(edges)
    big = 100000
    d = np.full((m, m), big, dtype=np.int64)
    for a in range(m):
        for b in range(m):
            if a == b:
                d[a, b] = 0
            elif set(edges[a]) & set(edges[b]):
                d[a, b] = 1
    for k in range(m):
        d = np.minimum(d, d[:, k:k + 1] + d[k:k + 1, :])
    return d


def edge_neighbours(edges, dist):
    """skeleton.py:390-410 — per edge, sorted ids of edges within `dist` hops."""
    d = edge_distances(edges)
    return [[int(j) for j in np.nonzero(d[i] <= dist)[0]] for i in range(len(edges))]


def pool_plan(edges, last_pool=False):
    """skeleton.py:173-207 — chains from joint 0, merged pairwise (odd head kept alone).

    Returns (seq_list, pooling_list, new_edges).
    """
    degree = {}
    for a, b in edges:
        degree[a] = degree.get(a, 0) + 1
        degree[b] = degree.get(b, 0) + 1
    chains = []

    def walk(joint, chain):
        if degree[joint] > 2 and joint != 0:
            chains.append(chain)
            chain = []
        if degree[joint] == 1:
            chains.append(chain)
            return
        for idx, e in enumerate(edges):
            if e[0] == joint:
                walk(e[1], chain + [idx])

    walk(0, [])
    pooling, coarse = [], []
    for chain in chains:
        if last_pool:
            pooling.append(list(chain))
            continue
        rest = list(chain)
        if len(rest) % 2 == 1:
            pooling.append([rest[0]])
            coarse.append(edges[rest[0]])
            rest = rest[1:]
        for i in range(0, len(rest), 2):
            pooling.append([rest[i], rest[i + 1]])
            coarse.append([edges[rest[i]][0], edges[rest[i + 1]][1]])
    return chains, pooling, coarse


def pool_matrix(pooling, n_in_edges, ch):
    """skeleton.py:219-224 — mean over the merged edges, channel-wise."""
    w = torch.zeros(len(pooling) * ch, n_in_edges * ch)
    for i, grp in enumerate(pooling):
        for j in grp:
            for c in range(ch):
                w[i * ch + c, j * ch + c] = 1.0 / len(grp)
    return w


def unpool_matrix(pooling, ch):
    """skeleton.py:248-253 — every fine edge copies its coarse edge."""
    n_out = sum(len(g) for g in pooling)
    w = torch.zeros(n_out * ch, len(pooling) * ch)
    for i, grp in enumerate(pooling):
        for j in grp:
            for c in range(ch):
                w[j * ch + c, i * ch + c] = 1.0
    return w


def conv_mask(neigh, cin, cout, ksize):
    """skeleton.py:34-39, 58-60 — 1 where the input channel belongs to a neighbour edge."""
    n = len(neigh)
    ci, co = cin // n, cout // n
    m = torch.zeros(cout, cin, ksize)
    for i, nb in enumerate(neigh):
        cols = [k * ci + c for k in nb for c in range(ci)]
        m[co * i:co * (i + 1), cols, :] = 1
    return m


# --------------------------------------------------------------------------------------
# rotation lift + forward kinematics (my_tools.py:6-39, fk_layer.py:47-93)
# --------------------------------------------------------------------------------------
def _unit(v):
    """my_tools.py:6-8 — v / max(||v||_2, 1e-6)."""
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-6)


def _cross(u, v):
    """my_tools.py:10-17 — component formula with separately rounded products (no fma), which is
    what decides the result for (near-)parallel inputs."""
    return torch.stack((u[..., 1] * v[..., 2] - u[..., 2] * v[..., 1],
                        u[..., 2] * v[..., 0] - u[..., 0] * v[..., 2],
                        u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]), dim=-1)


def rot6d_to_matrix(p):
    """my_tools.py:19-39 — x=n(a), z=n(x×b), y=z×x; columns of R are x,y,z."""
    a, b = p[..., 0:3], p[..., 3:6]
    x = _unit(a)
    z = _unit(_cross(x, b))
    y = _cross(z, x)
    return torch.stack((x, y, z), dim=-1)


def matrix_to_rot6d(r):
    """seq_two_hier_sa_vae.py:773-779 — first two columns, [col0 ‖ col1]."""
    return torch.cat((r[..., :, 0], r[..., :, 1]), dim=-1)


def fk_positions(rot, parents, offsets):
    """fk_layer.py:63-93 — p_0 = off_0, G_0 = R_0; G_i = G_par·R_i, p_i = G_par·off_i + p_par.

    rot: (N,J,3,3) or (N,J,6); offsets: (J,3) tensor.  Returns (N,J,3).
    """
    if rot.shape[-1] == 6:
        rot = rot6d_to_matrix(rot)
    n, j = rot.shape[0], rot.shape[1]
    off = offsets.to(rot.dtype)
    glob = [None] * j
    pos = [None] * j
    glob[0] = rot[:, 0]
    pos[0] = off[0].expand(n, 3)
    for i in range(1, j):
        pa = parents[i]
        glob[i] = glob[pa] @ rot[:, i]
        pos[i] = (glob[pa] @ off[i]) + pos[pa]
    return torch.stack(pos, dim=1)


# --------------------------------------------------------------------------------------
# layers (skeleton.py:95-105, 228-231, 258-261)
# --------------------------------------------------------------------------------------
def skel_conv(x, weight, mask, bias, stride, pad, pad_mode="reflect"):
    """skeleton.py:95-105."""
    xp = F.pad(x, (pad, pad), mode=pad_mode) if pad > 0 else x
    return F.conv1d(xp, weight * mask, bias, stride)


def _pad_mode(name):
    return {"zeros": "constant", "reflection": "reflect"}.get(name, name)


def level_strides(T, L):
    """seq_two_hier_sa_vae.py:79-90, 107-118 — per-level conv stride (and timestep ladder)."""
    s = []
    for i in range(L):
        if T == 8:
            s.append(1 if (i == 0 or i == L - 1) else 2)
        elif T == 16:
            s.append(1 if i == 0 else 2)
        else:
            s.append(2)
    ts = [T]
    for st in s:
        ts.append(ts[-1] // st)
    return s, ts


class Plan:
    """Static structure of the TwoHierSAVAE for one config (seq_two_hier_sa_vae.py:53-258)."""

    def __init__(self, hp, parents=SMPL24_PARENTS):
        L = hp["num_layers"]
        T = hp["train_seq_len"]
        self.hp, self.L, self.T = hp, L, T
        self.K = hp["kernel_size"]
        self.pad = (self.K - 1) // 2
        self.pad_mode = _pad_mode(hp["padding_mode"])
        self.strides, self.tsteps = level_strides(T, L)
        self.ch_base = [6 * (2 ** i) for i in range(L + 1)]
        self.topo = [skeleton_edges(parents)]
        self.neigh, self.pooling, self.edge_num = [], [], [len(self.topo[0])]
        self.enc_cin, self.enc_cout = [], []
        for i in range(L):
            nb = edge_neighbours(self.topo[i], hp["skeleton_dist"])
            self.neigh.append(nb)
            self.enc_cin.append(self.ch_base[i] * self.edge_num[i])
            self.enc_cout.append(self.ch_base[i + 1] * self.edge_num[i])
            _, pl, ne = pool_plan(self.topo[i], last_pool=(i == L - 1))
            self.pooling.append(pl)
            self.topo.append(ne)
            self.edge_num.append(len(ne) if i < L - 1 else len(pl))
        # NB: on the last level new_edges is empty (last_pool) — edge_num follows len(pooling).
        self.edge_num[L] = len(self.pooling[L - 1])
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
        d, sd = hp["latent_d"], hp["shallow_latent_d"]
        self.lat_dim = [sd] + [d] * (L - 1)            # encoder level i -> latent width

    def dec_conv_key(self, i):
        """Sequential index of the conv inside dec.layers[i]: [Upsample?] + Unpool + Conv (:233-251)."""
        return "dec.layers.%d.%d" % (i, 2 if self.dec_up[i] else 1)

    def dec_unpool_key(self, i):
        return "dec.layers.%d.%d" % (i, 1 if self.dec_up[i] else 0)


def reference_param_shapes(plan):
    """Names/shapes of the reference state_dict tensors that carry values (no aliases, no constants)."""
    L, K = plan.L, plan.K
    out = {}
    for i in range(L):
        out["enc.layers.%d.0.weight" % i] = (plan.enc_cout[i], plan.enc_cin[i], K)
        out["enc.layers.%d.0.bias" % i] = (plan.enc_cout[i],)
        feat = plan.ch_base[i + 1] * plan.tsteps[i + 1]
        out["enc.latent_enc_layers.%d.weight" % i] = (2 * plan.lat_dim[i], feat)
        out["enc.latent_enc_layers.%d.bias" % i] = (2 * plan.lat_dim[i],)
    for i in range(L):
        out[plan.dec_conv_key(i) + ".weight"] = (plan.dec_cout[i], plan.dec_cin[i], K)
        if plan.dec_bias[i]:
            out[plan.dec_conv_key(i) + ".bias"] = (plan.dec_cout[i],)
        lat = plan.lat_dim[L - 1 - i]
        feat = plan.ch_base[L - i] * plan.dec_ts[i]
        out["dec.latent_dec_layers.%d.weight" % i] = (feat, lat)
        out["dec.latent_dec_layers.%d.bias" % i] = (feat,)
    return out


def name_seeded_params(plan):
    """Deterministic weights keyed by reference state_dict name (fixtures are too big to commit).

    value = N(0,1) from numpy default_rng(crc32(name)), scaled ~kaiming; conv weights are multiplied
    by the neighbour mask exactly like a trained reference checkpoint (masked entries stay 0).
    """
    params = {}
    masks = oracle_masks(plan)
    for name, shape in reference_param_shapes(plan).items():
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        v = rng.standard_normal(shape).astype(np.float32)
        if name.endswith("bias"):
            v *= 0.05
        elif len(shape) == 3:
            m = masks[name.replace("weight", "mask")].numpy()
            fan = m[:, :, 0].sum(axis=1, keepdims=True)[:, :, None] * shape[2]
            v = v * m * np.sqrt(2.0 / np.maximum(fan, 1.0)).astype(np.float32)
        else:
            v *= np.float32(math.sqrt(2.0 / shape[1]))
        params[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return params


def oracle_masks(plan):
    L, K = plan.L, plan.K
    out = {}
    for i in range(L):
        out["enc.layers.%d.0.mask" % i] = conv_mask(plan.neigh[i], plan.enc_cin[i], plan.enc_cout[i], K)
        out[plan.dec_conv_key(i) + ".mask"] = conv_mask(plan.neigh[L - 1 - i], plan.dec_cin[i], plan.dec_cout[i], K)
    return out


# --------------------------------------------------------------------------------------
# model
# --------------------------------------------------------------------------------------
class OracleVAE:
    """Functional restatement of TwoHierSAVAEModel (seq_two_hier_sa_vae.py:296-474).

    `params` maps reference state_dict names to leaf tensors (requires_grad as needed).
    """

    def __init__(self, hp, params=None, parents=SMPL24_PARENTS, offsets=None, dtype=torch.float32):
        """dtype: float32 is the reference's arithmetic (what the golden vectors pin); float64 evaluates the SAME graph
        in double precision -- the yardstick for gradients whose fp32 evaluation error (the reference's own included)
        is comparable to the parity tolerance."""
        self.hp = hp
        self.dtype = dtype
        self.plan = Plan(hp, parents)
        self.parents = list(parents)
        if offsets is None:
            raise ValueError("offsets (J,3) required")
        self.offsets = torch.as_tensor(offsets, dtype=torch.float32).to(dtype)
        self.masks = {k: v.to(dtype) for k, v in oracle_masks(self.plan).items()}
        p = self.plan
        self.pool_w = [pool_matrix(p.pooling[i], p.edge_num[i], p.enc_cout[i] // p.edge_num[i]).to(dtype)
                       for i in range(p.L)]
        self.unpool_w = [unpool_matrix(p.pooling[p.L - 1 - i], p.dec_cin[i] // p.edge_num[p.L - 1 - i]).to(dtype)
                         for i in range(p.L)]
        self.params = {}
        self.branch, self.band, self.n_ambiguous = None, 1e-5, 0          # see _leaky
        if params is not None:
            self.load(params)

    def load(self, params):
        self.params = {k: v.detach().clone().to(self.dtype).requires_grad_(True) for k, v in params.items()
                       if k in reference_param_shapes(self.plan)}

    def trainable(self):
        """Tensors in the reference's `.parameters()` order restricted to requires_grad."""
        order = []
        p = self.plan
        for i in range(p.L):
            order += ["enc.layers.%d.0.weight" % i, "enc.layers.%d.0.bias" % i]
        for i in range(p.L):
            order += ["enc.latent_enc_layers.%d.weight" % i, "enc.latent_enc_layers.%d.bias" % i]
        for i in range(p.L):
            order += [p.dec_conv_key(i) + ".weight"]
            if p.dec_bias[i]:
                order += [p.dec_conv_key(i) + ".bias"]
        for i in range(p.L):
            order += ["dec.latent_dec_layers.%d.weight" % i, "dec.latent_dec_layers.%d.bias" % i]
        return [(n, self.params[n]) for n in order]

    # -- encoder (seq_two_hier_sa_vae.py:142-166) --
    def _leaky(self, pre, key):
        """LeakyReLU(0.2).  With self.branch set (checker aid for big batches, as in OracleTrajectory.encode): the
        derivative jumps at 0, so for an input within fp32 rounding of 0 (|x| < self.band) both sides are valid fp32
        evaluations; there -- and only there -- the oracle follows branch[key], a bool (B,C,T) tensor "the implementation
        under test took the positive side".  self.n_ambiguous counts those elements."""
        x = F.leaky_relu(pre, 0.2)
        br = None if self.branch is None else self.branch.get(key)
        if br is not None:
            amb = pre.detach().abs() < self.band
            self.n_ambiguous += int(amb.sum())
            x = torch.where(amb, torch.where(br, pre, 0.2 * pre), x)
        return x

    def encode(self, x):
        p, P = self.plan, self.params
        zs, acts = [], []
        for i in range(p.L):
            x = skel_conv(x, P["enc.layers.%d.0.weight" % i], self.masks["enc.layers.%d.0.mask" % i],
                          P["enc.layers.%d.0.bias" % i], p.strides[i], p.pad, p.pad_mode)
            x = torch.matmul(self.pool_w[i], x)
            x = self._leaky(x, ("enc", i))
            acts.append(x)
            b = x.shape[0]
            k_edges = x.shape[1] // p.ch_base[i + 1]
            feat = x.reshape(b, k_edges, -1)
            zs.append(F.linear(feat, P["enc.latent_enc_layers.%d.weight" % i], P["enc.latent_enc_layers.%d.bias" % i]))
        return x, zs, acts

    # -- decoder (seq_two_hier_sa_vae.py:260-294) --
    def decode_feats(self, z_list):
        p, P = self.plan, self.params
        L = p.L
        hier = []
        for zi in range(L):
            z = z_list[L - 1 - zi]
            f = F.linear(z, P["dec.latent_dec_layers.%d.weight" % zi], P["dec.latent_dec_layers.%d.bias" % zi])
            hier.append(f.reshape(z.shape[0], -1, p.dec_ts[zi]))
        x = None
        for i in range(L):
            if i == 0:
                x = hier[0]
            elif i == L - 1:
                b, _, t = x.shape
                k_edges = p.edge_num[L - i]
                x = torch.cat((x.reshape(b, k_edges, -1, t), hier[i].reshape(b, k_edges, -1, t)), dim=2).reshape(b, -1, t)
            if p.dec_up[i]:
                x = F.interpolate(x, scale_factor=2, mode=self.hp["upsampling"], align_corners=False)
            x = torch.matmul(self.unpool_w[i], x)
            ck = p.dec_conv_key(i)
            bias = P.get(ck + ".bias") if p.dec_bias[i] else None
            x = skel_conv(x, P[ck + ".weight"], self.masks[ck + ".mask"], bias, 1,
                          p.pad, p.pad_mode)
            if i != L - 1:
                x = self._leaky(x, ("dec", i))
        return x

    # -- decode tail (seq_two_hier_sa_vae.py:436-474) --
    def decode(self, z_list):
        out = self.decode_feats(z_list)                       # (B, 24*6, T)
        b = out.shape[0]
        T = self.plan.T
        nj = self.hp["n_joints"]
        dec = out.transpose(1, 2).reshape(b * T, nj, -1)
        c6 = dec[:, :, :self.hp["output_dim"]]
        R = rot6d_to_matrix(c6)
        pos = fk_positions(R, self.parents, self.offsets)
        return c6.reshape(b, T, nj, -1), R.reshape(b, T, nj, 3, 3), pos.reshape(b, T, nj, 3), out

    @staticmethod
    def kl(logvar, mu):
        """seq_two_hier_sa_vae.py:425-428."""
        return (-0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1)).mean()

    def train_forward(self, data, iterations, eps_list, backward=True):
        """seq_two_hier_sa_vae.py:335-417 with injected eps (one tensor per level, shape (B*k, d)).

        Returns a dict with every intermediate the parity tests compare.
        """
        hp, p = self.hp, self.plan
        r6, rmat = data[0].to(self.dtype), data[1].to(self.dtype)
        eps_list = [None if e is None else e.to(self.dtype) for e in eps_list]
        b, T, _ = r6.shape
        nj = hp["n_joints"]
        gt_pos = fk_positions(rmat.reshape(b * T, nj, 3, 3), self.parents, self.offsets).detach().reshape(b, T, -1)
        x = r6.transpose(1, 2)
        _, zs, _ = self.encode(x)
        z_list, kls, mus, lvs = [], [], [], []
        for zi, dist in enumerate(zs):
            d = p.lat_dim[zi]
            k_edges = dist.shape[1]
            mu = dist[:, :, :d].reshape(-1, d)
            lv = dist[:, :, d:].reshape(-1, d)
            mus.append(mu)
            lvs.append(lv)
            if hp["kl_w"] != 0:
                z = eps_list[zi] * torch.exp(0.5 * lv) + mu      # :419-423
            else:
                z = mu
            z = z.reshape(b, k_edges, -1)
            if zi == len(zs) - 1:
                kl = self.kl(lv, mu)
            elif zi == 0:
                if iterations < hp["iteration_interval"]:
                    kl = self.kl(lv.detach(), mu.detach())
                    z = z.detach()
                else:
                    kl = self.kl(lv, mu)
            else:
                kl = torch.zeros(1)
            z_list.append(z)
            kls.append(kl)
        c6, R, pos, dec_out = self.decode(z_list)
        if backward:
            dec_out.retain_grad()
        l6 = ((c6.reshape(b, T, -1) - r6) ** 2).mean()
        lr = ((R.reshape(b, T, -1) - rmat) ** 2).mean()
        lp = ((pos.reshape(b, T, -1) - gt_pos) ** 2).mean()
        l_kl = hp["kl_w"] * kls[-1] + hp["shallow_kl_w"] * kls[0]
        total = hp["rec_6d_w"] * l6 + hp["rec_rot_w"] * lr + hp["rec_pose_w"] * lp + hp["kl_w"] * kls[-1] + \
            hp["shallow_kl_w"] * kls[0]
        if backward:
            for t in self.params.values():
                t.grad = None
            total.backward()
        return dict(total=total, kl=l_kl, rec_6d=l6, rec_rot=lr, rec_pose=lp, kl_list=kls, z_list=z_list,
                    z_dist=zs, dec_out=dec_out, c6=c6, R=R, pos=pos, gt_pos=gt_pos)


    # -- test-time optimisation (seq_two_hier_sa_vae.py:1356-1389; masked criterion :717-735) -- SURVEY.md §8 row f-2
    def refine_forward(self, z_shallow, z_deep, target_rmat, mask, anchor=None, backward=True, traj=None,
                       root_trans=None, joint_mean=None, joint_std=None):
        """One iteration's losses of the interpolation / completion loop and their gradients.

        z_shallow (B,k1,d0), z_deep (B,kL,d): leaf tensors (requires_grad) or plain tensors; target_rmat (B,T,24,3,3);
        mask (B,T,24); anchor: name -> original decoder tensor (the `optimize_decoder` regulariser :1369-1373) or None.
        The released routine does not run (stale `_decode` signature), so nothing pins this restatement but the
        reference's text: parity unpinned.
        traj (an OracleTrajectory with the trained trajectory model's weights), root_trans (B,T,3), joint_mean / joint_std
        (72,) = mean_vals / std_vals[360:432]: the `optimize_trajectory` term (:1376-1389) -- sampled_seq_test
        (trajectory_pred_model.py:397-439) on the decoded rotations, then cal_key_frame_root_loss (:1031-1045) restated AS
        RELEASED: the caller passes the (T, 24) mask and the routine selects with `nonzero(mask[0])`, i.e. the JOINTS
        observed in frame 0 serve as frame indices; a batch of windows averages over all windows' terms.
        """
        hp, p = self.hp, self.plan
        b, T = target_rmat.shape[0], target_rmat.shape[1]
        nj = hp["n_joints"]
        z_list = [z_shallow] + [torch.zeros(b, p.edge_num[i + 1], p.lat_dim[i], dtype=self.dtype)
                                for i in range(1, p.L - 1)] + [z_deep]
        c6, R, pos, dec_out = self.decode(z_list)
        gtR = target_rmat.to(self.dtype)
        g6 = torch.stack((gtR[..., 0], gtR[..., 1]), dim=-2).reshape(b, T, nj, 6)          # :1279-1280
        gt_pos = fk_positions(gtR.reshape(b * T, nj, 3, 3), self.parents, self.offsets).reshape(b, T, nj, 3)

        def masked(pred, gt):                                                               # :717-735
            m = mask.to(self.dtype).reshape(tuple(mask.shape) + (1,) * (pred.dim() - 3))
            return (((pred - gt) ** 2) * m).mean()

        l6, lr, lp = masked(c6, g6), masked(R, gtR), masked(pos, gt_pos)
        l_reg = (z_shallow ** 2).mean() + (z_deep ** 2).mean()                              # :1366-1367
        l_reg_dec = torch.zeros(())
        if anchor is not None:
            for n, t0 in anchor.items():
                l_reg_dec = l_reg_dec + ((self.params[n] - t0) ** 2).mean()
        reg_dec_w = hp.get("reg_w_decoder", 0.0) if anchor is not None else 0.0
        total = hp["rec_6d_w"] * l6 + hp["rec_rot_w"] * lr + hp["rec_pose_w"] * lp + hp.get("reg_w", 0.0) * l_reg + \
            reg_dec_w * l_reg_dec
        l_traj = torch.zeros(())
        if traj is not None:
            jm = torch.as_tensor(joint_mean, dtype=self.dtype).reshape(1, 72)
            js = torch.as_tensor(joint_std, dtype=self.dtype).reshape(1, 72)
            x = ((pos.reshape(-1, 72) - jm) / js).reshape(b, T, 72)                          # standardize_data_specify_dim
            v = traj.root_velocity(x)                                                         # (B,T,3) normalised
            pred = traj.trajectory(pos.to(traj.dtype).transpose(0, 1), v.transpose(0, 1))    # (T,B,24,3), :486-499
            pred_root = pred[:, :, 0, :]                                                      # :1379
            terms = []
            for i in range(b):
                idx = torch.nonzero(mask[i][0]).squeeze(-1)                                   # :1036, row 0 of the (T,24) mask
                sp, sg = pred_root[:, i][idx], root_trans[i].to(self.dtype)[idx]
                terms.append(((sp[1:] - sp[:-1]) - (sg[1:] - sg[:-1])) ** 2)
            terms = torch.cat(terms).reshape(-1)
            if terms.numel():
                l_traj = terms.mean()
            total = total + hp.get("reg_w_trajectory", 0.0) * l_traj
        if backward:
            for t in list(self.params.values()) + [z_shallow, z_deep]:
                t.grad = None
            total.backward()
        return dict(total=total, rec_6d=l6, rec_rot=lr, rec_pose=lp, reg=l_reg, reg_decoder=l_reg_dec, trajectory=l_traj,
                    c6=c6, R=R, pos=pos)


# --------------------------------------------------------------------------------------
# TrajectoryModel (trajectory_pred_model.py:45-115, 176-265, 283-305) — SURVEY.md §8 row f-1
# --------------------------------------------------------------------------------------
class TrajPlan:
    """Static structure of TrajectoryModel's encoder (trajectory_pred_model.py:45-104): stride-1 convs, T constant."""

    def __init__(self, hp, parents=SMPL24_PARENTS):
        L = hp["num_layers"]
        self.hp, self.L, self.T = hp, L, hp["train_seq_len"]
        self.K = hp["kernel_size"]
        self.pad = (self.K - 1) // 2
        self.pad_mode = _pad_mode(hp["padding_mode"])
        c0 = 3 if hp["trajectory_input_joint_pos"] else 6                      # :50-53
        self.ch_base = [c0 * (2 ** i) for i in range(L + 1)]
        self.topo = [skeleton_edges(parents)]
        self.neigh, self.pooling, self.edge_num = [], [], [len(self.topo[0])]
        self.enc_cin, self.enc_cout = [], []
        for i in range(L):
            self.neigh.append(edge_neighbours(self.topo[i], hp["skeleton_dist"]))
            self.enc_cin.append(self.ch_base[i] * self.edge_num[i])
            self.enc_cout.append(self.ch_base[i + 1] * self.edge_num[i])
            _, pl, ne = pool_plan(self.topo[i], last_pool=(i == L - 1))
            self.pooling.append(pl)
            self.topo.append(ne)
            self.edge_num.append(len(pl))
        self.d_model = self.ch_base[-1]
        self.k_edges = self.edge_num[L]
        self.fc_in = self.d_model * 7                                          # :192 hard-codes 7 coarse edges


def traj_param_shapes(plan):
    out = {}
    for i in range(plan.L):
        out["enc.layers.%d.0.weight" % i] = (plan.enc_cout[i], plan.enc_cin[i], plan.K)
        out["enc.layers.%d.0.bias" % i] = (plan.enc_cout[i],)
    out["fc_mapping.weight"] = (3, plan.fc_in)
    out["fc_mapping.bias"] = (3,)
    return out


def traj_masks(plan):
    return {"enc.layers.%d.0.mask" % i: conv_mask(plan.neigh[i], plan.enc_cin[i], plan.enc_cout[i], plan.K)
            for i in range(plan.L)}


def traj_name_seeded_params(plan):
    """Same recipe as `name_seeded_params`, keyed "traj:" + name so the two models do not share draws."""
    params, masks = {}, traj_masks(plan)
    for name, shape in traj_param_shapes(plan).items():
        rng = np.random.default_rng(zlib.crc32(("traj:" + name).encode()))
        v = rng.standard_normal(shape).astype(np.float32)
        if name.endswith("bias"):
            v *= 0.05
        elif len(shape) == 3:
            m = masks[name.replace("weight", "mask")].numpy()
            fan = m[:, :, 0].sum(axis=1, keepdims=True)[:, :, None] * shape[2]
            v = v * m * np.sqrt(2.0 / np.maximum(fan, 1.0)).astype(np.float32)
        else:
            v *= np.float32(math.sqrt(2.0 / shape[1]))
        params[name] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return params


class OracleTrajectory:
    """Functional restatement of TrajectoryModel (trajectory_pred_model.py:176-265).

    std_root: the three de-standardisation factors `std_vals[0, 576:579]` (:283-290; the mean cancels in the loss).
    """

    def __init__(self, hp, params, mean_root, std_root, parents=SMPL24_PARENTS, dtype=torch.float32):
        """dtype: float32 = the reference's arithmetic; float64 = the same graph in double (see OracleVAE)."""
        self.hp = hp
        self.dtype = dtype
        self.plan = p = TrajPlan(hp, parents)
        self.masks = {k: v.to(dtype) for k, v in traj_masks(p).items()}
        self.pool_w = [pool_matrix(p.pooling[i], p.edge_num[i], p.enc_cout[i] // p.edge_num[i]).to(dtype)
                       for i in range(p.L)]
        self.mean_root = torch.as_tensor(mean_root, dtype=torch.float32).reshape(1, 1, 3).to(dtype)
        self.std_root = torch.as_tensor(std_root, dtype=torch.float32).reshape(1, 1, 3).to(dtype)
        shapes = traj_param_shapes(p)
        self.params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in params.items() if k in shapes}

    def trainable(self):
        return [(n, self.params[n]) for n in traj_param_shapes(self.plan)]

    def encode(self, x, branch=None, band=1e-5):
        """Encoder.forward (:106-113): conv -> SkeletonPool(mean) -> LeakyReLU(0.2), four times, stride 1.

        branch (checker aid for big batches): per layer a bool (B,C,T) tensor, "the implementation under test took the
        positive side".  LeakyReLU's derivative jumps at 0, so for an input within fp32 rounding of 0 (|x| < band) both
        sides are valid fp32 evaluations; there -- and only there -- the oracle follows `branch`.  self.n_ambiguous
        counts those elements."""
        p, P = self.plan, self.params
        self.n_ambiguous = 0
        for i in range(p.L):
            x = skel_conv(x, P["enc.layers.%d.0.weight" % i], self.masks["enc.layers.%d.0.mask" % i],
                          P["enc.layers.%d.0.bias" % i], 1, p.pad, p.pad_mode)
            pre = torch.matmul(self.pool_w[i], x)
            x = F.leaky_relu(pre, 0.2)
            if branch is not None:
                amb = pre.detach().abs() < band
                self.n_ambiguous += int(amb.sum())
                x = torch.where(amb, torch.where(branch[i], pre, 0.2 * pre), x)
        return x

    def min_abs_preactivation(self, joint_pos):
        """Smallest |LeakyReLU input| over all layers (tests skip draws where it is within fp32 rounding of 0)."""
        p, P = self.plan, self.params
        x = joint_pos.to(self.dtype).transpose(1, 2)
        m = float("inf")
        with torch.no_grad():
            for i in range(p.L):
                x = skel_conv(x, P["enc.layers.%d.0.weight" % i], self.masks["enc.layers.%d.0.mask" % i],
                              P["enc.layers.%d.0.bias" % i], 1, p.pad, p.pad_mode)
                x = torch.matmul(self.pool_w[i], x)
                m = min(m, float(x.abs().min()))
                x = F.leaky_relu(x, 0.2)
        return m

    def root_velocity(self, joint_pos, branch=None):
        """:220-235 — (B,T,72) normalised joint positions -> (B,T,3) normalised root velocity."""
        p = self.plan
        b, T, _ = joint_pos.shape
        latent = self.encode(joint_pos.to(self.dtype).transpose(1, 2), branch)
        k = latent.shape[1] // p.d_model
        feat = latent.reshape(b, k, p.d_model, T).transpose(2, 3).transpose(1, 2).reshape(b, T, -1)
        return F.linear(feat, self.params["fc_mapping.weight"], self.params["fc_mapping.bias"])

    def trajectory(self, pose, root_v):
        """gen_motion_w_trajectory (:292-305): pose (T,B,24,3), root_v (T,B,3) normalised.  Frame t >= 1 is moved by
        the running sum of the de-standardised velocities of frames 1..t; frame 0 stays."""
        v = self.mean_root + self.std_root * root_v
        out = [pose[0]]
        acc = torch.zeros_like(v[0])
        for t in range(1, pose.shape[0]):
            acc = acc + v[t]
            out.append(pose[t] + acc[:, None, :])
        return torch.stack(out, dim=0)

    def train_forward(self, data, backward=True, branch=None):
        """:196-265.  data = the 7-tuple of the loader; uses rot_pos [2], joint_pos [3], root_v [6]."""
        hp = self.hp
        rot_pos, joint_pos, root_v = data[2].to(self.dtype), data[3].to(self.dtype), data[6].to(self.dtype)
        b, T, _ = joint_pos.shape
        out_v = self.root_velocity(joint_pos, branch)
        l_v = ((out_v - root_v) ** 2).mean()
        if hp["use_accumulation_root_v"]:
            pose = rot_pos.reshape(b, T, 24, 3).transpose(0, 1)
            pred = self.trajectory(pose, out_v.transpose(0, 1))
            gt = self.trajectory(pose, root_v.transpose(0, 1))
            l_t = ((pred - gt) ** 2).mean()
        else:
            pred = gt = None
            l_t = torch.zeros(())
        total = hp["rec_root_v_w"] * l_v + hp["rec_root_trans_w"] * l_t
        if backward:
            for t in self.params.values():
                t.grad = None
            total.backward()
        return dict(total=total, rec_root_v=l_v, rec_root_trans=l_t, root_v_out=out_v, pred_traj=pred, gt_traj=gt)


def synthetic_traj_batch(b, T, seed=0, nj=24):
    """Loader-shaped 7-tuple for TrajectoryModel: normalised joint positions / root velocity ~ N(0,1), un-normalised
    root-relative pose ~ 0.3*N(0,1) with the root at the origin."""
    g = torch.Generator().manual_seed(seed)
    joint_pos = torch.randn(b, T, nj * 3, generator=g)
    rot_pos = 0.3 * torch.randn(b, T, nj, 3, generator=g)
    rot_pos[:, :, 0] = 0.0
    root_v = torch.randn(b, T, 3, generator=g)
    z144 = torch.zeros(b, T, nj * 6)
    z216 = torch.zeros(b, T, nj * 9)
    z72 = torch.zeros(b, T, nj * 3)
    return (z144, z216, rot_pos.reshape(b, T, nj * 3), joint_pos, z72, z72.clone(), root_v)


# --------------------------------------------------------------------------------------
# data path: MotionSeqData.__getitem__ (utils_motion_vae.py:124-187) — SURVEY.md §8 row f-3
# --------------------------------------------------------------------------------------
FPS_STRIDES = (1, 2, 3, 4, 5, 6, 8, 10, 12)            # utils_motion_vae.py:71


def rand_rotation_from_uniforms(randnums, deflection=1.0):
    """utils_motion_vae.py:16-57 (Graphics Gems III rand_rotation): three uniforms in [0,1] -> 3x3 matrix (float64)."""
    theta, phi, z = (float(v) for v in randnums)
    theta = theta * 2.0 * deflection * np.pi
    phi = phi * 2.0 * np.pi
    z = z * 2.0 * deflection
    r = np.sqrt(z)
    V = (np.sin(phi) * r, np.cos(phi) * r, np.sqrt(2.0 - z))
    st, ct = np.sin(theta), np.cos(theta)
    Rz = np.array(((ct, st, 0), (-st, ct, 0), (0, 0, 1)))
    return (np.outer(V, V) - np.eye(3)).dot(Rz)


def motion_item(seq, mean_std, T, crop_start, stride=1, root_rot=None):
    """One training clip as MotionSeqData.__getitem__ builds it, with the random choices made explicit.

    seq: (T_all, 579) array as stored on disk; mean_std: (2, 579) with zero std already replaced by 1 (:100-101);
    stride: the frame stride picked by change_fps (:66-83; 1 = no fps augmentation); crop_start: `random_t_idx` (:137),
    an index into the strided sequence; root_rot: the 3x3 matrix of rand_rotation_matrix() when random_root_rot_flag.
    Returns the 7-tuple of (T, n) float32 arrays (:187).
    """
    ori = np.asarray(seq)[0::stride, :]
    assert ori.shape[0] >= T and 0 <= crop_start <= ori.shape[0] - T
    std = (ori - mean_std[0][None, :]) / mean_std[1][None, :]                 # :142, float64 arithmetic
    raw = torch.from_numpy(np.ascontiguousarray(ori[crop_start:crop_start + T, :])).float()
    nrm = torch.from_numpy(np.ascontiguousarray(std[crop_start:crop_start + T, :])).float()
    r6 = raw[:, 0:144]
    rmat = raw[:, 144:360].clone()
    rot_pos = raw[:, 360:432]
    joint_pos, lin_v, ang_v, root_v = nrm[:, 360:432], nrm[:, 432:504], nrm[:, 504:576], nrm[:, 576:579]
    if root_rot is not None:                                                  # :161-185
        M = torch.from_numpy(np.asarray(root_rot)).float()[None].repeat(T, 1, 1)
        root = rmat[:, :9].reshape(-1, 3, 3)
        aug_v = torch.matmul(M, raw[:, 576:579][:, :, None]).squeeze(-1)
        root_v = ((aug_v.double() - torch.from_numpy(mean_std[0][None, 576:579])) /
                  torch.from_numpy(mean_std[1][None, 576:579])).float()
        rmat[:, :9] = torch.matmul(M, root).reshape(-1, 9)
        Rm = rmat.reshape(-1, 24, 3, 3)
        r6 = torch.stack((Rm[:, :, :, 0], Rm[:, :, :, 1]), dim=-2).reshape(T, 144)
    return tuple(np.ascontiguousarray(t.numpy()) for t in (r6, rmat, rot_pos, joint_pos, lin_v, ang_v, root_v))


def synthetic_motion_files(seed=77, lengths=(30, 47, 200, 12)):
    """Synthetic on-disk sequences (T_i x 579 float32) for the data-path fixtures; regenerated from the seed on both sides."""
    rng = np.random.default_rng(seed)
    return [rng.standard_normal((n, 579)).astype(np.float32) for n in lengths]


def make_adam(oracle, hp):
    """trainer_motion_vae.py:29-33 — Adam over requires_grad params + StepLR."""
    opt = torch.optim.Adam([t for _, t in oracle.trainable()], lr=hp["lr"], weight_decay=hp["weight_decay"])
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=hp["step_size"], gamma=hp["gamma"])
    return opt, sched


def synthetic_batch(b, T, seed=0, nj=24):
    """SURVEY.md §8(d) synthetic AMASS-shaped batch: valid rotations from N(0,1) 6D draws."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(b * T, nj, 6, generator=g)
    R = rot6d_to_matrix(raw)
    r6 = matrix_to_rot6d(R).reshape(b, T, nj * 6)
    rmat = R.reshape(b, T, nj * 9)
    root_v = torch.randn(b, T, 3, generator=g)
    z72 = torch.zeros(b, T, nj * 3)
    return (r6, rmat, z72, z72.clone(), z72.clone(), z72.clone(), root_v)


def level_eps(plan, b, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b * plan.edge_num[i + 1], plan.lat_dim[i], generator=g) for i in range(plan.L)]
