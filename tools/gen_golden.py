"""Generate tests/golden/*.npz|json by running the UNMODIFIED reference on CPU (build container only).

    python tools/gen_golden.py            # writes into tests/golden/

What is stored is data only: inputs, injected eps, and the reference's outputs / gradients /
post-Adam parameters (sampled for the big tensors).  Weights are NOT stored: both sides rebuild them
from the name-seeded recipe `oracle.hmvae_oracle.name_seeded_params` and load them through the
reference's own state_dict keys.
"""
import os
import sys
import json
import zlib

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

import ref_harness as H  # noqa: E402
from oracle import hmvae_oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def sample_idx(name, numel, n=64):
    rng = np.random.default_rng(zlib.crc32(("idx:" + name).encode()))
    return rng.integers(0, numel, size=min(n, numel))


def summarize_grads(named, store, prefix):
    none = []
    for name, p in named:
        if p.grad is None:
            none.append(name)
            continue
        g = p.grad.detach().reshape(-1).numpy()
        store[prefix + name + "|norm"] = np.float64(np.linalg.norm(g.astype(np.float64)))
        if g.size <= 20000:
            store[prefix + name + "|full"] = g.copy()
        else:
            idx = sample_idx(name, g.size)
            store[prefix + name + "|idx"] = idx
            store[prefix + name + "|val"] = g[idx].copy()
    return none


def topology_dump(ref_skel, parents_json):
    edges = ref_skel.get_edges(parents_json)
    out = {"edges0": [list(e) for e in edges], "levels": []}
    cur = edges
    for lvl in range(4):
        nb = ref_skel.find_neighbor(cur, 2)
        pool = ref_skel.SkeletonPool(cur, "mean", 3, last_pool=(lvl == 3))
        out["levels"].append({"neighbours": nb, "seq_list": pool.seq_list, "pooling_list": pool.pooling_list,
                              "new_edges": [list(e) for e in pool.new_edges]})
        if lvl < 3:
            cur = pool.new_edges
    return out


def fk_rot_golden(ref_tools, ref_fk_mod):
    g = torch.Generator().manual_seed(123)
    n = 32
    r6 = torch.randn(n, 24, 6, generator=g)
    r6[0, 0] = 0.0                                            # zero vector -> eps clamp, zero matrix
    r6[1, 3, 3:6] = r6[1, 3, 0:3] * 2.0                       # parallel a,b -> z = 0
    r6[2, 5, 3:6] = r6[2, 5, 0:3] * (1.0 + 1e-4) + 1e-5       # near-parallel
    r6[3] = r6[3] * 1e-3                                      # tiny magnitude
    r6[4] = r6[4] * 1e3                                       # huge magnitude
    jj = torch.arange(24).float()[:, None] * 6 + torch.arange(6).float()[None, :]
    r6[5] = torch.sin(0.37 * jj + 0.11)                       # SURVEY §7 analytic anchor input
    wR = torch.randn(n, 24, 3, 3, generator=g)
    wP = torch.randn(n, 24, 3, generator=g)
    x = r6.clone().requires_grad_(True)
    R = ref_tools.rotation_matrix_from_ortho6d(x)
    fk = ref_fk_mod.ForwardKinematicsLayer()
    P = fk(R)
    P6 = fk(x)                                                # 6D input path of the layer
    ((R * wR).sum() + (P * wP).sum()).backward()
    d6 = x.grad.clone()
    # FK-only backward w.r.t. rotation matrices
    Rin = R.detach().clone().requires_grad_(True)
    P2 = fk(Rin)
    (P2 * wP).sum().backward()
    # anchor: d(sum P^2)/d r6 for sample 5
    xa = r6[5:6].clone().requires_grad_(True)
    Pa = fk(ref_tools.rotation_matrix_from_ortho6d(xa))
    (Pa ** 2).sum().backward()
    ident = torch.eye(3)[None, None].repeat(1, 24, 1, 1)
    rest = fk(ident)
    return dict(r6=r6.numpy(), wR=wR.numpy(), wP=wP.numpy(), R=R.detach().numpy(), P=P.detach().numpy(),
                P_from6d=P6.detach().numpy(), d6=d6.numpy(), dR_fkonly=Rin.grad.numpy(),
                anchor_d6=xa.grad.numpy(), rest=rest.numpy())


def leaky_margin(cfg, plan, data, eps):
    """Smallest |LeakyReLU input| of the step (oracle restatement, both phases see the same activations).  LeakyReLU's
    derivative jumps at 0: an input within fp32 rounding of 0 makes the backward pass depend on summation order, so such a
    draw cannot pin anything (cf. traj_golden)."""
    import torch.nn.functional as F
    m = O.OracleVAE(cfg, O.name_seeded_params(plan), offsets=np.load(os.path.join(H.REF_ROOT, "utils/data/skeleton_offsets.npy")))
    mins = []
    orig = F.leaky_relu

    def lk(x, s=0.01, *a, **k):
        mins.append(float(x.detach().abs().min()))
        return orig(x, s, *a, **k)

    F.leaky_relu = lk
    try:
        m.train_forward(data, 0, [e.clone() for e in eps], backward=False)
    finally:
        F.leaky_relu = orig
    return min(mins)


def model_golden(cfg_name, B, seed, seq_len=None):
    import trainer_motion_vae as RT
    cfg = yaml.safe_load(open(os.path.join(H.REF_ROOT, "configs", cfg_name)))
    if seq_len is not None:                 # the T == 16 branch of the stride / upsampling ladders (:79-90, :107-118)
        cfg["train_seq_len"] = seq_len
    plan = O.Plan(cfg)
    tr = RT.Trainer(cfg)
    model = tr.model
    # name-seeded weights through the reference's own state_dict keys
    seeded = O.name_seeded_params(plan)
    sd = model.state_dict()
    for k, v in seeded.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        mk = k.replace("weight", "mask")
        if v.dim() == 3:
            assert torch.equal((sd[mk] == 0), (v == 0) | (sd[mk] == 0))
        sd[k] = v.clone()
    model.load_state_dict(sd)
    store = {}
    meta = {"cfg": cfg_name, "B": B, "state_dict_keys": list(sd.keys()),
            "trainable_order": [n for n, p in model.named_parameters() if p.requires_grad],
            "all_param_order": [n for n, p in model.named_parameters()]}
    T = cfg["train_seq_len"]
    while True:
        data = O.synthetic_batch(B, T, seed=seed)
        eps = O.level_eps(plan, B, seed=seed + 1)
        margin = leaky_margin(cfg, plan, data, eps)
        if margin >= 1.2e-5:        # ~10x the fp32 error of these sums (activations are O(1))
            break
        print("%s T=%d: seed %d has a LeakyReLU input at %.2e, trying the next seed" % (cfg_name, T, seed, margin))
        seed += 2
    meta["seed"] = seed
    store["min_abs_preact"] = np.float64(margin)
    store["in_r6"] = data[0].numpy()
    store["in_rmat"] = data[1].numpy()
    store["in_root_v"] = data[6].numpy()
    for i, e in enumerate(eps):
        store["eps%d" % i] = e.numpy()

    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    # mask / pool / unpool constants: record density + checksum so the oracle's derivation is pinned
    trainable_names = set(n for n, _ in named)
    for k, v in sd.items():
        if k.startswith("dec.enc."):
            continue
        if k not in trainable_names:
            store["const|" + k + "|sum"] = np.float64(v.double().sum().item())
            store["const|" + k + "|wsum"] = np.float64(
                (v.double().reshape(-1) * torch.arange(1, v.numel() + 1).double() % 7.0).sum().item())

    for phase, it in (("p1", 0), ("p2", cfg["iteration_interval"])):
        # capture intermediates with hooks on the reference modules
        caps = {}
        hooks = [model.enc.register_forward_hook(lambda m, i, o: caps.__setitem__("enc", o)),
                 model.dec.register_forward_hook(lambda m, i, o: caps.__setitem__("dec", o))]
        tr.gen_opt.zero_grad()
        orig_decode = model._decode

        def cap_decode(z_list, *a, **k):
            r = orig_decode(z_list, *a, **k)
            caps["decode"] = r
            caps["z_list"] = z_list
            return r

        model._decode = cap_decode
        dec_grad = {}

        def dec_hook(m, i, o):
            o.register_hook(lambda g: dec_grad.__setitem__("g", g.clone()))

        hooks.append(model.dec.register_forward_hook(dec_hook))
        with H.EpsInjector([e.clone() for e in eps]):
            ret = model(data, cfg, it)
        model._decode = orig_decode
        for h in hooks:
            h.remove()
        l_total, l_kl, l6, lrm, lp = [float(x) for x in ret[:5]]
        store[phase + "|losses"] = np.array([l_total, l_kl, l6, lrm, lp], dtype=np.float64)
        store[phase + "|kl_list"] = np.array([float(x) for x in ret[9]], dtype=np.float64)
        latent, zs = caps["enc"]
        for i, z in enumerate(zs):
            store[phase + "|z_dist%d" % i] = z.detach().numpy()
        store[phase + "|latent"] = latent.detach().numpy()
        for i, z in enumerate(caps["z_list"]):
            store[phase + "|z%d" % i] = z.detach().numpy()
        store[phase + "|dec_out"] = caps["dec"].detach().numpy()
        c6, R, pos = caps["decode"][:3]
        store[phase + "|R"] = R.detach().numpy()
        store[phase + "|pos"] = pos.detach().numpy()
        store[phase + "|d_dec_out"] = dec_grad["g"].numpy()
        none = summarize_grads(named, store, phase + "|grad|")
        meta[phase + "_none_grad"] = none

    # sampled branch of test(): decoder-only from a given z-list with the SEEDED weights (config 4 semantics)
    g = torch.Generator().manual_seed(seed + 7)
    zl = [torch.randn(B, plan.edge_num[i + 1], plan.lat_dim[i], generator=g) for i in range(plan.L)]
    with torch.no_grad():
        c6, R, pos = model._decode(zl)[:3]
    for i, z in enumerate(zl):
        store["samp|z%d" % i] = z.numpy()
    store["samp|c6"] = c6.numpy()
    store["samp|R"] = R.numpy()
    store["samp|pos"] = pos.numpy()
    # two full Trainer.gen_update steps (phase 1) from the seeded state: losses + sampled params
    model.load_state_dict(sd)
    tr2 = RT.Trainer(cfg)
    tr2.model.load_state_dict(sd)
    steps = []
    for s in range(2):
        with H.EpsInjector([e.clone() for e in eps]):
            out = tr2.gen_update(data, cfg, s, False)
        steps.append([float(x) for x in out])
    store["upd|returns"] = np.array(steps, dtype=np.float64)
    for n, p in tr2.model.named_parameters():
        if not p.requires_grad:
            continue
        v = p.detach().reshape(-1).numpy()
        idx = sample_idx("upd:" + n, v.size)
        store["upd|" + n + "|idx"] = idx
        store["upd|" + n + "|val"] = v[idx].copy()
        store["upd|" + n + "|delta"] = (v[idx].astype(np.float64) - seeded[n].reshape(-1).numpy()[idx].astype(np.float64))
    meta["lr_after_2"] = tr2.gen_opt.param_groups[0]["lr"]
    return store, meta


def traj_golden(B, T, seed):
    """TrajectoryModel (trajectory_pred_model.py): forward, losses, gradients, two Trainer.gen_update steps."""
    import trainer_motion_vae as RT
    cfg = yaml.safe_load(open(os.path.join(H.REF_ROOT, "configs", "trajectory_model.yaml")))
    cfg["train_seq_len"] = T
    plan = O.TrajPlan(cfg)
    tr = RT.Trainer(cfg)
    model = tr.model
    seeded = O.traj_name_seeded_params(plan)
    sd = model.state_dict()
    for k, v in seeded.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
        if v.dim() == 3:
            mk = k.replace("weight", "mask")
            assert torch.equal((sd[mk] == 0), (v == 0) | (sd[mk] == 0))
        sd[k] = v.clone()
    model.load_state_dict(sd)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    meta = {"cfg": "trajectory_model.yaml", "B": B, "T": T, "state_dict_keys": list(sd.keys()),
            "trainable_order": [n for n, _ in named], "all_param_order": [n for n, _ in model.named_parameters()]}
    store = {}
    trainable_names = set(n for n, _ in named)
    for k, v in sd.items():
        if k not in trainable_names:
            store["const|" + k + "|sum"] = np.float64(v.double().sum().item())
            store["const|" + k + "|wsum"] = np.float64(
                (v.double().reshape(-1) * torch.arange(1, v.numel() + 1).double() % 7.0).sum().item())
    store["mean_root"] = model.mean_vals[0, 576:579].numpy().copy()
    store["std_root"] = model.std_vals[0, 576:579].numpy().copy()
    data = O.synthetic_traj_batch(B, T, seed=seed)
    store["in_rot_pos"], store["in_joint_pos"], store["in_root_v"] = data[2].numpy(), data[3].numpy(), data[6].numpy()
    caps = {}
    hooks = [model.enc.register_forward_hook(lambda m, i, o: caps.__setitem__("latent", o)),
             model.fc_mapping.register_forward_hook(lambda m, i, o: caps.__setitem__("root_v_out", o))]
    for li, layer in enumerate(model.enc.layers):
        hooks.append(layer.register_forward_hook(lambda m, i, o, li=li: caps.__setitem__("act%d" % li, o)))
        hooks.append(layer[1].register_forward_hook(lambda m, i, o, li=li: caps.__setitem__("pre%d" % li, o)))
    tr.gen_opt.zero_grad()
    ret = model(data, cfg, 0)
    for h in hooks:
        h.remove()
    # LeakyReLU is discontinuous in its gradient at 0: a pre-activation within fp32 rounding of 0 makes the backward
    # pass depend on summation order.  Such a draw cannot pin anything; the caller moves on to the next seed.
    margin = min(float(caps["pre%d" % li].abs().min()) for li in range(cfg["num_layers"]))
    if margin < 2e-5:
        return None, margin
    store["min_abs_preact"] = np.float64(margin)
    store["ret"] = np.array([float(x) for x in ret], dtype=np.float64)
    store["root_v_out"] = caps["root_v_out"].detach().numpy()
    for li in range(cfg["num_layers"]):
        a = caps["act%d" % li].detach().reshape(-1).numpy()
        idx = sample_idx("act%d" % li, a.size, n=2048)
        store["act%d|idx" % li], store["act%d|val" % li] = idx, a[idx].copy()
        store["act%d|norm" % li] = np.float64(np.linalg.norm(a.astype(np.float64)))
    meta["none_grad"] = summarize_grads(named, store, "grad|")
    with torch.no_grad():
        gt, pred, _, _ = model.test(data, cfg, 0)
    store["test_gt"], store["test_pred"] = gt.numpy(), pred.numpy()
    tr2 = RT.Trainer(cfg)
    tr2.model.load_state_dict(sd)
    steps = []
    for s in range(2):
        steps.append([float(x) for x in tr2.gen_update(data, cfg, s, False)])
    store["upd|returns"] = np.array(steps, dtype=np.float64)
    for n, p in tr2.model.named_parameters():
        if not p.requires_grad:
            continue
        v = p.detach().reshape(-1).numpy()
        idx = sample_idx("upd:" + n, v.size)
        store["upd|" + n + "|idx"] = idx
        store["upd|" + n + "|val"] = v[idx].copy()
        store["upd|" + n + "|delta"] = (v[idx].astype(np.float64) - seeded[n].reshape(-1).numpy()[idx].astype(np.float64))
    return store, meta


def vae_fixtures(only=None):
    import skeleton as ref_skel
    import my_tools as ref_tools
    import fk_layer as ref_fk_mod
    if only is None:
        topo = topology_dump(ref_skel, os.path.join(H.REF_ROOT, "utils/data/joint24_parents.json"))
        json.dump(topo, open(os.path.join(OUT, "topology.json"), "w"))
        np.savez_compressed(os.path.join(OUT, "fk_rot6d.npz"), **fk_rot_golden(ref_tools, ref_fk_mod))
    metas = {}
    for tag, cfg, B, seed, T in (("len64", "len64_no_aug_hm_vae.yaml", 2, 11, None), ("len8", "len8_data_aug_hm_vae.yaml", 4, 21, None),
                                 ("len16", "len8_data_aug_hm_vae.yaml", 3, 51, 16)):
        if only is not None and tag != only:
            continue
        store, meta = model_golden(cfg, B, seed, T)
        np.savez_compressed(os.path.join(OUT, "model_%s.npz" % tag), **store)
        metas[tag] = meta
    mp = os.path.join(OUT, "model_meta.json")
    if only is not None and os.path.exists(mp):
        old = json.load(open(mp))
        old.update(metas)
        metas = old
    json.dump(metas, open(mp, "w"), indent=1)


def loader_golden():
    """MotionSeqData.__getitem__ (utils_motion_vae.py:124-187) on synthetic T x 579 files, every random choice logged."""
    import random
    import tempfile
    import utils_motion_vae as RU
    seqs = O.synthetic_motion_files()
    tmp = tempfile.mkdtemp()
    ids = {}
    for i, a in enumerate(seqs):
        np.save(os.path.join(tmp, "s%d.npy" % i), a)
        ids[str(i)] = "s%d.npy" % i
    jf = os.path.join(tmp, "ids.json")
    json.dump(ids, open(jf, "w"))
    ms_path = os.path.join(H.REF_ROOT, "utils/data/for_all_data_motion_model/all_amass_data_mean_std.npy")
    log = []
    orig_sample, orig_uniform = random.sample, np.random.uniform

    def sample(pop, k):
        r = orig_sample(pop, k)
        log.append(("s", int(r[0])))
        return r

    def uniform(*a, **k):
        r = orig_uniform(*a, **k)
        log.append(("u", [float(v) for v in np.asarray(r).reshape(-1)]))
        return r

    store, meta = {}, []
    random.seed(5)
    np.random.seed(6)
    random.sample, np.random.uniform = sample, uniform
    try:
        ci = 0
        for T, fps, rot in ((16, False, False), (16, True, False), (16, False, True), (8, True, True)):
            ds = RU.MotionSeqData(tmp, jf, ms_path, {"train_seq_len": T}, fps_aug_flag=fps, random_root_rot_flag=rot)
            for idx in range(len(seqs)):
                del log[:]
                out = ds[idx]
                # replay the log with the control flow of __getitem__ / change_fps to name the choices
                pos, cur = 0, idx
                while True:
                    n, stride = seqs[cur].shape[0], 1
                    if fps:
                        for _ in range(10):
                            f = log[pos][1]
                            pos += 1
                            if len(range(0, n, f)) >= T:
                                stride = f
                                break
                    n2 = len(range(0, n, stride))
                    if T <= n2:
                        crop = log[pos][1]
                        pos += 1
                        break
                    cur = log[pos][1]
                    pos += 1
                rn = None
                if rot:
                    rn = log[pos][1]
                    pos += 1
                assert pos == len(log), (pos, log)
                for k, t in enumerate(out):
                    store["c%d|out%d" % (ci, k)] = np.asarray(t.numpy(), dtype=np.float32)
                meta.append({"case": ci, "T": T, "fps": fps, "rot": rot, "asked": idx, "seq": cur, "stride": stride,
                             "crop": crop, "randnums": rn})
                ci += 1
    finally:
        random.sample, np.random.uniform = orig_sample, orig_uniform
    np.savez_compressed(os.path.join(OUT, "loader.npz"), **store)
    json.dump(meta, open(os.path.join(OUT, "loader_meta.json"), "w"), indent=1)


def main():
    only_traj = "--traj-only" in sys.argv
    only_loader = "--loader-only" in sys.argv
    H.install()
    os.makedirs(OUT, exist_ok=True)
    if only_loader:
        loader_golden()
        return
    for tag in ("len16", "len8", "len64"):
        if "--%s-only" % tag in sys.argv:
            import skeleton, my_tools, fk_layer  # noqa: F401  (same import order as the full run)
            vae_fixtures(only=tag)
            return
    if not only_traj:
        vae_fixtures()
        loader_golden()
    tmetas = {}
    for tag, B, T, seed in (("len128", 2, 128, 31), ("len40", 3, 40, 41)):
        store = None
        while store is None:
            store, meta = traj_golden(B, T, seed)
            if store is None:
                print("traj %s: seed %d has a LeakyReLU input at %.2e, trying the next seed" % (tag, seed, meta))
                seed += 1
        meta["seed"] = seed
        np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % tag), **store)
        tmetas[tag] = meta
    json.dump(tmetas, open(os.path.join(OUT, "traj_meta.json"), "w"), indent=1)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
