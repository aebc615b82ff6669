"""bench.py — hm-vae training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): train sequences/sec, len-64 24-joint 6D sequences, TwoHierSAVAE fwd+bwd+Adam,
configs[1] = len64_no_aug_hm_vae.yaml at batch 256 per GPU (weak scaling: B = 256*N), synthetic AMASS-shaped data
resident in HBM, fp32.  One process per GPU; for N>1 launch with torch.distributed.run (RCCL all-reduce of the flat
gradient bucket).  Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      dominant kernel family (fp32-MFMA gather-GEMM / wgrad), mask-effective FLOPs / HIP-event time
  fk_roofline   fused rot6D->R->FK decode tail at N=262144 frames (config 4), algorithmic bytes / HIP-event time
  cpu_baseline  the CPU oracle's training step on the host cores (rank 0, N=1 only), bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6    # same guide: bf16 MFMA = 16x the fp32 rate (~2.5 PFLOP/s dense)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.3 TB/s achievable with a float4 copy)
# fp32-grade products on the bf16 pipe cost SIX bf16 MFMAs per fp32-equivalent block (three exact planes per operand, the
# six significant cross terms): the ceiling of such a launch in fp32-equivalent terms
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def effective_peak(pipe_ms):
    """Time-weighted ceiling of a kernel family that mixes the two pipes: launches that run three-plane bf16 MFMAs are
    priced at 2516.6 / 6 = 419.4 TFLOP/s (fp32-equivalent), plain fp32 MFMA launches (and the reduce launches) at 157.3."""
    tot = pipe_ms["bf16x3"] + pipe_ms["fp32"]
    if tot <= 0:
        return PEAK_F32_MFMA_TFLOPS
    return (pipe_ms["bf16x3"] * PEAK_BF16X3_TFLOPS + pipe_ms["fp32"] * PEAK_F32_MFMA_TFLOPS) / tot


def synthetic_device_batch(B, T, seed, device):
    """N(0,1) 6D draws -> valid rotations; built with the HIP lift so no CPU oracle is involved."""
    from hm_vae_amd import my_tools
    g = torch.Generator(device="cpu").manual_seed(seed)
    raw = torch.randn(B * T, 24, 6, generator=g).to(device)
    R = my_tools.rotation_matrix_from_ortho6d(raw)                       # (B*T,24,3,3)
    r6 = torch.cat((R[..., :, 0], R[..., :, 1]), dim=-1).reshape(B, T, 144).contiguous()
    rmat = R.reshape(B, T, 216).contiguous()
    z72 = torch.zeros(B, T, 72, device=device)
    root_v = torch.randn(B, T, 3, generator=g).to(device)
    return (r6, rmat, z72, z72, z72, z72, root_v)


class KernelTimer:
    """HIP-event timing of individual kernel launches: the events are recorded on the launch stream immediately around
    the C-ABI call that launches the kernel (not around the Python that builds its descriptor), so an interval is the
    kernel's own duration plus one launch gap."""

    def __init__(self):
        self.records = []
        self.current = None            # (label, useful flops, issued flops) of the operator being launched
        self.pipe = None               # "bf16x3" / "fp32" of the operator being launched, or None = by entry point

    def wrap_op(self, obj, name, label, flops_fn, pipe_fn=None):
        """Python-level: tags the launches made inside obj.name(...) with a family label and their flops.
        pipe_fn(*args) -> "bf16x3" | "fp32": the matrix pipe the operator's launches run on (None: by entry point)."""
        orig = getattr(obj, name)
        timer = self

        def tagged(*a, **k):
            fl = flops_fn(*a, **k)
            timer.current = [label] + list(fl if isinstance(fl, tuple) else (fl, fl))
            timer.pipe = pipe_fn(*a, **k) if pipe_fn is not None else None
            try:
                return orig(*a, **k)
            finally:
                timer.current = None

        setattr(obj, name, tagged)
        return orig

    def wrap_c(self, lib, name):
        """C-ABI level: events around the launch itself; the flops of the operator are booked on its first launch."""
        orig = getattr(lib, name)
        timer = self

        def timed(*a):
            if timer.current is None:
                return orig(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a)
            e1.record()
            label, fl, issued = timer.current
            pipe = timer.pipe or ("bf16x3" if name == "hmvae_gather_gemm_v4" else "fp32")
            timer.records.append((label, name, e0, e1, fl, issued, pipe))
            timer.current[1] = timer.current[2] = 0.0
            return r

        setattr(lib, name, timed)
        return orig

    def summary(self):
        torch.cuda.synchronize()
        agg, per_kernel = {}, {}
        self.pipe_ms = {}                  # label -> {"bf16x3": ms, "fp32": ms}
        for label, cname, e0, e1, fl, issued, pipe in self.records:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(label, [0.0, 0.0, 0, 0.0])
            a[0] += ms
            a[1] += fl
            a[2] += 1
            a[3] += issued
            pm = self.pipe_ms.setdefault(label, {"bf16x3": 0.0, "fp32": 0.0})
            pm[pipe] += ms
            k = per_kernel.setdefault(cname, [0.0, 0, 0.0, 0.0])
            k[0] += ms
            k[1] += 1
            k[2] += fl
            k[3] += issued
        self.per_kernel = per_kernel
        return agg


def instrument_gemms(engine, timer, bp):
    """Tag every MFMA operator of the engine (flops = mask-effective 2*MACs of that launch at this batch) and time the
    launches behind them at the C ABI."""
    from hm_vae_amd import _lib
    restore = []
    all_ops = engine.enc_conv + engine.dec_conv + engine.enc_lin + list(engine.dec_lin.values())
    fam = "gather_gemm (gather_gemm4_kernel | gather_gemm2_kernel | conv_window_kernel)"

    def issued_per_col(side, bp_):
        """Flops per batch column the matrix pipe executes for this side's tuned launch: fp32 MFMAs of the v2 / v3 kernels
        (2 per multiply-add), or the SIX bf16 MFMAs per 16x16x32 block of the v4 kernel (the three-plane cross terms)."""
        n_items = sum(len(l) for l in side.pairs_host)
        p4 = side._tuned_v4.get(bp_)
        if p4 is None:
            return 2.0 * side.g.mfma_cost() * n_items
        sd = side.v4[p4[0]]
        nt = sd.g_ntiles.cpu().numpy()
        tiles = ((nt + p4[0] - 1) // p4[0]) * p4[0]          # a wave with one live tile multiplies all mtw of its tiles
        return 6.0 * 2.0 * 16.0 * float((tiles * (sd.g.g_nc4 // 8) * 32).sum()) * n_items
    for op in all_ops:
        useful_f = 2.0 * op.useful_macs_per_col          # algorithmic: the masked convolution's own multiply-adds
        for side in (op.fwd, op.bwd):
            if side is None:
                continue
            restore.append((side, "run", timer.wrap_op(side, "run", fam, lambda *a, _f=useful_f, _s=side, **k:
                                                       (_f * a[2], issued_per_col(_s, a[2]) * a[2]) if len(a) > 2 else (0.0, 0.0))))
        restore.append((op, "backward_weight", timer.wrap_op(op, "backward_weight",
                                                             "wgrad (wgrad_bf3_kernel | wgrad_pq_kernel | wgrad_lds_kernel | wgrad_kernel | wreduce_kernel)",
                                                             lambda *a, _f=useful_f, **k: _f * a[3],
                                                             pipe_fn=lambda *a, _o=op, **k: "bf16x3" if (_o.wg_mode(a[3]) in (2, 3, 4) or
                                                                                                          (k.get("src") is not None and _o.use_pq(a[3]))) else "fp32")))
    lib = _lib.load()
    for cname in ("hmvae_gather_gemm_v2", "hmvae_gather_gemm_v3", "hmvae_gather_gemm_v4", "hmvae_wgrad_gemm", "hmvae_wgrad_pq", "hmvae_wgrad_reduce"):
        restore.append((lib, cname, timer.wrap_c(lib, cname)))
    return restore


def load_traffic(kernel, batch, phase):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/profile_round.sh + tools/pmc_traffic.py:
    separate FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 x2 fetch correction).  None when the profile
    on disk is for another workload."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files or batch != 256 or phase != 2:
        return None, None
    d = json.load(open(files[-1]))
    # the family label names its kernels: "gather_gemm (a | b)" -> launch-weighted mean over a and b
    names = [n.strip() for n in kernel[kernel.index("(") + 1:kernel.rindex(")")].split("|")] if "(" in kernel else [kernel]
    # alternative kernels for the same launch -> launch-weighted mean; the weight-gradient family is a GEMM (direct or
    # LDS-staged kernel) plus its reduce: two parts of one logical launch -> mean over the GEMM kernels + the reduce
    parts = kernel.startswith("wgrad")
    tot = cnt = 0.0
    extra = 0.0
    for n in names:
        k = d["kernels"].get(n)
        if not k:
            continue
        if parts and n == "wreduce_kernel":
            extra += k["traffic_bytes_per_launch"]
            continue
        c = max(k["launches_fetch_pass"], 1)
        tot += k["traffic_bytes_per_launch"] * c
        cnt += c
    if not cnt:
        return None, None
    return tot / cnt + extra, os.path.basename(files[-1])



def fk_microbench(device, iters=20):
    """Config 4 kernel microbench: SoA decoder output (144 x 64 x 4096) -> AoS R and joint positions (+ optionally the
    frame-major copy of the 6D input that `_decode` returns)."""
    from hm_vae_amd import ops
    import numpy as np
    B, T = 4096, 64
    bp = ops.pad_batch(B)
    off = torch.from_numpy(np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "skeleton_offsets.npy"))).to(device)
    dec = torch.randn(144 * T, bp, device=device)
    c6 = torch.empty(B, T, 24, 6, device=device)
    R = torch.empty(B, T, 24, 3, 3, device=device)
    P = torch.empty(B, T, 24, 3, device=device)

    def timed(fn):
        for _ in range(3):
            fn()
        evs = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    n = B * T
    avg = timed(lambda: ops.decode_tail_fwd(dec, off, T, B, bp, None, R, P))           # what test()/gen_seq launch
    avg_c6 = timed(lambda: ops.decode_tail_fwd(dec, off, T, B, bp, c6, R, P))          # what _decode launches
    fused_bytes = 1728                                # 576 B 6D in, 864 B R + 288 B positions out
    # AoS FK kernel (ForwardKinematicsLayer on rotation matrices): 864 B in, 288 B out per frame
    rot = R.reshape(n, 24, 9)
    pos = torch.empty(n, 24, 3, device=device)
    par = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
    avg2 = timed(lambda: ops.fk_fwd(rot, 9, off, 0, par, n, None, pos))
    r6 = c6.reshape(n, 24, 6)
    avg3 = timed(lambda: ops.fk_fwd(r6, 6, off, 0, par, n, None, pos))
    # backward of the fused lift + FK (6D 576 + dR 864 + dpos 288 read, d6D 576 written = 2 304 B/frame)
    dR = torch.randn(n, 24, 9, device=device)
    dP = torch.randn(n, 24, 3, device=device)
    d6 = torch.empty(n, 24, 6, device=device)
    avg4 = timed(lambda: ops.fk_bwd(r6, 6, off, 0, par, n, dR, dP, d6))
    del dR, dP, d6
    # the training step's tail at the same frame count: 6D -> R -> FK -> three MSEs and their backward in one launch
    # (decoder output 576 + targets 576 + 864 + 288 read, d(decoder output) 576 written = 2 880 B/frame)
    g6, gR, gP = (torch.randn(c * T, bp, device=device) for c in (144, 216, 72))
    ddec = torch.empty(144 * T, bp, device=device)
    part = torch.zeros(ops.decode_tail_loss_blocks(T, B, bp), 3, device=device)
    avg5 = timed(lambda: ops.decode_tail_loss(dec, g6, gR, gP, off, T, B, bp, 1.0 / n, 1.0 / n, 1.0 / n, ddec, part))
    del g6, gR, gP, ddec

    def gbs(bytes_per_frame, ms):
        return bytes_per_frame * n / (ms * 1e-3) / 1e9

    def traffic_of(kernel):
        """HBM-side bytes per launch from the committed PMC passes of tools/profile_fk.sh (None if absent)."""
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fk_traffic.json")))
        if not files:
            return None
        k = json.load(open(files[-1])).get("kernels", {}).get(kernel)
        return None if not k else k["traffic_bytes_per_launch"]

    def entry(kernel, bytes_per_frame, ms, note=None):
        e = {"kernel": kernel, "bytes_per_frame": bytes_per_frame, "avg_us": ms * 1e3, "achieved": gbs(bytes_per_frame, ms),
             "frac": gbs(bytes_per_frame, ms) / PEAK_HBM_GBS, "traffic": traffic_of(kernel)}
        if note:
            e["note"] = note
        return e

    return {"kernel": "decode_tail_fwd_kernel<false>", "frames": n, "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
            "achieved": gbs(fused_bytes, avg), "frac": gbs(fused_bytes, avg) / PEAK_HBM_GBS,
            "bytes_per_frame_algorithmic": fused_bytes, "avg_us": avg * 1e3,
            "traffic": traffic_of("decode_tail_fwd_kernel<false>"),
            "fk_bwd_aos": entry("fk_bwd_aos_kernel<6>", 2304, avg4, "backward of the fused lift + FK, frame-major"),
            "tail_loss": entry("decode_tail_loss_kernel<false>" if os.environ.get("HMVAE_TAIL3", "1") == "0" else
                               "decode_tail_loss3_kernel<false>", 2880, avg5,
                               "training tail (lift + FK + 3 MSE + their backward) at config 4's frame count"),
            "note": "6D (SoA) in, R + positions (frame-major) out: the bytes moved are the algorithmic 1728 B/frame",
            "with_6d_copy": {"kernel": "decode_tail_fwd_kernel<true>", "bytes_per_frame_moved": 2304, "avg_us": avg_c6 * 1e3,
                             "achieved_moved": gbs(2304, avg_c6), "frac_moved": gbs(2304, avg_c6) / PEAK_HBM_GBS,
                             "achieved": gbs(fused_bytes, avg_c6), "frac": gbs(fused_bytes, avg_c6) / PEAK_HBM_GBS,
                             "traffic": traffic_of("decode_tail_fwd_kernel<true>"),
                             "note": "_decode also returns the frame-major copy of the 6D input: 576 B/frame more to "
                                     "write; achieved / frac count SURVEY's 1728 algorithmic bytes only"},
            "fk_only_aos": entry("fk_fwd_aos_kernel<9>", 1152, avg2, "ForwardKinematicsLayer on rotation matrices"),
            "fk_rot6d_aos": entry("fk_fwd_aos_kernel<6>", 864, avg3, "6D in, positions out")}


def sampling_bench(tr, cfg, device, iters=10):
    """BASELINE config 4 end to end: decoder-only sampling, z-list randn (4096,14,12), (4096,9,24), (4096,7,24) x 2 ->
    TwoHierSAVAEModel._decode (seq_two_hier_sa_vae.py:609-610,629-632 -> :436-474) -> 6D, R, joint positions for 262 144
    frames.  The latents are resident in HBM; the call is the API's (weight layouts re-packed, outputs allocated)."""
    from hm_vae_amd import _lib
    B = 4096
    model = tr.model
    eng = model.engine()
    p = eng.p
    g = torch.Generator(device="cpu").manual_seed(4)
    zl = [torch.randn(B, p.edge_num[i + 1], p.lat_dim[i], generator=g).to(device) for i in range(p.L)]
    for _ in range(2):
        model._decode(zl)                      # warm-up: workspace, launch shapes of the 4096-column problems
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        model._decode(zl)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    # the decoder's four forward convolutions alone: HIP events around their C-ABI launches, two more calls
    timer = KernelTimer()
    restore = []
    fl = 0.0
    for op in eng.dec_conv:
        f = 2.0 * op.useful_macs_per_col
        fl += f * B
        restore.append((op.fwd, "run", timer.wrap_op(op.fwd, "run", "decoder forward convolutions",
                                                     lambda *a, _f=f, **k: _f * B)))
    lib = _lib.load()
    for cname in ("hmvae_gather_gemm_v2", "hmvae_gather_gemm_v3", "hmvae_gather_gemm_v4"):
        restore.append((lib, cname, timer.wrap_c(lib, cname)))
    n_inst = 2
    for _ in range(n_inst):
        model._decode(zl)
    agg = timer.summary()
    for obj, name, orig in restore:
        setattr(obj, name, orig)
    cms = agg["decoder forward convolutions"][0] / n_inst
    peff = effective_peak(timer.pipe_ms["decoder forward convolutions"])
    bp = eng.workspace(B)["bp"]

    def pick(side):
        if side._tuned_v4.get(bp) is not None:
            return "v4"
        t = side._tuned.get(bp)
        return "v3" if (t and t[0] == "v3") else "v2"
    picks = [pick(op.fwd) for op in eng.dec_conv]
    out = {"workload": "len_64_test_interpolation.yaml decoder-only sampling, batch 4096: z-list -> _decode -> 6D, R, positions",
           "value": B / (ms * 1e-3), "unit": "sequences/s", "ms_per_call": ms, "frames": B * p.T, "calls": iters,
           "decoder_forward": {"ms": cms, "flops_effective": fl, "TFLOP/s": fl / (cms * 1e-3) / 1e12,
                               "frac": fl / (cms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "peak": PEAK_F32_MFMA_TFLOPS,
                               "peak_effective": peff, "frac_effective": fl / (cms * 1e-3) / 1e12 / peff,
                               "kernels": picks,
                               "timing": "HIP events around the four forward gather-GEMM launches, %d extra calls" % n_inst}}
    eng._ws.pop(B, None)                       # ~4 GB of buffers nobody needs afterwards
    torch.cuda.empty_cache()
    return out


def traj_bench(device, steps=30, warmup=10):
    """BASELINE config 5's per-GPU shape: trajectory_model.yaml with train_seq_len 64, 512 sequences per GPU (4096 / 8
    ranks), TrajectoryModel forward + both losses + backward + Adam (trajectory_pred_model.py:206-303) under hipGraph
    replay, synthetic batches resident in HBM."""
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from hm_vae_amd.utils_motion_vae import SyntheticMotionLoader
    from hm_vae_amd.graph_step import GraphedStep
    cfg = get_config(os.path.join(ROOT, "configs", "trajectory_model.yaml"))
    B, T = 512, 64
    cfg["batch_size"], cfg["train_seq_len"] = B, T
    tr = Trainer(cfg)
    tr.cuda()
    batches = list(SyntheticMotionLoader(B, T, device, 4, 0, with_positions=True))
    gs = GraphedStep(tr, cfg, batches[0], 0, use_graph=True)
    for i in range(warmup):
        gs.set_batch(batches[i % 4])
        gs.run(lag=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        gs.set_batch(batches[i % 4])
        gs.run(lag=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng = tr.model.engine(T)
    tf = 3.0 * eng.flops_per_seq * B / dt / 1e12
    out = {"workload": "trajectory_model.yaml, train_seq_len 64, batch 512 per GPU (config 5 = 8 x 512), fwd+bwd+Adam, "
                       "hipGraph replay", "value": B / dt, "unit": "sequences/s", "ms_per_step": dt * 1e3, "steps": steps,
           "conv_flops_per_step_effective": 3.0 * eng.flops_per_seq * B, "achieved_TFLOP/s": tf,
           "peak": PEAK_F32_MFMA_TFLOPS, "frac": tf / PEAK_F32_MFMA_TFLOPS, "loss_total": float(gs.last_losses()[0])}
    del gs, tr
    torch.cuda.empty_cache()
    return out


def _host_cores():
    """(usable logical CPUs of this process, physical cores among them)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    phys = usable
    try:
        import psutil
        phys = min(usable, psutil.cpu_count(logical=False) or usable)
    except Exception:
        pass
    # a container's CPU share (cgroup quota) is what the box really grants: more threads than that only oversubscribe
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0])
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                usable = max(1, min(usable, int(round(quota / period))))
                phys = min(phys, usable)
        except (OSError, ValueError, IndexError):
            pass
    return usable, phys


def cpu_baseline(cfg, budget_s=28.0):
    """The CPU oracle's full training step (fwd + bwd + Adam, SURVEY 8(d) "CPU baseline") on the host cores: a thread
    sweep at B=64 (1 / 8 / 16 / 32 / physical / usable cores), then B=256 at the best thread count.  Bounded: every
    point gets one warm-up step and then timed steps until its share of the budget is used (a point whose warm-up alone
    exceeds its share reports that step).  `value` is the best sequences/s found; the 1-thread figure sits beside it."""
    import numpy as np
    from oracle import hmvae_oracle as O
    plan = O.Plan(cfg)
    off = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "skeleton_offsets.npy"))
    it = cfg["iteration_interval"]
    usable, phys = _host_cores()
    sweep = sorted(set(t for t in (1, 8, 16, 32, phys, usable) if 1 <= t <= usable))
    was = torch.get_num_threads()

    def point(threads, B, share):
        torch.set_num_threads(threads)
        m = O.OracleVAE(cfg, O.name_seeded_params(plan), offsets=off)
        opt, sched = O.make_adam(m, cfg)
        batch = O.synthetic_batch(B, cfg["train_seq_len"], seed=0)
        times = []
        t_start = time.time()
        n = 0
        while True:
            t0 = time.time()
            opt.zero_grad()
            m.train_forward(batch, it, O.level_eps(plan, B, seed=n))
            opt.step()
            times.append(time.time() - t0)
            n += 1
            if time.time() - t_start > share or n >= 6:
                break
        timed = times[1:] if len(times) > 1 else times
        avg = sum(timed) / len(timed)
        return {"threads": threads, "batch": B, "sequences_per_s": B / avg, "ms_per_step": avg * 1e3,
                "timed_steps": len(timed), "warm": len(times) > 1}

    share = budget_s / (len(sweep) + 3.0)            # B=256 costs ~4x: it gets three shares
    points = [point(t, 64, share) for t in sweep]
    best = max(points, key=lambda r: r["sequences_per_s"])
    big = point(best["threads"], 256, 3.0 * share)
    torch.set_num_threads(was)
    top = max([best, big], key=lambda r: r["sequences_per_s"])
    one = next(r for r in points if r["threads"] == 1)
    return {"value": top["sequences_per_s"], "unit": "sequences/s", "cores": top["threads"], "kind": "port",
            "sample": "pure-PyTorch CPU oracle, len64 config, fwd+bwd+Adam: thread sweep %s at B=64 (one warm-up + up to 5 "
                      "timed steps per point, ~%.0f s each), then B=256 at the best thread count; value = best point "
                      "(B=%d, %d threads)" % (sweep, share, top["batch"], top["threads"]),
            "ms_per_step": top["ms_per_step"], "batch": top["batch"],
            "one_thread": {"value": one["sequences_per_s"], "batch": 64, "ms_per_step": one["ms_per_step"]},
            "b64_best": best, "b256_best_threads": big, "sweep": points,
            "host": {"usable_cpus": usable, "physical_cores": phys, "logical_cpus": os.cpu_count()}}


def kfd_gpu_count(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with
    simd_count > 0 is a GPU, the others are CPUs) -- no HIP call, nothing initialised: the launcher parent must stay a
    process that has never touched the GPU.  None when the topology is not readable (the ranks then report for themselves:
    a rank without a device fails in ddp.init_from_env / require_gpu and the launcher relays its exit code)."""
    import glob
    nodes = glob.glob(os.path.join(root, "*", "properties"))
    if not nodes:
        return None
    n = 0
    try:
        for path in nodes:
            for line in open(path):
                f = line.split()
                if len(f) == 2 and f[0] == "simd_count" and int(f[1]) > 0:
                    n += 1
    except (OSError, ValueError):
        return None
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if vis is not None and vis.strip() != "":
        n = min(n, len([v for v in vis.split(",") if v.strip() != ""]))
    return n


def self_launch(n):
    """Re-run this command line as N ranks of torch.distributed.run on 127.0.0.1 (a free port); returns its exit code."""
    import socket
    import subprocess
    n_dev = kfd_gpu_count()
    if n_dev is not None and n_dev < n and os.environ.get("HMVAE_SINGLE_DEVICE", "0") != "1":
        sys.stderr.write("bench.py: --gpus %d but this node shows %d GPU(s)\n" % (n, n_dev))
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU")
    ap.add_argument("--config", type=str, default=os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
    ap.add_argument("--phase", type=int, default=2, choices=(1, 2))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fk", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--sync-losses", type=int, default=0,
                    help="1: block on every step's loss read-back (like the reference's .item() calls); 0: the read-back "
                         "lags one step (pinned buffer + event), so the host queues step n+1 while step n runs")
    ap.add_argument("--no-phase1", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-4 sampling and config-5 trajectory lines")
    ap.add_argument("--only", type=str, default="", choices=("", "sampling", "traj"),
                    help="profiling aid: run only that extra section (N = 1) and print its object")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run plainly: become the launcher.  N ranks through torch.distributed.run (one
        # process per GPU, RCCL), started BEFORE anything here touches the GPU; this process only relays the exit code.
        return self_launch(args.gpus)

    from hm_vae_amd import ddp, _lib
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from hm_vae_amd.graph_step import GraphedStep

    rank, world, local = ddp.init_from_env()
    if world != args.gpus:
        # never measure fewer GPUs than asked for and print it as if it were the request
        sys.stderr.write("bench.py: --gpus %d but the launch has WORLD_SIZE=%d\n" % (args.gpus, world))
        if world > 1:
            torch.distributed.destroy_process_group()
        return 2
    _lib.require_gpu()
    # HMVAE_SINGLE_DEVICE=1 (with HMVAE_DIST_BACKEND=gloo): every rank on cuda:0 -- a control-flow rehearsal of the N > 1
    # launch on a one-GPU box, not a measurement
    one_dev = os.environ.get("HMVAE_SINGLE_DEVICE", "0") == "1"
    device = torch.device("cuda", local if (world > 1 and not one_dev) else 0)
    torch.cuda.set_device(device)
    cfg = get_config(args.config)
    torch.manual_seed(1234)
    if args.only == "traj":
        print(json.dumps({"traj": traj_bench(device, steps=args.steps, warmup=args.warmup)}))
        return 0
    tr = Trainer(cfg)
    tr.cuda()
    if args.only == "sampling":
        print(json.dumps({"sampling": sampling_bench(tr, cfg, device, iters=args.steps)}))
        return 0
    ddp.broadcast_parameters([p for p in tr.model.parameters() if p.requires_grad])
    T = cfg["train_seq_len"]
    B = args.batch
    data = synthetic_device_batch(B, T, seed=rank, device=device)
    it0 = cfg["iteration_interval"] if args.phase == 2 else 0

    step = GraphedStep(tr, cfg, data, it0, use_graph=bool(args.graph))
    lag = bool(args.graph) and not args.sync_losses
    # the warm-up runs the step exactly as the timed loop does (same read-back mode: its pinned buffers and events are made
    # on first use -- a millisecond of host work that belongs to no step -- and the capture happens in the first call)
    step.run(lag=lag)
    # Settling: the chip takes tens of milliseconds of continuous work to reach the clocks it then holds (a 20-step window
    # timed right behind a 5-step warm-up read 1.40 ms/step where the 100-step window behind 20 reads 1.365 on the same box);
    # the step is replayed until ~0.1 s of work has run, untimed, whatever W says.  Reported as config.settle_steps.
    settle = max(0, int(os.environ.get("HMVAE_BENCH_SETTLE", "64")) - args.warmup)
    for _ in range(settle):
        step.run(lag=lag)
    for _ in range(max(0, args.warmup - 1)):
        step.run(lag=lag)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step.run(lag=lag)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world * args.steps / elapsed
    # "shipped": every launch shape of the timed step came from hm_vae_amd/assets/tune_mi355x.json (choices made inside the
    # step, keyed by problem shape and batch); "fresh": some were timed alone in this process (4-6 % slower steps)
    from hm_vae_amd import ops as _ops
    tuned_state = _ops.tune_state()

    out = {"metric": "train sequences/sec (len-64, 24-joint, TwoHierSAVAE fwd+bwd+Adam)", "value": value,
           "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "len64_no_aug_hm_vae.yaml, batch %d per GPU, T=64, J=24, 6D; phase %d (iterations >= "
                                  "iteration_interval)" % (B, args.phase) if args.phase == 2 else
                      "len64_no_aug_hm_vae.yaml, batch %d per GPU, phase 1" % B,
                      "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world,
                      "launch": "hipGraph replay" if args.graph else "eager",
                      "tuned": tuned_state, "settle_steps": settle,
                      "loss_readback": "every step, lagging one step (non-blocking)" if lag else "every step, blocking"},
           "loss_total": float(step.last_losses()[0]) if lag else float(losses[0])}

    if not args.no_phase1 and args.phase == 2:
        # SURVEY 8(d) "report phase 1 too": iterations < iteration_interval -- the shallow latent head is detached
        # (seq_two_hier_sa_vae.py:380-385), two fewer tensors in the bucket.  Same trainer, same batch, every rank;
        # its own capture, a shorter timed run bracketed like the headline.
        step1 = GraphedStep(tr, cfg, data, 0, use_graph=bool(args.graph))
        k1 = max(10, args.steps // 2)
        for _ in range(max(3, args.warmup // 2)):
            step1.run(lag=lag)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t1 = time.perf_counter()
        for _ in range(k1):
            step1.run(lag=lag)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        e1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e1], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            e1 = float(t.item())
        out["phase1"] = {"value": B * world * k1 / e1, "unit": "sequences/s", "ms_per_step": e1 / k1 * 1e3, "steps": k1,
                         "note": "iterations < iteration_interval: level-0 latent branch detached"}
        del step1

    # N = 1 only: the extra eager steps below go through Trainer.gen_update, whose gradient all-reduce is a collective --
    # rank 0 alone must not enter it
    if rank == 0 and world == 1 and not args.no_kernel_timing:
        # instrumented extra steps (eager, same workload): per-launch HIP events around every MFMA kernel
        eng = tr.model.engine()
        timer = KernelTimer()
        bp = eng.workspace(B)["bp"]
        restore = instrument_gemms(eng, timer, bp)
        eager = GraphedStep(tr, cfg, data, it0, use_graph=False)
        n_inst = 5
        # one stream for these steps: a launch's HIP-event duration is then the kernel's own time.  (In the production
        # graph the weight-gradient kernels share the CUs with the data-gradient chain; the profiler's per-kernel times
        # of that graph are longer for both.  profiles/*_serial_kernel_stats.csv is the rocprofv3 run of this mode.)
        was_overlap, eng.overlap = eng.overlap, False
        for _ in range(n_inst):
            eager.run()
        agg = timer.summary()
        eng.overlap = was_overlap
        for obj, name, orig in restore:
            setattr(obj, name, orig)
        fam = max(agg.items(), key=lambda kv: kv[1][0])
        label, (ms, fl, cnt, issued) = fam
        tfl = fl / (ms * 1e-3) / 1e12
        traffic, tsrc = load_traffic(label, B, args.phase)
        # fp32 and bf16 issued flops do not add up to one rate: with both gather kernels in the step the family-level
        # issued rate is left out and per_entry_point carries one per pipe
        mixed = "hmvae_gather_gemm_v4" in timer.per_kernel and "hmvae_gather_gemm_v2" in timer.per_kernel
        peff = effective_peak(timer.pipe_ms[label])
        out["roofline"] = {"kernel": label, "bound": "mfma", "achieved": tfl, "peak": PEAK_F32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": tfl / PEAK_F32_MFMA_TFLOPS,
                           "peak_effective": peff, "frac_effective": tfl / peff,
                           "peak_effective_note": "time-weighted over the family's launches: %.1f TFLOP/s fp32-equivalent for "
                                                  "launches on the bf16 pipe (three planes, six MFMAs per block: 2516.6 / 6), "
                                                  "157.3 for fp32-MFMA launches; `frac` stays against 157.3 (SURVEY 8(d)), so "
                                                  "a frac above 1 would not mean skipped work" % PEAK_BF16X3_TFLOPS,
                           "ms_by_pipe_per_step": {k: v / n_inst for k, v in timer.pipe_ms[label].items()},
                           "traffic": traffic,
                           "traffic_unit": "HBM-side bytes per launch (PMC)", "traffic_source": tsrc,
                           "launches_per_step": cnt // n_inst, "avg_launch_us": ms / cnt * 1e3,
                           "timing": "HIP events on the launch stream around each C-ABI launch call of the family, %d extra "
                                     "eager one-stream steps of the same workload" % n_inst,
                           "per_entry_point": {k: dict({"ms_per_step": v[0] / n_inst, "launches_per_step": v[1] // n_inst,
                                                        "algorithmic_TFLOP/s": v[2] / (v[0] * 1e-3) / 1e12},
                                                       **({"issued_bf16_TFLOP/s": v[3] / (v[0] * 1e-3) / 1e12,
                                                           "issued_frac_of_bf16_peak": v[3] / (v[0] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
                                                          if k == "hmvae_gather_gemm_v4" else
                                                          {"issued_fp32_TFLOP/s": v[3] / (v[0] * 1e-3) / 1e12,
                                                           "issued_frac_of_fp32_peak": v[3] / (v[0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
                                                          if k.startswith("hmvae_gather") else {}))
                                               for k, v in timer.per_kernel.items()},
                           "flops_per_step_effective": fl / n_inst,
                           "flops_per_step_issued": issued / n_inst,
                           "issued_TFLOP/s": None if mixed else issued / (ms * 1e-3) / 1e12,
                           "note": "achieved = algorithmic (fp32, mask-effective) flops of the convolutions / kernel time, "
                                   "priced against the fp32 MFMA peak the reference's arithmetic would need.  The family mixes "
                                   "two pipes: hmvae_gather_gemm_v2 issues fp32 MFMAs (fewer flops than algorithmic where "
                                   "taps are folded, more where 16-row tiles are padded), hmvae_gather_gemm_v4 issues six "
                                   "bf16 MFMAs per fp32-equivalent block (three exact bf16 planes per operand); "
                                   "per_entry_point prices each against its own pipe's dense peak (the family-level issued rate "
                                   "is null while both run: the two kinds of flops do not add)",
                           "all_kernels": {k: {"ms_per_step": v[0] / n_inst, "TFLOP/s": v[1] / (v[0] * 1e-3) / 1e12,
                                               "frac": v[1] / (v[0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                               "issued_TFLOP/s": (None if (mixed and k.startswith("gather")) else
                                                                  v[3] / (v[0] * 1e-3) / 1e12),
                                               "peak_effective": effective_peak(timer.pipe_ms[k]),
                                               "frac_effective": v[1] / (v[0] * 1e-3) / 1e12 / effective_peak(timer.pipe_ms[k]),
                                               "launches_per_step": v[2] // n_inst}
                                           for k, v in agg.items()}}
        conv_flops_step = 3.0 * eng.flops_per_seq * B
        out["roofline"]["step_level"] = {"conv_flops_per_step_effective": conv_flops_step,
                                         "achieved_TFLOP/s": conv_flops_step / (ms_per_step * 1e-3) / 1e12,
                                         "frac": conv_flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    if rank == 0 and world == 1 and not args.no_fk:
        out["fk_roofline"] = fk_microbench(device)
    if rank == 0 and world == 1 and not args.no_extra:
        out["sampling"] = sampling_bench(tr, cfg, device)
        out["traj"] = traj_bench(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
