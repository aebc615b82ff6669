"""Choose kernel variants IN the training step (GPU box).

    python tools/tune_in_situ.py [--config configs/len64_no_aug_hm_vae.yaml] [--batch 256] [--out hm_vae_amd/assets/tune_mi355x.json]

Start from the autotuner's alone-timed picks, then coordinate descent: one knob of one layer at a time -- the gather
kernels' (column units, K split), the weight-gradient GEMMs' (split, direct / LDS-staged) -- the whole hipGraph step is
re-captured and timed, a candidate is kept when it beats the incumbent by more than the measurement noise.  The result
is written in the autotune-cache format (keys name shape and batch) and merged into the output file.

On the HIP runtime bundled with the torch wheel (7.0.2) replaced graphs are parked, not destroyed (graph_step.py), and with
many hundred parked graphs replays slow down (seen once: 1.81 -> 3.96 ms/step at the start of a second pass after ~700
graphs).  Run the tuner on the image's 7.2.0 runtime instead, where graphs are destroyed normally:
    LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 python tools/tune_in_situ.py ...  HMVAE_FORCE_SPLIT=1 tunes the four-graph DDP step: starting from the shipped choices it found < 1 % (1.825 ->
1.809 ms), so there is no separate file for it.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HMVAE_TUNE_DEFAULTS", "0")        # start from a fresh alone-timed autotune
VERBOSE = os.environ.get("TUNE_VERBOSE", "0") == "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--phase", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--seq-len", type=int, default=0, help="override train_seq_len (trajectory model: config 5 quotes 64)")
    ap.add_argument("--fast", action="store_true", help="weight-gradient splits 1, 2, 3, 4, 6, 8 only")
    ap.add_argument("--no-v4", action="store_true", help="leave the v4 (bf16-plane) gather choices as the autotuner made them")
    ap.add_argument("--only", default="", help="comma list of knob kinds to walk: g, g4, w, p = only the weight gradients' "
                    "place in the step (default: all)")
    ap.add_argument("--match", default="", help="walk only the knobs whose tune key contains this")
    ap.add_argument("--out", default=os.path.join(ROOT, "hm_vae_amd", "assets", "tune_mi355x.json"))
    a = ap.parse_args()
    import bench
    from hm_vae_amd import ops
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from hm_vae_amd.graph_step import GraphedStep, retire
    dev = torch.device("cuda:0")
    ops.KEEP_IO = True
    cfg = get_config(a.config)
    if a.seq_len:
        cfg["train_seq_len"] = a.seq_len
    torch.manual_seed(1234)
    tr = Trainer(cfg)
    tr.cuda()
    B, T = a.batch, cfg["train_seq_len"]
    is_vae = cfg["model_name"] == "TwoHierSAVAEModel"
    if is_vae:
        data = bench.synthetic_device_batch(B, T, seed=0, device=dev)
        it0 = cfg["iteration_interval"] if a.phase == 2 else 0
    else:
        from hm_vae_amd.utils_motion_vae import SyntheticMotionLoader
        data = next(iter(SyntheticMotionLoader(B, T, dev, 1, 0, with_positions=True)))
        it0 = 0
    gs = GraphedStep(tr, cfg, data, it0, use_graph=True)
    for _ in range(5):
        gs.run()
    eng = tr.model.engine() if is_vae else tr.model.engine(T)
    bp = eng.workspace(B)["bp"]

    def measure():
        if gs.graphs is not None:              # never destroyed: see graph_step.retire
            retire(*gs.graphs)
            retire(gs.g1b)
        gs.graphs = None                       # re-capture with the current choices
        for _ in range(4):
            gs.run(lag=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            gs.run(lag=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    if is_vae:
        all_ops = eng.enc_conv + eng.dec_conv + [eng.enc_lin[i] for i in (0, eng.p.L - 1)] + [eng.dec_lin[i] for i in (0, eng.p.L - 1)]
    else:
        all_ops = eng.conv + [eng.head]
    knobs = []
    for op in all_ops:
        for sd in (op.fwd, op.bwd):
            if sd is not None and bp in sd._tuned:
                knobs.append(("g", sd))
                if sd._v4_args is not None and ops.USE_V4 != "0" and not a.no_v4:
                    knobs.append(("g4", sd))
        if bp in op._split_tuned:
            knobs.append(("w", op))
    base = min(measure(), measure())
    noise = 0.002 * base
    print("start: %.4f ms/step (%.0f seq/s), %d knobs" % (base, B / base * 1e3, len(knobs)), flush=True)
    for ps in range(a.passes):
        improved = False
        for kind, obj in knobs:
            only = a.only.split(",") if a.only else []
            if a.match and a.match not in (obj.wg_tune_key(bp) if kind == "w" else obj.tune_key(bp)):
                continue
            place_only = kind == "w" and "p" in only and "w" not in only      # "p": only the weight gradients' place in the step
            if only and kind not in only and not place_only:
                continue
            if place_only and (getattr(obj, "fwd", None) is None or not hasattr(obj, "t_out")):
                continue                         # (convolutions only: the engine tags their launches)
            base = measure()                 # the incumbent, re-timed: clocks drift over a run of several minutes
            noise = 0.002 * base
            if kind == "g4":
                # the bf16-plane kernel's launch shape (rows per group, columns per unit, K split) or None = the v2 / v3
                # pick: every shape is timed alone first (cheap), the three best and "off" are tried in the step
                cur = obj._tuned_v4.get(bp)
                src_dst = getattr(obj, "_last_io", None)
                cands = []
                for mtw in (1, 2):
                    sd4 = obj.v4_side(mtw)
                    if sd4 is None:
                        continue
                    for nc in (128, 64):
                        if bp % nc:
                            continue
                        for sp in sd4.v4_splits(bp, nc):
                            cands.append((mtw, nc, sp))
                if src_dst is not None and len(cands) > 3:
                    timed = []
                    for c in cands:
                        sd4 = obj.v4_side(c[0])
                        ops.MultiPack(dev).run([(sd4, obj._last_dense)])
                        timed.append((obj._time(lambda: sd4.run_v4(src_dst[0], src_dst[2], bp, None, 0, 0.2, c[1], c[2])), c))
                    cands = [c for _, c in sorted(timed)[:3]]
                cands = [c for c in cands + [None] if c != cur]
                best_c, best_t = cur, base
                for c in cands:
                    if VERBOSE:
                        print("    try", obj.tune_key(bp) + "_v4", c, flush=True)
                    if c is not None and c[2] > 1:
                        obj.v4_side(c[0])._v4_buffers(bp, c[1], c[2])
                    obj._tuned_v4[bp] = c
                    ops.MultiPack.refresh(obj)
                    t = measure()
                    if t < best_t - noise:
                        best_c, best_t = c, t
                obj._tuned_v4[bp] = best_c
                ops.MultiPack.refresh(obj)
                label = obj.tune_key(bp) + "_v4"
            elif kind == "g":
                cur = tuple(obj._tuned[bp])
                v2_now = tuple(obj._tuned_v2[bp])
                cands = [tuple(c) for c in obj.v2_candidates(bp)]
                if ops.USE_V3 and obj.n_pos >= 4 and obj.g.kt > 1:      # the window kernel where its plan exists
                    for chunk, ts in ((8, 1), (4, 2)):
                        if obj.window_plan(chunk, ts) is not None:
                            cands.append(("v3", chunk, ts) + v2_now)
                cands = [c for c in cands if c != cur]
                best_c, best_t = cur, base
                for c in cands:
                    if VERBOSE:
                        print("    try", obj.tune_key(bp), c, flush=True)
                    obj._tuned[bp] = c
                    if c[0] != "v3":
                        obj._tuned_v2[bp] = c
                    try:
                        t = measure()
                    except Exception as e:                           # a shape the kernel refuses: skip
                        print("    skip", obj.tune_key(bp), c, type(e).__name__, flush=True)
                        obj._tuned[bp] = cur
                        obj._tuned_v2[bp] = v2_now
                        gs.graphs = None
                        continue
                    if t < best_t - noise:
                        best_c, best_t = c, t
                obj._tuned[bp] = best_c
                obj._tuned_v2[bp] = tuple(best_c[3:5]) if best_c[0] == "v3" else tuple(best_c[:2])
                label = obj.tune_key(bp)
            else:
                cur = (obj._split_tuned[bp], obj._mode_tuned.get(bp, 0))
                place0 = (bool(obj._pq_tuned.get(bp, False)), int(obj._delay_tuned.get(bp, 0)))
                modes = [0] + ([1] if obj.lds_eligible() else []) + ([2] if (ops.WGRAD_BF3 and obj.bf3_eligible()) else []) + \
                    ([3] if (ops.WGRAD_BF3 and obj.lds_eligible() and obj.pair_eligible()) else [])
                splits = [c for c in obj.split_candidates(bp) if not a.fast or c in (1, 2, 3, 4, 6, 8)]
                cands = [(c, m) for m in modes for c in splits if (c, m) != cur]
                if place_only:
                    cands = []
                if not place_only and ops.WGRAD_BF3 and obj.lds_eligible() and obj.wide() is not None:      # wide tiles: their own short list of splits
                    cands += [(c, 4) for c in obj.wide_split_candidates(bp) if c in obj.split_candidates(bp) and (c, 4) != cur]
                best_c, best_t = cur, base
                for c, m in cands:
                    if VERBOSE:
                        print("    try", obj.wg_tune_key(bp), (c, m), flush=True)
                    obj._split_tuned[bp], obj._mode_tuned[bp] = c, m
                    t = measure()
                    if t < best_t - noise:
                        best_c, best_t = (c, m), t
                obj._split_tuned[bp], obj._mode_tuned[bp] = best_c
                # the position-pair launch and the launch's place in the step (beside / behind the layer's data gradient)
                best_p = place0
                for pl in ((False, 0), (False, 1), (True, 0), (True, 1), (True, 2)):
                    if pl == place0 or (pl[0] and obj.pq() is None):
                        continue
                    obj._pq_tuned[bp], obj._delay_tuned[bp] = pl
                    t = measure()
                    if VERBOSE:
                        print("    try", obj.wg_tune_key(bp), "pq/delay", pl, "%.4f" % t, flush=True)
                    if t < best_t - noise:
                        best_p, best_t = pl, t
                obj._pq_tuned[bp], obj._delay_tuned[bp] = best_p
                if best_p != place0:
                    best_c = tuple(best_c) + ("pq%d delay%d" % best_p,)
                label = obj.wg_tune_key(bp)
            if best_t < base - noise:
                print("  %-58s %s -> %s   %.4f -> %.4f ms" % (label, cur, best_c, base, best_t), flush=True)
                base = best_t
                improved = True
        base = min(measure(), measure())
        print("pass %d: %.4f ms/step (%.0f seq/s)" % (ps, base, B / base * 1e3), flush=True)
        if not improved:
            break
    out = {}
    if os.path.exists(a.out):
        try:
            out = json.load(open(a.out))
        except ValueError:
            out = {}
    for kind, obj in knobs:
        if kind == "g":
            out[obj.tune_key(bp)] = list(obj._tuned[bp])
        elif kind == "g4":
            c = obj._tuned_v4.get(bp)
            out[obj.tune_key(bp) + "_v4"] = list(c) if c else [0]
        else:
            out[obj.wg_tune_key(bp)] = [obj._split_tuned[bp], 0, int(obj._mode_tuned.get(bp, 0)),
                                        int(bool(obj._pq_tuned.get(bp, False))), int(obj._delay_tuned.get(bp, 0))]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=0, sort_keys=True)
    print("wrote %d entries to %s" % (len(out), a.out), flush=True)


if __name__ == "__main__":
    main()
