"""Batch assembly: hmvae_motion_batch on a device-resident synthetic dataset vs the CPU oracle's
per-item path.  Prints one JSON line.   python tools/bench_loader.py [--batch 256] [--seq-len 64] [--rot]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=64)
    ap.add_argument("--rot", action="store_true")
    a = ap.parse_args()
    from hm_vae_amd.utils_motion_vae import DeviceMotionDataset, DeviceMotionLoader
    rng = np.random.default_rng(0)
    seqs = [rng.standard_normal((int(n), 579)).astype(np.float32) for n in rng.integers(200, 2000, size=400)]
    dev = torch.device("cuda:0")
    ds = DeviceMotionDataset(seqs, dev)
    B, T = a.batch, a.seq_len
    ld = DeviceMotionLoader(ds, B, T, fps_aug=True, root_rot=a.rot, seed=1)
    choices = [ld.sample_clip(i % len(seqs)) for i in range(B)]
    clips = ld.describe(choices)
    for _ in range(3):
        ds.assemble(clips, T, a.rot)
    evs = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ds.assemble(clips, T, a.rot)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    dev_us = sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2] * 1e3      # descriptor copy + kernel
    t0 = time.perf_counter()
    n_it = 0
    for batch in ld:                                                              # host sampling + descriptors + launch
        n_it += 1
    torch.cuda.synchronize()
    epoch_s = time.perf_counter() - t0
    by = B * T * (579 + 651) * 4
    out = {"metric": "batch_assembly", "batch": B, "seq_len": T, "rot": a.rot, "device_us_per_batch": dev_us,
           "algorithmic_GBps": by / dev_us / 1e3, "frac_hbm_peak": by / dev_us / 1e3 / 8000.0,
           "loader_batches_per_s_incl_host_sampling": n_it / epoch_s, "dataset_rows": ds.n_rows}
    from oracle import hmvae_oracle as O
    ms = np.load(os.path.join(ROOT, "hm_vae_amd", "assets", "all_amass_data_mean_std.npy"))
    ms[1, ms[1] == 0] = 1.0
    t0 = time.perf_counter()
    for seq, stride, crop, M in choices[:64]:
        O.motion_item(seqs[seq], ms, T, crop, stride, M)
    out["cpu_oracle_items_per_s"] = 64 / (time.perf_counter() - t0)
    out["device_items_per_s"] = B / (dev_us * 1e-6)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
