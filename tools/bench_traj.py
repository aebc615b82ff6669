"""TrajectoryModel step timing (not the BASELINE metric, so not part of bench.py's line).

    python tools/bench_traj.py [--batch 8] [--steps 30] [--warmup 5] [--cpu]

Prints one JSON line: sequences/s of Trainer.gen_update (forward + losses + backward + Adam) on synthetic batches resident
in HBM, the mask-effective conv TFLOP/s that corresponds to, and with --cpu the CPU oracle's rate on the same step.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--seq-len", type=int, default=0, help="override train_seq_len (BASELINE config 5 quotes len-64)")
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--graph", type=int, default=1, help="1: hipGraph replay of the step (GraphedStep), 0: eager launches")
    a = ap.parse_args()
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from hm_vae_amd.utils_motion_vae import SyntheticMotionLoader
    cfg = get_config(os.path.join(ROOT, "configs", "trajectory_model.yaml"))
    cfg["batch_size"] = a.batch
    if a.seq_len:
        cfg["train_seq_len"] = a.seq_len
    dev = torch.device("cuda:0")
    tr = Trainer(cfg)
    tr.cuda()
    T = cfg["train_seq_len"]
    batches = list(SyntheticMotionLoader(a.batch, T, dev, 4, 0, with_positions=True))
    from hm_vae_amd.graph_step import GraphedStep
    gs = GraphedStep(tr, cfg, batches[0], 0, use_graph=bool(a.graph))
    for i in range(a.warmup):
        gs.set_batch(batches[i % 4])
        gs.run(lag=bool(a.graph))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        gs.set_batch(batches[i % 4])
        gs.run(lag=bool(a.graph))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    eng = tr.model.engine(T)
    out = {"metric": "trajectory_train_sequences_per_sec", "value": a.batch / dt, "ms_per_step": dt * 1e3,
           "batch": a.batch, "seq_len": T, "dtype": "f32", "graph": bool(a.graph),
           "algorithmic_tflops": 3 * eng.flops_per_seq * a.batch / dt / 1e12}
    if a.cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import hmvae_oracle as O
        import numpy as np
        plan = O.TrajPlan(cfg)
        orc = O.OracleTrajectory(cfg, O.traj_name_seeded_params(plan), np.zeros(3), np.ones(3))
        opt, _ = O.make_adam(orc, cfg)
        data = O.synthetic_traj_batch(a.batch, T, seed=1)
        orc.train_forward(data)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 10.0:
            opt.zero_grad()
            orc.train_forward(data)
            opt.step()
            n += 1
        out["cpu_oracle_seq_per_s"] = n * a.batch / (time.perf_counter() - t0)
        out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
