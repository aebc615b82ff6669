"""Test-time optimisation loop throughput: iterations/s and windows/s of LatentRefiner on synthetic
targets, the two phases timed separately, with the CPU oracle's iteration (torch autograd, host threads) beside it.

    python tools/bench_refine.py [--windows 64] [--iters 60] [--cpu-iters 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "len_64_test_interpolation.yaml"))
    ap.add_argument("--windows", type=int, default=64)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--cpu-iters", type=int, default=2)
    a = ap.parse_args()
    from hm_vae_amd import my_tools, refine
    from hm_vae_amd.trainer_motion_vae import Trainer
    cfg = yaml.safe_load(open(a.config))
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = Trainer(cfg)
    tr.cuda()
    B, T = a.windows, cfg["train_seq_len"]
    target = my_tools.rotation_matrix_from_ortho6d(torch.randn(B * T * 24, 6, device=dev)).reshape(B, T, 24, 3, 3)
    mask = torch.from_numpy(refine.temporal_mask(T, cfg["interpolation_window"]))[None, :, None].repeat(B, 1, 24)
    out = {"windows": B, "T": T}
    for phase, after in (("latent", 10 ** 9), ("decoder", -1)):
        refine.DECODER_PHASE_AFTER = after
        ref = refine.LatentRefiner(tr.model, cfg, target, mask)
        for _ in range(5):
            ref.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            ref.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        ref.restore()
        out[phase] = {"ms_per_iter": dt * 1e3, "window_iters_per_s": B / dt}
    if a.cpu_iters:                               # the CPU leg: the oracle's restatement of one iteration
        from oracle import hmvae_oracle as O
        target, mask = target.cpu(), mask.cpu()
        offs = tr.model.fk_layer.positions[0].cpu().numpy()
        sd = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
        orc = O.OracleVAE(cfg, sd, offsets=offs)
        plan = orc.plan
        bc = min(B, 8)
        z0 = torch.randn(bc, plan.edge_num[1], plan.lat_dim[0]).requires_grad_(True)
        zd = torch.randn(bc, plan.edge_num[plan.L], plan.lat_dim[plan.L - 1]).requires_grad_(True)
        orc.refine_forward(z0, zd, target[:bc], mask[:bc])
        t0 = time.perf_counter()
        for _ in range(a.cpu_iters):
            orc.refine_forward(z0, zd, target[:bc], mask[:bc])
        dt = (time.perf_counter() - t0) / a.cpu_iters
        out["cpu_oracle"] = {"windows": bc, "ms_per_iter": dt * 1e3, "window_iters_per_s": bc / dt,
                             "threads": torch.get_num_threads()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
