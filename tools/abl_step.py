"""Dev tool (GPU box): what a group of launches costs IN the step -- the hipGraph step is captured without them (wrong
results, timing only) and timed against the complete one.

    python tools/abl_step.py bridge        # the latent bridge: heads, reparametrisation, their data gradients
    python tools/abl_step.py heads_wgrad   # the heads' weight gradients as well
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "bridge"
    import bench
    from hm_vae_amd import ops
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    from hm_vae_amd.graph_step import GraphedStep, retire
    dev = torch.device("cuda:0")
    cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
    torch.manual_seed(1234)
    tr = Trainer(cfg)
    tr.cuda()
    B, T = 256, cfg["train_seq_len"]
    data = bench.synthetic_device_batch(B, T, seed=0, device=dev)
    gs = GraphedStep(tr, cfg, data, cfg["iteration_interval"], use_graph=True)
    eng = tr.model.engine()

    def measure(steps=100):
        if gs.graphs is not None:
            retire(*gs.graphs)
            retire(gs.g1b)
        gs.graphs = None
        for _ in range(10):
            gs.run(lag=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gs.run(lag=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    base = [measure(), measure()]
    cap = torch.cuda.is_current_stream_capturing

    def gate(fn, ret=None):
        def wrapped(*a, **k):
            if cap():
                return ret
            return fn(*a, **k)
        return wrapped

    L = eng.p.L
    lins = [eng.enc_lin[i] for i in (0, L - 1)] + [eng.dec_lin[i] for i in (0, L - 1)]
    for op in lins:
        op.forward = gate(op.forward)
        op.backward_data = gate(op.backward_data, False)
        if what == "heads_wgrad":
            op.backward_weight = gate(op.backward_weight)
    ops.reparam_kl_fwd = gate(ops.reparam_kl_fwd)
    ops.reparam_kl_bwd = gate(ops.reparam_kl_bwd)
    abl = [measure(), measure()]
    print("complete step %.4f / %.4f ms, without the %s launches %.4f / %.4f ms" % (base[0], base[1], what, abl[0], abl[1]))


if __name__ == "__main__":
    main()
