"""Dev tool (GPU box): the sixteen gather-GEMM launches of the VAE step at B=256 for library variants (HMVAE_LIB=...),
each with a fresh autotune; prints forward / backward-data totals.  One process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from hm_vae_amd.trainer_motion_vae import Trainer
    from hm_vae_amd.utils_common import get_config
    cfg = get_config(os.path.join(ROOT, "configs", "len64_no_aug_hm_vae.yaml"))
    tr = Trainer(cfg); tr.cuda(); eng = tr.model.engine(); B = 256
    w = eng.workspace(B); bp = w["bp"]; dev = eng.device
    eng.pack_weights()
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return sorted(x.elapsed_time(y) for x, y in evs)[len(evs) // 2] * 1e3
    names = ["enc%d" % i for i in range(4)] + ["dec%d" % i for i in range(4)]
    tot = {"f": 0.0, "b": 0.0}; parts = []
    for name, op in zip(names, eng.enc_conv + eng.dec_conv):
        for tag, sd in (("f", op.fwd), ("b", op.bwd)):
            if sd is None:
                continue
            rows_src = int(sd.chan_rows.max().item()) + 70 * max(sd.src_ts, 1) + 8
            rows_dst = int(sd.dst_rows.max().item()) + 70 * max(sd.dst_ts, 1) + 8
            src = torch.randn(rows_src, bp, device=dev); dst = torch.zeros(rows_dst, bp, device=dev)
            t = timeit(lambda: sd.run(src, dst, bp))
            tot[tag] += t; parts.append("%s%s %.0f%s" % (name, tag, t, str(sd._tuned.get(bp))))
    print(os.path.basename(os.environ.get("HMVAE_LIB", "default")), "fwd %.0f us  bwd %.0f us  sum %.0f | %s" % (tot["f"], tot["b"], tot["f"] + tot["b"], " ".join(parts)), flush=True)
else:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, HMVAE_LIB=os.path.join(ROOT, "hm_vae_amd", "lib", lib), HMVAE_TUNE_DEFAULTS="0"))
