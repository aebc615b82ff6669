"""Dev tool (GPU box): the config-4 FK / tail kernels at 262 144 frames (bench.py's fk_microbench) as a table."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
r = bench.fk_microbench(torch.device("cuda:0"), iters=30)
print("%-34s %8.1f us  frac %.3f" % (r["kernel"], r["avg_us"], r["frac"]))
w = r["with_6d_copy"]
print("%-34s %8.1f us  frac %.3f (1728 B)  %.3f (2304 B moved)" % (w["kernel"], w["avg_us"], w["frac"], w["frac_moved"]))
for k in ("fk_only_aos", "fk_rot6d_aos", "fk_bwd_aos", "tail_loss"):
    e = r[k]
    print("%-34s %8.1f us  frac %.3f" % (e["kernel"], e["avg_us"], e["frac"]))
