"""The config-4 FK kernels on their own (for rocprofv3 passes): a few launches of every variant bench.py's fk_roofline
reports, at 4096 x 64 = 262 144 frames.   python3 tools/fk_profile.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
out = bench.fk_microbench(torch.device("cuda:0"), iters=10)
print({k: (round(v["avg_us"], 1) if isinstance(v, dict) else v) for k, v in out.items() if isinstance(v, dict) or k == "avg_us"})
