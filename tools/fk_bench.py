"""FK / decode-tail microbench only (the fk_roofline object of bench.py)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
print(json.dumps(bench.fk_microbench(torch.device("cuda:0"), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 30)))
