"""Runs the graph-destroy abort experiments one after another on the GPU box, each as a child process under a timeout,
logs under gpurun_out/abort/.  A child that aborts (glibc heap check) is a RESULT, the study goes on; a child that times
out ends the study (no further GPU step after a hang).

    python tools/repro/abort_study.py [names...]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out", "abort")
os.makedirs(OUT, exist_ok=True)
PY = sys.executable
TUNER = [PY, os.path.join(ROOT, "tools", "tune_in_situ.py"), "--config", os.path.join(ROOT, "configs", "trajectory_model.yaml"),
         "--batch", "8", "--passes", "1", "--steps", "30", "--out", os.path.join(OUT, "tuned.json")]
GDB = ["/opt/rocm/bin/rocgdb", "-batch", "-ex", "set pagination off", "-ex", "run", "-ex", "bt 40", "--args"]
REPRO = os.path.join(ROOT, "tools", "repro", "graph_destroy_repro")
TORCH = [PY, os.path.join(ROOT, "tools", "repro", "graph_destroy_torch.py")]
CHECK = {"MALLOC_CHECK_": "3", "TUNE_VERBOSE": "1"}
EXPERIMENTS = [
    ("hip_mode0", [REPRO, "0", "200"], CHECK, 300),
    ("hip_mode1", [REPRO, "1", "200"], CHECK, 300),
    ("torch_mode0", TORCH + ["0", "100"], CHECK, 300),
    ("torch_mode1", TORCH + ["1", "100"], CHECK, 300),
    ("tuner_destroy_joinall_gdb", GDB + TUNER, dict(CHECK, HMVAE_GRAPH_PARK="0", HMVAE_JOIN_ALL="1"), 900),
    ("tuner_destroy_joinforked", TUNER, dict(CHECK, HMVAE_GRAPH_PARK="0", HMVAE_JOIN_ALL="0"), 900),
]


def main():
    want = set(sys.argv[1:])
    for name, cmd, env, limit in EXPERIMENTS:
        if want and name not in want:
            continue
        t0 = time.time()
        log = os.path.join(OUT, name + ".log")
        with open(log, "w") as f:
            try:
                rc = subprocess.run(cmd, env=dict(os.environ, **env), stdout=f, stderr=subprocess.STDOUT, timeout=limit,
                                    cwd=ROOT).returncode
            except subprocess.TimeoutExpired:
                rc = "timeout"
        print("%-28s rc=%s  %.0f s" % (name, rc, time.time() - t0), flush=True)
        with open(os.path.join(OUT, "summary.txt"), "a") as f:
            f.write("%s rc=%s %.0fs\n" % (name, rc, time.time() - t0))
        if rc == "timeout":
            print("timeout: study ends here")
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
