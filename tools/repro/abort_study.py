"""HISTORICAL (round 2): the driver that ran the graph-destroy reproducers many times to locate the fault in the bundled HIP
7.0.2 runtime.  The cause is established (tools/repro/README.md) and the product parks graphs on that runtime; do NOT re-run this on
the pool -- it makes a known crash happen again.  Kept as a record of how the evidence was gathered."""
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
EXPERIMENTS = []
TORCH_LIB = os.path.join(os.path.dirname(__import__("torch").__file__), "lib")       # the ROCm 7.0.2 runtime torch bundles
OLD = dict(CHECK, LD_LIBRARY_PATH=TORCH_LIB)
NEW = dict(CHECK, LD_PRELOAD="/opt/rocm/lib/libamdhip64.so.7")                         # the image's ROCm 7.2.0 runtime
for rep in range(6):
    for ja, rot in ((0, 0), (0, 1)):
        EXPERIMENTS.append(("hip_rt702_j%d_r%d_rep%d" % (ja, rot, rep), [REPRO, str(ja), str(rot), "1", "1", "1", "300"], OLD, 120))
for rep in range(6):
    EXPERIMENTS.append(("torch_rt720_j0_r1_rep%d" % rep, TORCH + ["0", "1", "0", "0", "100"], NEW, 300))
    EXPERIMENTS.append(("torch_rt720_j0_r0_rep%d" % rep, TORCH + ["0", "0", "0", "0", "100"], NEW, 300))


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
