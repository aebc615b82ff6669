"""Build libhmvae_hip.so (gfx950) in-tree with hipcc.  `python -m hm_vae_amd.build`."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhmvae_hip.so")
SOURCES = ["gemm_kernels.hip", "gemm_v2.hip", "gemm_v3.hip", "gemm_v4.hip", "wgrad_lds.hip", "wgrad_bf3.hip", "wgrad_pq.hip", "elementwise_kernels.hip", "latent_bridge.hip", "fk_kernels.hip"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = ["-DHMVAE_STAMP"] if os.environ.get("HMVAE_STAMP") else []      # diagnostic build: in-kernel s_memtime stamps
    if os.environ.get("HMVAE_ABLATE"):                                        # diagnostic build: timing ablations of the v3 kernel
        extra.append("-DHMVAE_ABLATE")
    if os.environ.get("HMVAE_CXXFLAGS"):                                      # experiments: extra -D flags for every file
        extra += os.environ["HMVAE_CXXFLAGS"].split()
    if extra:
        force = True
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "pack_taps_inl.h"), os.path.join(CSRC, "wreduce_inl.h"), os.path.join(ROOT, "include", "hmvae.h")]
    if not force and not _newer(LIB_PATH, deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s).replace(".hip", ".o"))
        if force or _newer(o, [s] + deps[len(srcs):]):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                   "-c", s, "-o", o] + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
