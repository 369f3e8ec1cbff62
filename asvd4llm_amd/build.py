"""Build libasvd_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a plain C-ABI
shared object (include/asvd_hip.h) loaded with ctypes (asvd4llm_amd/_lib.py)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["svd_jacobi.hip", "aux_kernels.hip", "sigma_max.hip", "comm.hip", "lowrank_forward.hip"]
LIB = os.path.join(HERE, "libasvd_hip.so")


def _newest_source_mtime():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "asvd_hip.h")]
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
