"""Build libasvd_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a plain C-ABI
shared object (include/asvd_hip.h) loaded with ctypes (asvd4llm_amd/_lib.py)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["svd_jacobi.hip", "evd_wave.hip", "aux_kernels.hip", "sigma_max.hip", "comm.hip", "lowrank_forward.hip"]
LIB = os.path.join(HERE, "libasvd_hip.so")


def _newest_source_mtime():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "asvd_hip.h")]
    return max(os.path.getmtime(p) for p in paths)


# per-source extra flags.  evd_wave.hip: the SLP vectoriser would pack the per-register FMAs of the wave-local eigen-solver into
# v_pk_fma_f32, which cannot take the DPP operand its column rotations live on (csrc/jacobi_shared.h)
EXTRA_FLAGS = {"evd_wave.hip": ["-fno-slp-vectorize"]}


def build(force=False, verbose=True, extra_flags=None, out=None):
    """extra_flags / out: MEASUREMENT builds only (tools/bench_supgram.py: -DASVD_SG_TIMING into libasvd_hip_meas.so) — the product library is
    always built without extra flags into libasvd_hip.so."""
    lib_path = out or LIB
    if not force and os.path.exists(lib_path) and os.path.getmtime(lib_path) >= _newest_source_mtime():
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" if not out else "build_" + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + list(extra_flags or [])
    procs, objs = [], []
    for src in SOURCES:  # one hipcc per translation unit, all at once
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, f"hipcc -c {failed}")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv)
