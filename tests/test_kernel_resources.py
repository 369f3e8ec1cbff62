"""No kernel of libasvd_hip may use scratch memory (VGPR spills / indexed local arrays).  Measured on MI355X / ROCm 7.2 (DESIGN.md 3.8 a):
a kernel with spills gave nondeterministic corruption as soon as kernels of several streams were in flight.  hipcc cross-compiles
without a GPU; the compiler's own resource remarks are the evidence."""
import os
import re
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asvd4llm_amd", "csrc")


@pytest.mark.parametrize("src", ["svd_jacobi.hip", "evd_wave.hip", "aux_kernels.hip", "sigma_max.hip", "lowrank_forward.hip"])
def test_no_kernel_uses_scratch(src, tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    from asvd4llm_amd.build import EXTRA_FLAGS  # the flags the library is built with
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only"] + EXTRA_FLAGS.get(src, []) +
                         ["-c", os.path.join(CSRC, src), "-o", str(tmp_path / "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    assert len(names) == len(scratch) and len(names) >= 1
    bad = [(n, s) for n, s in zip(names, scratch) if s != 0]
    assert not bad, bad
