"""The C-ABI library builds for gfx950, loads on a CPU-only box and exports every symbol include/asvd_hip.h declares.
No compute calls here (no GPU); only host-side size queries and argument validation."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def declared_functions():
    src = open(os.path.join(ROOT, "include", "asvd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(asvd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    from asvd4llm_amd import _lib
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in asvd_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in _lib.SIGNATURES"
    assert sorted(_lib.SIGNATURES) == names


def test_no_torch_types_in_abi():
    src = open(os.path.join(ROOT, "include", "asvd_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # comments may cite the torch calls an entry point replaces; declarations may not
    assert "torch" not in code.lower() and "tensor" not in code.lower()
    assert "at::" not in src and "c10::" not in src and "#include <torch" not in src
    # every parameter type of every prototype is a plain C type
    for proto in re.findall(r"\basvd_[a-z0-9_]+\s*\(([^)]*)\)", code):
        for prm in [p.strip() for p in proto.split(",") if p.strip() and p.strip() != "void"]:
            assert re.match(r"^(const\s+)?(void|int|float|double|char|size_t|int64_t|long long)\b[\s\*const]*\w*$", prm), prm


def test_host_side_queries(built_lib):
    lib = built_lib
    assert lib.asvd_version() >= 100
    assert lib.asvd_status_string(0).decode() == "ok"
    nb = ctypes.c_size_t()
    assert lib.asvd_svd_worksize(1, 4096, 4096, 1, ctypes.byref(nb)) == 0
    # panels: (4096 + 4096) rows x 4096 cols fp32 = 128 MiB plus small buffers
    # direct path: panels + packed original (192 MiB); reduction path: panels 64 + fp64 Gram/scaled Gram 2x128 + R 64 + right
    # vectors 2x64 + the inner square problem's own workspace
    assert 192 * 2**20 <= nb.value <= 1024 * 2**20
    nb2 = ctypes.c_size_t()
    assert lib.asvd_svd_worksize(1, 4096, 11008, 1, ctypes.byref(nb2)) == 0  # wide: oriented internally
    nb3 = ctypes.c_size_t()
    assert lib.asvd_svd_worksize(1, 11008, 4096, 1, ctypes.byref(nb3)) == 0
    assert nb2.value == nb3.value
    assert lib.asvd_svd_worksize(0, 4, 4, 1, ctypes.byref(nb)) == -1
    assert lib.asvd_svd_worksize(1, 4, 4, 1, None) == -1
    assert lib.asvd_absstat_worksize(2048, 4096, ctypes.byref(nb)) == 0 and nb.value > 0
    assert lib.asvd_absstat_worksize(0, 4096, ctypes.byref(nb)) == -1
    # partial sums (64 x 64 tiles x 2 doubles) + zero-padded K-contiguous copies of the 16-bit factors ((m + n) x rp x 2 bytes)
    assert lib.asvd_reconstruct_worksize(4096, 4096, 512, ctypes.byref(nb)) == 0 and nb.value == 64 * 64 * 16 + (4096 + 4096) * 512 * 2
    assert lib.asvd_reconstruct_worksize(4096, 4096, 0, ctypes.byref(nb)) == -1
    assert lib.asvd_fro_worksize(4096, 4096, ctypes.byref(nb)) == 0


def test_bad_arguments_rejected_without_device(built_lib):
    lib = built_lib
    # null pointers / bad dtypes are rejected before any HIP call
    assert lib.asvd_absstat_accum(None, 1, 8, 8, 8, None, 1, 0, None, 0, None) == -1
    assert lib.asvd_make_scale(None, None, 1, 8, 0.5, 1e-6, None, None) == -1
    assert lib.asvd_truncate_split(None, 0, None, None, 0, None, 0, 1, 1, 1, 0, None, None, 1, None, None) == -1
    assert lib.asvd_svd(None, 0, 8, 8, 8, None, 0, None, None, None, 8, 0, 0.0, None, 0, None, None) == -1


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: device-required entry raises when no gfx950 is visible (this test only asserts on CPU-only boxes)."""
    import torch
    from asvd4llm_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AsvdHipError):
        _lib.load(require_device=True)
    with pytest.raises(_lib.AsvdHipError):
        ops.svd(torch.zeros(8, 8))
    lin = torch.nn.Linear(8, 8)
    from asvd4llm_amd.modules.svd_linear import SVDLinear
    with pytest.raises(_lib.AsvdHipError):
        SVDLinear.from_linear(lin, 0.5)


def test_environment_names_in_header_exist_in_the_sources():
    """The header is the contract a maintainer of the reference reads: every ASVD_* environment variable it mentions must be read somewhere
    in csrc/ (getenv) or in the package (os.environ), and every knob csrc/ reads must be documented in the header or in DESIGN.md — a knob
    that was deleted from the code may not survive in the prose."""
    src = open(os.path.join(ROOT, "include", "asvd_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    defined = set(re.findall(r"\b(ASVD_[A-Z0-9_]+)\b", code))            # macros / enumerators the header itself defines
    mentioned = set(re.findall(r"\b(ASVD_[A-Z0-9_]+)\b", src)) - defined - {"ASVD_E_", "ASVD_N_"}
    csrc = os.path.join(ROOT, "asvd4llm_amd", "csrc")
    read_c = set()
    for f in os.listdir(csrc):
        read_c |= set(re.findall(r'getenv\("(ASVD_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    read_py = set()
    for d, _, files in os.walk(os.path.join(ROOT, "asvd4llm_amd")):
        for f in files:
            if f.endswith(".py"):
                read_py |= set(re.findall(r'environ[^\n]*?"(ASVD_[A-Z0-9_]+)"', open(os.path.join(d, f)).read()))
    assert mentioned, "the header documents its environment knobs"
    stale = sorted(mentioned - read_c - read_py)
    assert not stale, f"asvd_hip.h mentions environment names nothing reads: {stale}"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    undocumented = sorted(n for n in read_c if n not in src and n not in design)
    assert not undocumented, f"csrc/ reads environment names neither asvd_hip.h nor DESIGN.md mention: {undocumented}"


def test_path_bits_of_the_header_match_the_python_binding():
    """ASVD_PATH_* (include/asvd_hip.h) are what ops.SvdInfo decodes: the values in asvd4llm_amd/_lib.py must be the header's"""
    from asvd4llm_amd import _lib as L
    src = open(os.path.join(ROOT, "include", "asvd_hip.h")).read()
    bits = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+ASVD_PATH_([A-Z_]+)\s+(\d+)", src)}
    assert set(bits) == {"REDUCED", "REDUCE_FALLBACK", "PLAIN_RETRY", "SPLIT", "SPLIT_REFUSED", "GRAM_RETRY"}, bits
    for name, val in bits.items():
        assert getattr(L, "PATH_" + name) == val, name
    assert len(set(bits.values())) == len(bits) and all(v & (v - 1) == 0 for v in bits.values())   # distinct single bits
