"""ctypes binding of libasvd_hip.so (C ABI in include/asvd_hip.h).

The HIP library is the product path: there is NO CPU fallback.  `load()` raises if the shared object is missing or if
no gfx950 device is visible when a device is required."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ASVD_HIP_LIB") or os.path.join(_HERE, "libasvd_hip.so")   # ASVD_HIP_LIB: A/B runs of measurement builds (tools/)

F32, F16, BF16 = 0, 1, 2
FUSE = {"UV": 0, "U": 1, "V": 2}
STAT_ABS_MEAN, STAT_ABS_MAX, STAT_SQ_MEAN = 0, 1, 2
PATH_REDUCED, PATH_REDUCE_FALLBACK, PATH_PLAIN_RETRY, PATH_SPLIT, PATH_SPLIT_REFUSED, PATH_GRAM_RETRY = 1, 2, 4, 8, 16, 32
OK, E_BADARG, E_WORKSPACE, E_HIP, E_NODEVICE, N_NOCONV, N_NAN = 0, -1, -2, -3, -4, 1, 2

_c = ctypes
_vp, _i, _i64, _f, _sz = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

# name -> (restype, argtypes); mirrors include/asvd_hip.h one to one
SIGNATURES = {
    "asvd_version": (_i, []),
    "asvd_status_string": (_c.c_char_p, [_i]),
    "asvd_device_count": (_i, []),
    "asvd_absstat_worksize": (_i, [_i64, _i64, _c.POINTER(_sz)]),
    "asvd_absstat_accum": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _i, _vp, _sz, _vp]),
    "asvd_absstat_partial": (_i, [_vp, _i, _i64, _i64, _i64, _i, _vp, _sz, _vp]),
    "asvd_absstat_finalize": (_i, [_vp, _sz, _i64, _i64, _vp, _i, _i, _vp]),
    "asvd_make_scale": (_i, [_vp, _vp, _i, _i64, _f, _f, _vp, _vp]),
    "asvd_scale_cols": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _vp, _i64, _vp]),
    "asvd_svd_worksize": (_i, [_i, _i64, _i64, _i, _c.POINTER(_sz)]),
    "asvd_svd_batched": (_i, [_i, _c.POINTER(_vp), _i, _i64, _i64, _i64, _c.POINTER(_vp), _i, _c.POINTER(_vp),
                              _c.POINTER(_vp), _c.POINTER(_vp), _i64, _i, _f, _vp, _sz, _c.POINTER(_i), _vp]),
    "asvd_svd": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _vp, _vp, _vp, _i64, _i, _f, _vp, _sz, _c.POINTER(_i), _vp]),
    "asvd_sigma_max_worksize": (_i, [_i, _i64, _i64, _i, _c.POINTER(_sz)]),
    "asvd_sigma_max_batched": (_i, [_i, _c.POINTER(_vp), _i, _i64, _i64, _i64, _c.POINTER(_vp), _i, _f, _vp, _sz, _c.POINTER(_i), _vp]),
    "asvd_truncate_split": (_i, [_vp, _i64, _vp, _vp, _i64, _vp, _i, _i64, _i64, _i64, _i, _vp, _vp, _i, _vp, _vp]),
    "asvd_truncate_split_batched": (_i, [_i, _c.POINTER(_vp), _i64, _c.POINTER(_vp), _c.POINTER(_vp), _i64, _c.POINTER(_vp), _i, _i64, _i64, _i64, _i,
                                         _c.POINTER(_vp), _c.POINTER(_vp), _i, _vp, _vp]),
    "asvd_make_scale_batched": (_i, [_i, _c.POINTER(_vp), _c.POINTER(_vp), _i, _i64, _f, _f, _c.POINTER(_vp), _vp]),
    "asvd_comm_init": (_i, [_c.POINTER(_vp), _i, _i, _i, _c.c_char_p, _i]),
    "asvd_comm_allgather_f32": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "asvd_comm_allgather_f64": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "asvd_comm_destroy": (_i, [_vp]),
    "asvd_lowrank_padded_rank": (_i64, [_i64]),
    "asvd_lowrank_work_bytes": (_sz, [_i64, _i64]),
    "asvd_lowrank_forward_f16": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "asvd_fro_worksize": (_i, [_i64, _i64, _c.POINTER(_sz)]),
    "asvd_fro_norm_sq": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "asvd_reconstruct_worksize": (_i, [_i64, _i64, _i64, _c.POINTER(_sz)]),
    "asvd_reconstruct_err": (_i, [_vp, _i, _i64, _vp, _vp, _i, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "asvd_test_supdate": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "asvd_test_supgram": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "asvd_test_gram": (_i, [_vp, _i64, _i64, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp]),
    "asvd_test_super_schedule": (_i, [_i, _i, _vp, _i, _c.POINTER(_i), _c.POINTER(_i)]),
    "asvd_test_evd_wave": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "asvd_svd_get_last_path": (_i, []),
    "asvd_svd_set_profiling": (None, [_i]),
    "asvd_svd_set_call_cus": (None, [_i]),
    "asvd_svd_set_split": (None, [_i]),
    "asvd_svd_get_split_profile": (_i, [_c.POINTER(_f), _c.POINTER(_i), _c.POINTER(_f)]),
    "asvd_svd_get_profile": (_i, [_c.POINTER(_f), _c.POINTER(_i)]),
    "asvd_svd_get_pair_counts": (_i, [_c.POINTER(_c.c_longlong)]),
    "asvd_svd_get_sweep_times": (_i, [_c.POINTER(_f), _c.POINTER(_c.c_longlong), _i]),
}

_lib = None


class AsvdHipError(RuntimeError):
    pass


_device_seen = False


def load(require_device=False):
    """dlopen the in-tree library and attach prototypes.  Raises AsvdHipError when it is absent (run
    `python -m asvd4llm_amd.build` / `__graft_entry__.build()`), or when require_device and no gfx950 GPU is visible."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AsvdHipError(f"{LIB_PATH} not built: run `python -m asvd4llm_amd.build` (hipcc --offload-arch=gfx950). "
                               "There is no CPU fallback for the ASVD hot path.")
        # torch first: it bundles its own libamdhip64, and a process must not end up with two HIP runtimes — if this library pulled in
        # /opt/rocm's copy before torch loaded, torch.cuda later reports "No HIP GPUs are available" (seen with build() + smoke() in one process)
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    global _device_seen
    if require_device and not _device_seen:  # asked once per process: a device does not go away, and hot wrappers call this per launch
        if _lib.asvd_device_count() <= 0:
            raise AsvdHipError("libasvd_hip.so loaded but no gfx950 device is visible; the ASVD hot path has no CPU fallback")
        _device_seen = True
    return _lib


def status_string(code):
    return load().asvd_status_string(int(code)).decode()


def check(code, what):
    """negative status -> exception; positive (numerical) status is returned to the caller"""
    if code < 0:
        raise AsvdHipError(f"{what} failed: {status_string(code)} ({code})")
    return code
