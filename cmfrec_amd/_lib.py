"""ctypes binding of the C ABI in include/cmfrec_hip.h (libcmfrec_hip_{double,float}.so).

The libraries are built in-tree by ``__graft_entry__.build()`` / ``make -C cmfrec_amd/csrc``.
There is no CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMFREC_HIP_LIBDIR: load the two libraries from another directory (a second build kept beside the default one)
LIBDIR = os.environ.get("CMFREC_HIP_LIBDIR") or os.path.join(_HERE, "lib")

RETURN_CODES = {0: "ok", 1: "out of memory", 2: "invalid or unsupported input", 3: "interrupted",
                4: "HIP runtime failure"}


class Model(C.Structure):
    """``cmfrec_hip_model`` (include/cmfrec_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "implicit", "m", "n", "k", "k_main", "k_user", "k_item", "user_bias", "item_bias",
        "scale_lam", "scale_lam_sideinfo", "use_cg", "precondition_cg", "max_cg_steps",
        "p", "q", "m_u", "n_i")] + [("lam", C.c_double), ("w_user", C.c_double), ("w_item", C.c_double)] + \
        [(n, C.c_int32) for n in ("row_begin", "row_end", "col_begin", "col_end", "m_x", "n_x")]


class ModelF(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "implicit", "m", "n", "k", "k_main", "k_user", "k_item", "user_bias", "item_bias",
        "scale_lam", "scale_lam_sideinfo", "use_cg", "precondition_cg", "max_cg_steps",
        "p", "q", "m_u", "n_i")] + [("lam", C.c_float), ("w_user", C.c_float), ("w_item", C.c_float)] + \
        [(n, C.c_int32) for n in ("row_begin", "row_end", "col_begin", "col_end", "m_x", "n_x")]


EXPORTED = [
    "fit_collective_implicit_als", "fit_collective_explicit_als",
    "factors_collective_explicit_multiple", "factors_collective_implicit_multiple", "cmfrec_hip_factors_multiple", "cmfrec_hip_factors_multiple_l1",
    "cmfrec_hip_optimizeA_implicit", "cmfrec_hip_optimizeA_explicit", "cmfrec_hip_optimizeA_explicit_weighted",
    "cmfrec_hip_optimizeA_dense_full", "cmfrec_hip_optimizeA_collective", "cmfrec_hip_optimizeA_collective_sparse", "cmfrec_hip_topN_batch",
    "cmfrec_hip_session_create", "cmfrec_hip_session_destroy", "cmfrec_hip_last_error", "cmfrec_hip_last_error_code",
    "cmfrec_hip_session_set_X", "cmfrec_hip_session_set_A_parts", "cmfrec_hip_session_nparts", "cmfrec_hip_session_part_range", "cmfrec_hip_session_stream_wait_part", "cmfrec_hip_session_set_X_coo", "cmfrec_hip_session_set_X_coo_weighted", "cmfrec_hip_session_set_X_weighted", "cmfrec_hip_session_set_X_coo_device", "cmfrec_hip_session_precompute", "cmfrec_hip_session_init_biases", "cmfrec_hip_session_get_X", "cmfrec_hip_session_set_factors", "cmfrec_hip_session_get_factors",
    "cmfrec_hip_session_set_sideinfo", "cmfrec_hip_session_set_sideinfo_local", "cmfrec_hip_session_sideinfo_partial", "cmfrec_hip_session_sideinfo_finish", "cmfrec_hip_session_set_nonneg", "cmfrec_hip_session_set_l1", "cmfrec_hip_session_set_lam_unique", "cmfrec_hip_session_set_scale_bias_const", "cmfrec_hip_session_set_NA_as_zero_X", "cmfrec_hip_session_set_zero_rows", "cmfrec_hip_session_set_closed_form_rows", "cmfrec_hip_session_set_lambda_multipliers", "cmfrec_hip_session_set_implicit_features", "cmfrec_hip_session_get_implicit_features", "cmfrec_hip_session_set_sideinfo_sparse", "cmfrec_hip_session_update", "cmfrec_hip_session_iterate",
    "cmfrec_hip_session_sync", "cmfrec_hip_session_device_ptr", "cmfrec_hip_session_stream",
    "cmfrec_hip_session_after_gather", "cmfrec_hip_session_kernel_time",
    "cmfrec_hip_session_reset_timers", "cmfrec_hip_session_bin_overlaps", "cmfrec_hip_session_vh_mode", "cmfrec_hip_session_vh_min", "cmfrec_hip_session_lowrank_info", "cmfrec_hip_session_bin_stats", "cmfrec_hip_sizeof_real", "cmfrec_hip_sizeof_model", "cmfrec_hip_build_info", "cmfrec_hip_reload_switches", "cmfrec_hip_selftest_lanes", "cmfrec_hip_exchange_plan", "cmfrec_hip_gemm_probe", "cmfrec_hip_sym_eig", "precompute_collective_explicit", "precompute_collective_implicit", "topN_old_collective_explicit", "topN_old_collective_implicit", "cmfrec_hip_random_parallel",
]

_cache = {}


def lib_path(dtype):
    suffix = "double" if np.dtype(dtype) == np.float64 else "float"
    return os.path.join(LIBDIR, "libcmfrec_hip_%s.so" % suffix)


def load(dtype=np.float64):
    """Returns the ctypes handle of the library for ``dtype`` (float64 / float32)."""
    dtype = np.dtype(dtype).type
    if dtype not in (np.float64, np.float32):
        raise ValueError("dtype must be float64 or float32")
    if dtype in _cache:
        return _cache[dtype]
    path = lib_path(dtype)
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: the HIP extension has not been built (run __graft_entry__.build() or "
            "`make -C cmfrec_amd/csrc`). cmfrec_amd has no CPU fallback." % path)
    # PyTorch ships its own HIP runtime under the same soname as /opt/rocm's.  Whichever is mapped first serves the whole
    # process; if this library's copy came first, a later `import torch` would run on a runtime its other libraries were not
    # built against ("No HIP GPUs are available").  The package uses torch for device memory and streams anyway, so it goes
    # first whenever it is installed.
    # A process that will never import torch can skip this (seconds of import time, torch's HIP runtime initialised):
    # CMFREC_AMD_NO_TORCH_PRELOAD=1.  The ordering requirement then is the caller's: torch, if it comes at all, before this library.
    if os.environ.get("CMFREC_AMD_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    lib.cmfrec_hip_last_error.restype = C.c_char_p
    lib.cmfrec_hip_build_info.restype = C.c_char_p
    lib.cmfrec_hip_session_create.restype = C.c_void_p
    lib.cmfrec_hip_session_device_ptr.restype = C.c_void_p
    lib.cmfrec_hip_session_stream.restype = C.c_void_p
    assert lib.cmfrec_hip_sizeof_real() == np.dtype(dtype).itemsize
    mirror = Model if dtype is np.float64 else ModelF
    if lib.cmfrec_hip_sizeof_model() != C.sizeof(mirror):
        raise RuntimeError("cmfrec_amd: the ctypes mirror of cmfrec_hip_model (%d bytes) does not match the library (%d)"
                           % (C.sizeof(mirror), lib.cmfrec_hip_sizeof_model()))
    _cache[dtype] = lib
    return lib


def real(dtype):
    return C.c_double if np.dtype(dtype) == np.float64 else C.c_float


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(rc, lib, what, interrupt_ok=False):
    if rc == 0 or (rc == 3 and interrupt_ok):
        return
    msg = lib.cmfrec_hip_last_error()
    msg = msg.decode() if msg else ""
    if rc == 1:
        raise MemoryError("%s: out of memory. %s" % (what, msg))
    if rc == 3:
        raise InterruptedError("%s: procedure was interrupted" % what)
    raise RuntimeError("%s failed with code %d (%s). %s" % (what, rc, RETURN_CODES.get(rc, "?"), msg))
