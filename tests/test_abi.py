"""CPU-side checks of the C-ABI boundary: the libraries load, export every symbol that
include/cmfrec_hip.h declares, and fail loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cmfrec_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt)
    keep = [n for n in names if n.startswith(("cmfrec_hip_", "fit_collective_", "factors_collective_", "precompute_collective_", "topN_old_collective_"))]
    return sorted(set(keep))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_exports(dtype):
    from cmfrec_amd import _lib
    lib = _lib.load(dtype)
    syms = declared_symbols()
    assert "fit_collective_implicit_als" in syms and "fit_collective_explicit_als" in syms
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
    assert set(syms) == set(_lib.EXPORTED)
    assert lib.cmfrec_hip_sizeof_real() == np.dtype(dtype).itemsize
    assert b"gfx950" in lib.cmfrec_hip_build_info()


def test_model_struct_layout():
    from cmfrec_amd import _lib
    # 18 int32 + lam, w_user, w_item + row/col ranges (4) + m_x, n_x; and the compiled struct says the same
    assert C.sizeof(_lib.Model) == 18 * 4 + 3 * 8 + 6 * 4
    assert C.sizeof(_lib.ModelF) == 18 * 4 + 3 * 4 + 6 * 4
    for dt, mirror in ((np.float64, _lib.Model), (np.float32, _lib.ModelF)):
        lib = C.CDLL(_lib.lib_path(dt))
        assert lib.cmfrec_hip_sizeof_model() == C.sizeof(mirror)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cmfrec_amd import ops, CMF_implicit
    A = np.zeros((4, 8)); B = np.ones((3, 8))
    csr = (np.array([0, 1, 2, 2, 3], np.uint64), np.array([0, 1, 2], np.int32), np.ones(3))
    with pytest.raises((RuntimeError, MemoryError)):
        ops.optimizeA_implicit(A, B, csr, 1.0)
    with pytest.raises((RuntimeError, MemoryError)):
        CMF_implicit(k=4, niter=1, use_float=False).fit((np.array([0, 1]), np.array([1, 2]), np.ones(2)), shape=(4, 3))


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under cmfrec_amd/ may reference it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "cmfrec_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|oracle/|cmf_oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
