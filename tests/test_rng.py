"""CPU: the start-value generator of the product (host code inside libcmfrec_hip_*.so) against the
streams captured from the real reference (tests/golden/g7_rng_*.npz) and, where oracle/_ref is
available, against the reference run live on long streams and on whole (A, B) pairs."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
from oracle.bindings import Reference, ref_available


def draw(dtype, sizeA, sizeB, seed, normal):
    from cmfrec_amd import _lib
    lib = _lib.load(dtype)
    A = np.zeros(sizeA, dtype); B = np.zeros(max(sizeB, 1), dtype)
    exact = lib.cmfrec_hip_random_parallel(_lib.ptr(A), C.c_size_t(sizeA), _lib.ptr(B) if sizeB else None,
                                           C.c_size_t(sizeB), C.c_int(seed), C.c_bool(normal))
    return A, B[:sizeB], exact


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_streams_match_golden(dtype):
    g = gc.load("g7_rng", dtype)
    for seed in (1, 123):
        for size in (1000, 2 ** 18 + 1000):
            for normal in (True, False):
                A, _, exact = draw(dtype, size, 0, seed, normal)
                exp = g["seed%d_size%d_%s" % (seed, size, "normal" if normal else "unif")]
                if exact:
                    assert np.array_equal(A[:64], exp), (seed, size, normal)
                else:
                    assert np.allclose(A[:64], exp, rtol=4 * np.finfo(dtype).eps, atol=0)


def test_known_first_values():
    """seed 1 first normals / first uniforms (size > 2^18)."""
    A, _, _ = draw(np.float64, 1000, 0, 1, True)
    assert np.allclose(A[:4], [-0.00640981, -0.00368934, 0.01426238, 0.00135869], atol=5e-9)
    U, _, _ = draw(np.float64, 2 ** 18 + 1000, 0, 1, False)
    assert np.allclose(U[:4], [0.00360818, 0.00458913, 0.00620718, 0.00088105], atol=5e-9)
    N, _, _ = draw(np.float64, 1000, 0, 1, False)        # Q4: small requests are always normal
    assert (N < 0).any() and np.array_equal(N, A)
    assert np.abs(A).max() * 128 < 3.7                   # truncated normal


@pytest.mark.skipif(not ref_available(np.float64), reason="oracle/_ref not built")
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_long_streams_match_reference_live(dtype):
    R = Reference(dtype)
    for (sa, sb, seed, normal) in ((300000, 0, 7, True), (300001, 70001, 11, True), (300001, 70000, 11, False),
                                   (5000, 3000, 3, True), (2 ** 18 - 10, 20, 5, False)):
        A, B, exact = draw(dtype, sa, sb, seed, normal)
        Ar, Br = R.random_parallel(sa, sb, seed, normal, nthreads=3)
        if exact:
            assert np.array_equal(A, Ar) and np.array_equal(B, Br), (sa, sb, seed, normal)
        else:
            assert np.allclose(A, Ar, rtol=1e-6) and np.allclose(B, Br, rtol=1e-6)
