"""The library's own symmetric eigen-decomposition (eig_kernels.hpp: tridiagonalisation + QL) against the one-workgroup Jacobi kernel
on the matrices of the low-rank path (w C^T C at k_c = 128 / 256 / 257 / 320): device milliseconds per decomposition, residuals."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from cmfrec_amd import _lib

for dt in (np.float32, np.float64):
    lib = _lib.load(dt)
    for n in (64, 128, 256, 257, 320):
        rng = np.random.default_rng(n)
        Cm = rng.standard_normal((512, n)); A = (Cm.T @ Cm).astype(dt)
        for method, name in ((0, "tridiagonalisation + QL"), (2, "  of which tridiagonalisation"), (1, "one-workgroup Jacobi")):
            Q = np.empty((n, n), dt); lam = np.empty(n, dt); ms = C.c_double(0)
            rc = lib.cmfrec_hip_sym_eig(C.c_int(n), _lib.ptr(A), _lib.ptr(Q), _lib.ptr(lam), C.c_int(method), C.c_int(5), C.byref(ms))
            Q64, l64, A64 = Q.astype(np.float64), lam.astype(np.float64), A.astype(np.float64)
            res = np.abs(A64 @ Q64 - Q64 * l64).max() / np.abs(l64).max()
            orth = np.abs(Q64.T @ Q64 - np.eye(n)).max()
            print("%-7s n %3d %-24s %8.3f ms  residual %.1e  orthogonality %.1e  rc %d" % (np.dtype(dt).name, n, name, ms.value, res, orth, rc))
