"""The library's own MFMA GEMM (dense_kernels.hpp, gemm_mfma_kernel) against rocBLAS on the shapes of the side-information
path: w U C at configuration-5 width (1.56 M x 512 times 512 x 256), U^T A (512 x 1.56 M times 1.56 M x 256, split over K),
A^T A for k > 64, and configuration 3's (k = 128, q = 64 in double precision)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from cmfrec_amd import _lib

shapes = {np.float32: [("w U C  (c5 shard)", 1_562_500, 256, 512, 0), ("U^T A  (c5 shard)", 512, 256, 1_562_500, 1),
                       ("A^T A  k = 256", 256, 256, 1_562_500, 1), ("I D  (c5 shard items)", 125_000, 256, 512, 0)],
          np.float64: [("I D  (c3)", 10_677, 128, 64, 0), ("I^T B  (c3)", 64, 128, 10_677, 1), ("A^T A  k = 128", 128, 128, 69_878, 1),
                       ("w U C  fp64 200k x 64 x 64", 200_000, 64, 64, 0)]}
for dt, lst in shapes.items():
    lib = _lib.load(dt)
    lib.cmfrec_hip_gemm_probe.restype = C.c_int
    for name, M, N, K, ta in lst:
        a, b, d = C.c_double(0), C.c_double(0), C.c_double(0)
        rc = lib.cmfrec_hip_gemm_probe(C.c_int(M), C.c_int(N), C.c_int(K), C.c_int(ta), C.c_int(5), C.byref(a), C.byref(b), C.byref(d))
        fl = 2.0 * M * N * K
        print("%-7s %-28s M %8d N %4d K %8d : own %8.3f ms (%6.1f TFLOP/s)  rocBLAS %8.3f ms (%6.1f)  own / rocBLAS %.2f  max rel diff %.1e  rc %d"
              % (np.dtype(dt).name, name, M, N, K, a.value, fl / a.value / 1e9, b.value, fl / b.value / 1e9, a.value / b.value, d.value, rc))
