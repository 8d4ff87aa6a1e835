#!/usr/bin/env python3
"""Where a wavefront's time goes in the register-tiled CG kernels: a -DCMF_CG_TICKS build of the library
(make -C cmfrec_amd/csrc OUTDIR=../lib_ticks EXTRA=-DCMF_CG_TICKS; loaded through CMFREC_HIP_LIBDIR) sums, per kernel family,
the s_memtime ticks its wavefronts spend (a) between the last load of a row's gather being issued and the data being there,
(b) in the row's CG passes; this script runs BASELINE config 2 and prints the shares.

    CMFREC_HIP_LIBDIR=cmfrec_amd/lib_ticks CMFREC_HIP_BINS_PAR=1 python tools/microbench/cg_ticks.py [iterations]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                       # noqa: E402
from cmfrec_amd import _lib                        # noqa: E402
from cmfrec_amd.session import AlsSession          # noqa: E402

NAMES = ["cg_rows_kernel<W=1> (33..64)", "cg_rows_kernel<W=2> (65..128)", "cg_rows_kernel<W=4> (129..256)", "cg_rows_kernel<W=8> (257..512)",
         "(unused slot)"]


def read_ticks(lib):
    buf = (C.c_ulonglong * 32)()
    assert lib.cmfrec_hip_debug_cg_ticks(buf) == 0
    return np.array(buf[:], dtype=np.float64).reshape(8, 4)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    lib = _lib.load(np.float64)
    if not hasattr(lib, "cmfrec_hip_debug_cg_ticks"):
        raise SystemExit("not an instrumented build: set CMFREC_HIP_LIBDIR to a -DCMF_CG_TICKS build")
    m, n, K = bench.M_USERS, bench.N_ITEMS, bench.K
    row, col, val = bench.synth_block(m, n, bench.NNZ, seed=2)
    rng = np.random.default_rng(100)
    A0 = rng.random((m, K)) * 2.0 ** -7
    s = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=bench.LAM, use_cg=True, max_cg_steps=bench.MAX_CG_STEPS)
    s.set_X_coo(row, col, val.astype(np.float64))
    s.set_factors(A=A0, B=np.zeros((n, K)))
    s.iterate(3); s.sync()
    for which in ("B", "A"):
        read_ticks(lib)
        t0 = time.perf_counter()
        for _ in range(iters):
            s.update(which)
        s.sync()
        ms = (time.perf_counter() - t0) / iters * 1e3
        t = read_ticks(lib)
        print("%s-step, %d updates, %.3f ms each (instrumented build, bins %s)" % (which, iters, ms,
              "in line" if os.environ.get("CMFREC_HIP_BINS_PAR") == "1" else "side by side"))
        for i, name in enumerate(NAMES):
            wait, busy, rows, total = t[i]
            if rows == 0:
                continue
            print("   %-52s rows/update %8.0f | per row and wavefront: gather wait %7.0f ticks, passes %7.0f ticks | "
                  "of a wavefront's time in the row loop: wait %4.1f %%, passes %4.1f %%, rest %4.1f %%"
                  % (name, rows / iters, wait / max(rows, 1), busy / max(rows, 1), 100 * wait / total, 100 * busy / total,
                     100 * (total - wait - busy) / total))
            # (multi-wave teams: rows are counted by their first wavefront, the tick sums run over all W)


if __name__ == "__main__":
    main()
