"""Probe of the Cholesky row kernel: time of one half-step for R identical rows of `nnz` entries each
(R <= 256 -> one row per workgroup: the per-row critical path; larger R -> throughput)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cmfrec_amd.session import AlsSession


def probe(R, nnz, k=129, n=60000, reps=5, dtype=np.float64):
    rng = np.random.default_rng(1)
    idx = np.stack([np.sort(rng.choice(n, nnz, replace=False)) for _ in range(min(R, 8))])
    idx = idx[np.arange(R) % len(idx)]
    indptr = (np.arange(R + 1) * nnz).astype(np.uint64)
    indices = idx.reshape(-1).astype(np.int32)
    vals = rng.normal(size=R * nnz)
    order = np.argsort(indices, kind="stable")
    cp = np.zeros(n + 1, np.uint64); np.cumsum(np.bincount(indices, minlength=n), out=cp[1:])
    rows = np.repeat(np.arange(R), nnz)[order].astype(np.int32)
    s = AlsSession(R, n, k, implicit=False, dtype=dtype, lam=10.0, use_cg=False)
    s.set_X((indptr, indices, vals), (cp, rows, vals[order]))
    s.set_factors(A=rng.normal(size=(R, k)) * 0.01, B=rng.normal(size=(n, k)) * 0.1)
    s.update("A", True); s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.update("A", True)
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    A = s.get_factors()["A"]
    return dt, np.isfinite(A).all()


if __name__ == "__main__":
    for R, nnz in [(1, 32), (1, 1024), (1, 8192), (1, 32768), (256, 32), (256, 1024), (256, 8192), (4096, 128), (4096, 1024)]:
        dt, ok = probe(R, nnz)
        chunks = nnz / 16.0
        print("R=%5d nnz=%6d  %9.1f us per half-step   %7.2f us per 16 gathered rows (per row-slot)  finite=%s"
              % (R, nnz, dt * 1e6, dt * 1e6 / chunks / max(1, R / 256), ok), flush=True)
