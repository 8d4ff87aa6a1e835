"""Throughput of the batched top-N kernel at LastFM-like size (host buffers in/out, so PCIe included)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cmfrec_amd import ops

rng = np.random.default_rng(0)
n, k = 160112, 50
B = rng.standard_normal((n, k)); bias = None
for dtype in (np.float64, np.float32):
    for nu in (1024, 8192):
        A = rng.standard_normal((nu, k)).astype(dtype)
        lens = rng.integers(10, 90, nu)
        ep = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        ei = rng.integers(0, n, int(lens.sum())).astype(np.int32)
        Bd = B.astype(dtype)
        ops.topN_batch(A[:64], Bd, 10, exclude=(ep[:65], ei[:int(ep[64])]))
        t0 = time.perf_counter()
        ids, sc = ops.topN_batch(A, Bd, 10, exclude=(ep, ei))
        dt = time.perf_counter() - t0
        S = A[:3].astype(np.float64) @ Bd.astype(np.float64).T
        ok = all(set(ids[u].tolist()) <= set(np.argsort(-S[u])[:200].tolist()) for u in range(3))
        print("%s nu=%5d: %.3f s  -> %8.0f users/s, %.2f Gscores/s (sanity %s)" % (np.dtype(dtype).name, nu, dt, nu / dt, nu * n / dt / 1e9, ok), flush=True)
