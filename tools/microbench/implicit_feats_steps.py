"""Per-update timings of the implicit-features iteration at the C1 shape (MovieLens-10M, k = 50 fp64, CG):
python tools/microbench/implicit_feats_steps.py            (CMFREC_HIP_NAZ_PER_ROW=1: the per-row factorisation of Bi / Ai)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from cmfrec_amd.session import AlsSession

m, n, nnz, k = 69_878, 10_677, 10_000_054, 50
row, col, _ = bench.synth_block(m, n, nnz, seed=1)
rng = np.random.default_rng(1)
val = 0.5 * rng.integers(1, 11, nnz); val = val - val.mean()
sess = AlsSession(m, n, k, implicit=False, dtype=np.float64, lam=0.05, use_cg=True, max_cg_steps=3, user_bias=True, item_bias=True,
                  scale_lam=True, device=0)
sess.set_X(bench.to_csr(row, col, val, m), bench.to_csr(col, row, val, n))
sess.set_factors(A=rng.standard_normal((m, k)) * 2.0 ** -7, B=rng.standard_normal((n, k)) * 2.0 ** -7, biasA=np.zeros(m), biasB=np.zeros(n))
sess.set_implicit_features(0.5)
sess.iterate(2); sess.sync()
for which in "baBA":
    reps = 10
    sess.sync(); t0 = time.perf_counter()
    for _ in range(reps):
        sess.update(which, use_cholesky=which in "ab")
    sess.sync()
    print("update %s: %.3f ms" % (which, (time.perf_counter() - t0) / reps * 1e3))
