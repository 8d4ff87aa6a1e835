"""C3 shape (MovieLens-10M-like, k=128, dense item side info q=64) with use_cg=True: the block CG path
(generic one-wavefront-per-row kernel) beside the Cholesky path."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from cmfrec_amd.session import AlsSession

m, n, nnz = 69_878, 10_677, 10_000_054
row, col, _ = bench.synth_block(m, n, nnz, seed=1)
rng = np.random.default_rng(1)
val = 0.5 * rng.integers(1, 11, nnz); val = val - val.mean()
II = rng.standard_normal((n, 64)); II -= II.mean(0)
for use_cg in (True, False):
    s = AlsSession(m, n, 128, implicit=False, dtype=np.float64, lam=0.05, use_cg=use_cg, max_cg_steps=3, user_bias=True,
                   item_bias=True, scale_lam=True, q=64, n_i=n)
    s.set_X_coo(row, col, val)
    s.set_sideinfo(II=II)
    s.set_factors(A=rng.standard_normal((m, 128)) * 2.0 ** -7, B=rng.standard_normal((n, 128)) * 2.0 ** -7,
                  biasA=np.zeros(m), biasB=np.zeros(n), Dm=np.zeros((64, 128)))
    s.iterate(1); s.sync(); s.reset_timers()
    t0 = time.perf_counter(); s.iterate(2); s.sync(); dt = (time.perf_counter() - t0) / 2
    a, ca = s.kernel_time("A"); b, cb = s.kernel_time("B")
    print("use_cg=%s: %.2f ms/iteration (A-step %.2f, B-step %.2f)" % (use_cg, dt * 1e3, a / ca, b / cb), flush=True)
