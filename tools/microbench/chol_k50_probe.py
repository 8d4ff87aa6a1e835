"""One Cholesky ALS iteration at k=50 on the C1 (explicit, biases) and C2 (implicit) shapes: what the default
finalize_chol=True of the explicit model costs next to a CG iteration."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from cmfrec_amd.session import AlsSession

rng = np.random.default_rng(1)
m, n, nnz = 69_878, 10_677, 10_000_054
row, col, _ = bench.synth_block(m, n, nnz, seed=1)
val = 0.5 * rng.integers(1, 11, nnz); val = val - val.mean()
for use_cg in (True, False):
    s = AlsSession(m, n, 50, implicit=False, dtype=np.float64, lam=0.05, use_cg=use_cg, max_cg_steps=3, user_bias=True,
                   item_bias=True, scale_lam=True)
    s.set_X_coo(row, col, val)
    s.set_factors(A=rng.standard_normal((m, 50)) * 2.0 ** -7, B=np.zeros((n, 50)), biasA=np.zeros(m), biasB=np.zeros(n))
    s.iterate(1); s.sync(); s.reset_timers()
    t0 = time.perf_counter(); s.iterate(3); s.sync(); dt = (time.perf_counter() - t0) / 3
    a, ca = s.kernel_time("A"); b, cb = s.kernel_time("B")
    print("C1 shape  use_cg=%-5s %.2f ms/iteration (A %.2f, B %.2f)" % (use_cg, dt * 1e3, a / ca, b / cb), flush=True)
row, col, val = bench.synth_block(bench.M_USERS, bench.N_ITEMS, bench.NNZ, seed=2)
for use_cg in (True, False):
    s = AlsSession(bench.M_USERS, bench.N_ITEMS, 50, implicit=True, dtype=np.float64, lam=5.0, use_cg=use_cg, max_cg_steps=3)
    s.set_X_coo(row, col, val)
    s.set_factors(A=rng.random((bench.M_USERS, 50)) * 2.0 ** -7, B=np.zeros((bench.N_ITEMS, 50)))
    s.iterate(1); s.sync(); s.reset_timers()
    t0 = time.perf_counter(); s.iterate(3); s.sync(); dt = (time.perf_counter() - t0) / 3
    a, ca = s.kernel_time("A"); b, cb = s.kernel_time("B")
    print("C2 shape  use_cg=%-5s %.2f ms/iteration (A %.2f, B %.2f)" % (use_cg, dt * 1e3, a / ca, b / cb), flush=True)
