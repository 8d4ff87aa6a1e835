import numpy as np, time, sys
sys.path.insert(0, ".")
import bench
from cmfrec_amd.models import CMF
import scipy.sparse as sp
m, n, nnz = 69878, 10677, 10000054
row, col, _ = bench.synth_block(m, n, nnz, seed=1)
rng = np.random.default_rng(1)
val = 0.5 * rng.integers(1, 11, nnz)
X = sp.coo_matrix((val, (row, col)), shape=(m, n))
mdl = CMF(k=50, lambda_=0.05, scale_lam=True, niter=15, use_cg=True, finalize_chol=False, use_float=False, precompute_for_predictions=False)
for _ in range(2):
    t0 = time.perf_counter(); mdl.fit(X); print("C1 whole fit", time.perf_counter() - t0, "s")
print(np.isfinite(mdl.A_).all(), mdl.glob_mean_)
