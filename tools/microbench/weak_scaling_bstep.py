"""What one rank's half-steps cost in the weak-scaling layout of bench.py at N GPUs, measured on ONE GPU: rank 0's shard
is built exactly as GpuEngine.from_user_block builds it (own user block as CSR; its 1/N of the items as CSC over the
users of all N blocks) and update('B') / update('A') are timed without any collective.  Shows how the item rows grow
N times heavier with N (more of them take the split-row path)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from cmfrec_amd.session import AlsSession
from cmfrec_amd.distributed import equal_boundaries

K = bench.K
for N in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    m_blk, n = bench.M_USERS, bench.N_ITEMS
    cb = equal_boundaries(n, N)
    c0, c1 = cb[0], cb[1]
    rows, cols, vals = [], [], []
    own = None
    for r in range(N):
        row, col, val = bench.synth_block(m_blk, n, bench.NNZ, seed=2 + r)
        if r == 0:
            own = (row, col, val)
        keep = (col >= c0) & (col < c1)
        rows.append(row[keep].astype(np.int64) + r * m_blk); cols.append(col[keep] - c0); vals.append(val[keep])
    crow = np.concatenate(rows).astype(np.int32); ccol = np.concatenate(cols); cval = np.concatenate(vals)
    csr = bench.to_csr(own[0], own[1], own[2], m_blk)
    csc = bench.to_csr(ccol, crow, cval, c1 - c0)
    m = m_blk * N
    s = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=bench.LAM, use_cg=True, max_cg_steps=3,
                   row_range=(0, m_blk), col_range=(c0, c1))
    s.set_X(csr, csc)
    rng = np.random.default_rng(1)
    s.set_factors(A=rng.random((m, K)) * 2.0 ** -7, B=rng.random((n, K)) * 2.0 ** -7)
    for _ in range(2):
        s.update("B"); s.update("A")
    s.sync(); s.reset_timers()
    for _ in range(5):
        s.update("B"); s.update("A")
    s.sync()
    a, ca = s.kernel_time("A"); b, cb_ = s.kernel_time("B")
    nnz_c = len(cval)
    cnt = np.bincount(ccol, minlength=c1 - c0)
    print("N=%d: B-step %.2f ms (%d items, %.1f M entries, %.0f %% of them in rows > 1024), A-step %.2f ms"
          % (N, b / cb_, c1 - c0, nnz_c / 1e6, 100.0 * cnt[cnt > 1024].sum() / max(nnz_c, 1), a / ca), flush=True)
    for bin_ in range(6):
        ms, c, r_, z = s.bin_stats("B", bin_)
        if c:
            print("     bin %d: %.3f ms  rows %d nnz %.2f M" % (bin_, ms / c, r_, z / 1e6), flush=True)
    s.close()
