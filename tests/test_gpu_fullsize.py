"""Parity properties at BASELINE.json's full size (configuration 2: 358,858 x 160,112, 17.3 M non-zeros, k=50 fp64),
where the CPU oracle would take minutes per call: size-independent properties of the ALS half-steps instead
(run-to-run bit reproducibility, the normal equations of sampled rows, monotone objective, invariance to the order
of the COO entries, CG -> closed form)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

K, LAM = 50, 5.0


@pytest.fixture(scope="module")
def c2():
    import bench
    row, col, val = bench.synth_block(bench.M_USERS, bench.N_ITEMS, bench.NNZ, seed=2)
    rng = np.random.default_rng(100)
    A0 = rng.random((bench.M_USERS, K)) * 2.0 ** -7
    return bench.M_USERS, bench.N_ITEMS, row, col, val.astype(np.float64), A0


def _session(c2, use_cg, rowperm=None):
    from cmfrec_amd.session import AlsSession
    m, n, row, col, val, A0 = c2
    s = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=LAM, use_cg=use_cg, max_cg_steps=3)
    if rowperm is None:
        s.set_X_coo(row, col, val)
    else:
        s.set_X_coo(row[rowperm], col[rowperm], val[rowperm])
    s.set_factors(A=A0, B=np.zeros((n, K)))
    return s


def _objective(A, B, row, col, val, lam):
    """Implicit-feedback objective  sum_all c_ui (p_ui - a_u.b_i)^2 + lam (|A|^2 + |B|^2),  c = 1 + x, p = [x > 0],
    with the dense part through  sum_all (a.b)^2 = <A^T A, B^T B>  (Hu, Koren, Volinsky; the model of
    fit_collective_implicit_als, src/collective.c:9375)."""
    pred = np.einsum("ij,ij->i", A[row], B[col])
    dense = float(np.sum((A.T @ A) * (B.T @ B)))
    nz = float(np.sum((1.0 + val) * (1.0 - pred) ** 2 - pred ** 2))
    return dense + nz + lam * (float(np.sum(A * A)) + float(np.sum(B * B)))


def test_full_size_properties(c2):
    m, n, row, col, val, A0 = c2
    # ---- 1. bit reproducibility: two independent sessions, two CG iterations ----
    s1 = _session(c2, True); s1.iterate(2); f1 = s1.get_factors()
    s2 = _session(c2, True); s2.iterate(2); f2 = s2.get_factors()
    assert np.array_equal(f1["A"], f2["A"]) and np.array_equal(f1["B"], f2["B"])
    assert np.isfinite(f1["A"]).all() and np.isfinite(f1["B"]).all()
    # ---- 2. the objective never increases over CG iterations ----
    obj = [_objective(f1["A"], f1["B"], row, col, val, LAM)]
    for _ in range(3):
        s1.iterate(1); f = s1.get_factors()
        obj.append(_objective(f["A"], f["B"], row, col, val, LAM))
    assert all(b <= a * (1 + 1e-12) for a, b in zip(obj, obj[1:])), obj
    # ---- 3. the order of the COO entries only changes rounding (row sums are re-associated) ----
    perm = np.random.default_rng(5).permutation(len(row))
    s3 = _session(c2, True, rowperm=perm); s3.iterate(2); f3 = s3.get_factors()
    assert np.abs(f3["A"] - f1["A"]).max() <= 1e-9 * np.abs(f1["A"]).max()
    assert np.abs(f3["B"] - f1["B"]).max() <= 1e-9 * np.abs(f1["B"]).max()


def test_full_size_normal_equations(c2):
    """Cholesky half-step at full size: sampled users (the heaviest, the lightest, random ones) satisfy
    (B^T B + lam I + sum_j x_j b_j b_j^T) a = sum_j (x_j + 1) b_j   (factors_implicit_chol, common.c:2063-2126)."""
    m, n, row, col, val, A0 = c2
    from cmfrec_amd.session import AlsSession
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, K)) * 0.1
    s = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=LAM, use_cg=False)
    s.set_X_coo(row, col, val)
    s.set_factors(A=A0, B=B)
    s.update("A")
    A = s.get_factors()["A"]
    cnt = np.bincount(row, minlength=m)
    order = np.argsort(row, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    G = B.T @ B + LAM * np.eye(K)
    users = np.concatenate([np.argsort(-cnt)[:5], np.nonzero(cnt == 1)[0][:5], rng.choice(np.nonzero(cnt > 0)[0], 40, replace=False)])
    for u in users:
        e = order[ptr[u]:ptr[u + 1]]
        Bu = B[col[e]]; x = val[e]
        M = G + (Bu * x[:, None]).T @ Bu
        rhs = Bu.T @ (x + 1.0)
        assert np.abs(M @ A[u] - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max()), (u, cnt[u])
    assert not A[cnt == 0].any()                                   # rows without entries are zero in Cholesky mode (:3334)


def test_full_size_explicit_cg_reaches_closed_form():
    """Configuration 1's shape (69,878 x 10,677, 10 M ratings): many CG steps from a warm start converge to the
    Cholesky solution of the same half-step (explicit model, common.c:1098-1188 vs :978-1070)."""
    import bench
    from cmfrec_amd.session import AlsSession
    m, n, nnz, k = 69_878, 10_677, 10_000_054, 50
    row, col, _ = bench.synth_block(m, n, nnz, seed=1)
    rng = np.random.default_rng(1)
    val = 0.5 * rng.integers(1, 11, nnz); val = val - val.mean()
    B = rng.standard_normal((n, k)) * 0.1
    out = {}
    for use_cg in (False, True):
        s = AlsSession(m, n, k, implicit=False, dtype=np.float64, lam=0.05, use_cg=use_cg, max_cg_steps=60, scale_lam=True)
        s.set_X_coo(row, col, val)
        s.set_factors(A=np.zeros((m, k)), B=B)
        s.update("A")
        out[use_cg] = s.get_factors()["A"]
    err = np.abs(out[True] - out[False]).max() / np.abs(out[False]).max()
    assert err < 1e-3, err                       # CG stops at |r|^2 <= 1e-8 (absolute, common.c:1180)


def test_c4_shard_properties():
    """One GPU's share of configuration 4 (1.25 M x 1 M, 62.5 M entries, implicit ALS-CG k = 64 in SINGLE precision, the data
    generated on the device as `bench.py --gpus 8` does): two independent runs are bit-identical (no floating-point
    atomics, fixed summation orders -- also with 64-bit CSR offsets at this size) and the implicit objective never
    increases over the iterations."""
    import torch
    import bench
    from cmfrec_amd.session import AlsSession
    dev = torch.device("cuda", 0)
    m, n, nnz, k = 1_250_000, 1_000_000, 62_500_000, 64
    row, col, val = bench.synth_block_torch(m, n, nnz, seed=40, item_seed=4, device=dev)
    hrow, hcol, hval = row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    del row, col, val
    torch.cuda.empty_cache()
    A0 = (np.random.default_rng(7).random((m, k), dtype=np.float32) * np.float32(2.0 ** -7))

    def run(niter):
        s = AlsSession(m, n, k, implicit=True, dtype=np.float32, lam=LAM, use_cg=True, max_cg_steps=3)
        s.set_X_coo(hrow, hcol, hval)
        s.set_factors(A=A0, B=np.zeros((n, k), np.float32))
        out = []
        for _ in range(niter):
            s.iterate(1)
            out.append(s.get_factors())
        return out

    r1, r2 = run(3), run(2)
    assert np.array_equal(r1[1]["A"], r2[1]["A"]) and np.array_equal(r1[1]["B"], r2[1]["B"])
    assert np.isfinite(r1[2]["A"]).all() and np.isfinite(r1[2]["B"]).all()
    v64 = hval.astype(np.float64)
    obj = [_objective(f["A"].astype(np.float64), f["B"].astype(np.float64), hrow, hcol, v64, LAM) for f in r1]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(obj, obj[1:])), obj


def test_c3_shape_normal_equations():
    """Configuration 3's shape (69,878 x 10,677, 10 M ratings, k = 128 + biases, dense item side information q = 64, double
    precision, Cholesky): sampled ITEM rows -- the heaviest (sliced over several wavefronts), single-entry ones, random
    ones -- satisfy the collective normal equations
        (sum_j a_j a_j^T + w D^T D (+) 0 + lam n_i I) b = w D^T i_row (+) 0 + sum_j (x_ij - bias_j) a_j
    with the user-bias column of A fixed to 1 (collective_closed_form_block, collective.c:1534-1846), and sampled USER rows
    the plain ones (factors_closed_form, common.c:978-1070)."""
    import bench
    from cmfrec_amd.session import AlsSession
    m, n, nnz, k, q = 69_878, 10_677, 10_000_054, 128, 64
    row, col, _ = bench.synth_block(m, n, nnz, seed=1)
    rng = np.random.default_rng(1)
    val = 0.5 * rng.integers(1, 11, nnz); val = val - val.mean()
    II = rng.standard_normal((n, q)); II -= II.mean(0)
    lam, w = 0.05, 1.0
    s = AlsSession(m, n, k, implicit=False, dtype=np.float64, lam=lam, use_cg=False, user_bias=True, item_bias=True, scale_lam=True,
                   q=q, n_i=n, w_item=w)
    s.set_X_coo(row, col, val)
    s.set_sideinfo(II=II)
    A0 = rng.standard_normal((m, k)) * 0.1
    biasA0 = rng.standard_normal(m) * 0.1
    Dm0 = rng.standard_normal((q, k)) * 0.1
    s.set_factors(A=A0, B=np.zeros((n, k)), biasA=biasA0, biasB=np.zeros(n), Dm=Dm0)
    s.update("B"); s.after_gather("B")
    f = s.get_factors()
    B, biasB = f["B"], f["biasB"]
    cnt = np.bincount(col, minlength=n)
    order = np.argsort(col, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    kt = k + 1
    items = np.concatenate([np.argsort(-cnt)[:4], np.nonzero(cnt == 1)[0][:4], rng.choice(np.nonzero(cnt > 0)[0], 24, replace=False)])
    assert cnt[items[0]] > 2048
    Ab = np.concatenate([A0, np.ones((m, 1))], axis=1)             # unknowns of an item: [b (k) | bias], gathered rows [a_j | 1]
    DtD = w * Dm0.T @ Dm0
    for i in items:
        e = order[ptr[i]:ptr[i + 1]]
        Aj = Ab[row[e]]; x = val[e] - biasA0[row[e]]
        M = Aj.T @ Aj + lam * cnt[i] * np.eye(kt)
        M[:k, :k] += DtD
        rhs = Aj.T @ x
        rhs[:k] += w * (II[i] @ Dm0)
        sol = np.concatenate([B[i], [biasB[i]]])
        assert np.abs(M @ sol - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max()), (i, cnt[i])
    # user side: plain explicit rows with the item biases just computed
    s.update("A"); s.after_gather("A")
    f2 = s.get_factors()
    A, biasA = f2["A"], f2["biasA"]
    ucnt = np.bincount(row, minlength=m)
    uorder = np.argsort(row, kind="stable")
    uptr = np.concatenate([[0], np.cumsum(ucnt)])
    Bb = np.concatenate([B, np.ones((n, 1))], axis=1)
    users = np.concatenate([np.argsort(-ucnt)[:4], rng.choice(np.nonzero(ucnt > 0)[0], 24, replace=False)])
    for u in users:
        e = uorder[uptr[u]:uptr[u + 1]]
        Bj = Bb[col[e]]; x = val[e] - biasB[col[e]]
        M = Bj.T @ Bj + lam * ucnt[u] * np.eye(kt)
        rhs = Bj.T @ x
        sol = np.concatenate([A[u], [biasA[u]]])
        assert np.abs(M @ sol - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max()), (u, ucnt[u])


@pytest.mark.parametrize("eig", ["ql", "jacobi"])
def test_c5_shard_properties(eig, monkeypatch):
    """A quarter of `bench.py --workload c5shard` (BASELINE config 5's shape per GPU: k = 256 + biases in SINGLE precision, 512
    dense side-information columns on both sides, Cholesky, 20 entries per user): 390,625 users x 31,250 items, 7.8 M entries --
    the size at which the dispatch of the benchmark is taken, not just its instantiations: the automatic low-rank selection for
    the users (asserted), the eigen-decomposition (eig_kernels.hpp: tridiagonalisation + QL; the one-workgroup Jacobi kernel when forced),
    the 17-tile 16-wave row kernel for the items, 64-bit offsets, the library GEMMs at production width.  Sampled user rows
    (s <= 128 entries: low-rank path) and item rows satisfy the collective normal equations in float64 arithmetic
        (sum_j b_j b_j^T + w C^T C (+) 0 + lam max(n_row, 1) I) a = w C^T u_row (+) 0 + sum_j (x_j - bias_j) b_j
    (collective_closed_form_block, collective.c:1534-1846) to 2e-4, two runs are bit-identical, C and D are finite."""
    import bench
    from cmfrec_amd.session import AlsSession
    if eig == "jacobi":
        monkeypatch.setenv("CMFREC_HIP_EIG", "jacobi")
    sc = 0.25
    m, n, nnz = int(1_562_500 * sc), int(125_000 * sc), int(31_250_000 * sc)
    k, p, q = 256, 512, 512
    row, col, _ = bench.synth_block(m, n, nnz, seed=5)
    rng = np.random.default_rng(5)
    val = (0.5 * rng.integers(1, 11, nnz)).astype(np.float32); val -= val.mean()
    U = rng.standard_normal((m, p), dtype=np.float32)
    II = rng.standard_normal((n, q), dtype=np.float32)
    lam, w = 0.05, 1.0
    A0 = rng.standard_normal((m, k), dtype=np.float32) * np.float32(0.1)
    B0 = rng.standard_normal((n, k), dtype=np.float32) * np.float32(0.1)
    bA0 = rng.standard_normal(m).astype(np.float32) * np.float32(0.1)
    bB0 = rng.standard_normal(n).astype(np.float32) * np.float32(0.1)
    C0 = rng.standard_normal((p, k), dtype=np.float32) * np.float32(0.05)
    D0 = rng.standard_normal((q, k), dtype=np.float32) * np.float32(0.05)

    def run():
        s = AlsSession(m, n, k, implicit=False, dtype=np.float32, lam=lam, use_cg=False, user_bias=True, item_bias=True, scale_lam=True,
                       p=p, m_u=m, q=q, n_i=n)
        s.set_X_coo(row, col, val)
        s.set_sideinfo(U=U, II=II)
        s.set_factors(A=A0, B=B0, biasA=bA0, biasB=bB0, Cm=C0, Dm=D0)
        s.update("B"); s.after_gather("B")
        fB = s.get_factors()
        s.update("A"); s.after_gather("A")
        info = s.lowrank_info()
        fA = s.get_factors()
        s.update("C"); s.update("D")
        fC = s.get_factors()
        return fB, fA, fC, info

    fB, fA, fC, info = run()
    # the users go through the low-rank kernels by themselves (>= 32 k qualifying rows), on the eigen-decomposition asked for
    ucnt = np.bincount(row, minlength=m)
    assert info[0] >= 32768 and info[0] == int((ucnt <= 128).sum()), info
    assert info[1] == (2 if eig == "jacobi" else 3), info
    assert np.isfinite(fC["C"]).all() and np.isfinite(fC["D"]).all() and np.isfinite(fA["A"]).all() and np.isfinite(fB["B"]).all()
    kt = k + 1
    f64 = lambda a: np.asarray(a, np.float64)
    # ---- item rows after the B-step (17-tile row kernel; the heaviest are sliced) ----
    B, biasB = f64(fB["B"]), f64(fB["biasB"])
    ccnt = np.bincount(col, minlength=n)
    order = np.argsort(col, kind="stable"); ptr = np.concatenate([[0], np.cumsum(ccnt)])
    Ab = np.concatenate([f64(A0), np.ones((m, 1))], axis=1)
    DtD = w * f64(D0).T @ f64(D0)
    items = np.concatenate([np.argsort(-ccnt)[:3], rng.choice(np.nonzero(ccnt > 0)[0], 12, replace=False)])
    for i in items:
        e = order[ptr[i]:ptr[i + 1]]
        Aj = Ab[row[e]]; x = f64(val[e]) - f64(bA0)[row[e]]
        M = Aj.T @ Aj + lam * (ccnt[i] + 0) * np.eye(kt)
        M[:k, :k] += DtD
        rhs = Aj.T @ x
        rhs[:k] += w * (f64(II[i]) @ f64(D0))
        sol = np.concatenate([B[i], [biasB[i]]])
        assert np.abs(M @ sol - rhs).max() <= 2e-4 * max(1.0, np.abs(rhs).max(), np.abs(M).max() * np.abs(sol).max()), (i, ccnt[i])
    # ---- user rows after the A-step (low-rank path for s <= 128, full factorisation above) ----
    A, biasA = f64(fA["A"]), f64(fA["biasA"])
    uorder = np.argsort(row, kind="stable"); uptr = np.concatenate([[0], np.cumsum(ucnt)])
    Bb = np.concatenate([B, np.ones((n, 1))], axis=1)
    CtC = w * f64(C0).T @ f64(C0)
    light = np.nonzero((ucnt > 0) & (ucnt <= 128))[0]
    users = np.concatenate([np.argsort(-ucnt)[:3], rng.choice(light, 24, replace=False), np.nonzero(ucnt == 0)[0][:2]])
    for u in users:
        e = uorder[uptr[u]:uptr[u + 1]]
        Bj = Bb[col[e]]; x = f64(val[e]) - biasB[col[e]]
        M = Bj.T @ Bj + lam * max(ucnt[u], 1) * np.eye(kt)             # rows without entries: multiplier 1 (collective.c:1285-1355)
        M[:k, :k] += CtC
        rhs = Bj.T @ x
        rhs[:k] += w * (f64(U[u]) @ f64(C0))
        sol = np.concatenate([A[u], [biasA[u]]])
        assert np.abs(M @ sol - rhs).max() <= 2e-4 * max(1.0, np.abs(rhs).max(), np.abs(M).max() * np.abs(sol).max()), (u, ucnt[u])
    # ---- bit reproducibility of the whole sequence ----
    gB, gA, gC, _ = run()
    for a, b in ((fB["B"], gB["B"]), (fA["A"], gA["A"]), (fC["C"], gC["C"]), (fC["D"], gC["D"])):
        assert np.array_equal(a, b)


def test_c5_rank_true_share():
    """ONE rank of BASELINE config 5 on 8 GPUs at its TRUE share: rank 0's shard built as
    GpuEngine.from_collective_block builds it -- the 100 M x 257 replica of A (102.8 GB), 12.5 M users with 250 M entries, its
    nnz-balanced item block with the entries of all eight user blocks, 25.6 GB of U drawn on the device -- through
    tools/microbench/c5_rank_of_n.py.  Size-independent properties: the factors are finite, sampled user rows (the three heaviest
    and random ones) satisfy the collective normal equations in float64 arithmetic to 2e-4, and one iteration run twice from the
    same state is bit-identical.  Needs most of the part's 288 GB; skipped where less than 230 GB are free."""
    import importlib.util
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 230e9:
        pytest.skip("needs 230 GB of free HBM, %.0f GB are free" % (free / 1e9))
    spec = importlib.util.spec_from_file_location("c5_rank_of_n", os.path.join(ROOT, "tools", "microbench", "c5_rank_of_n.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        res = mod.run(N=8, scale=1.0, timed_iters=1, n_check=12, verbose=False)
    finally:
        os.chdir(cwd)
    assert res["users_in_block"] == 12_500_000 and res["replica_rows"]["A"] == 100_000_000
    assert res["lowrank_rows"] >= 12_000_000                          # 20 entries per user: the low-rank path takes (nearly) all of them
    c = res["checks"]
    assert c["finite"] and c["reproducible"]
    assert c["user_rows_normal_equations_relres"] <= 2e-4, c
    assert res["memory_GB"]["high_water_used"] < 280
