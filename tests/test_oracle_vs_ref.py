"""CPU: the oracle against the REAL reference run live (oracle/_ref, built from /root/reference by
oracle/Makefile) on fresh seeded problems, both precisions, 1 vs many threads.  Skipped where the
reference library is not available (it is built in the build container and shipped to the GPU box
as a binary; /root/reference itself never travels)."""
import numpy as np
import pytest

from conftest import make_coo, rel_err
from oracle.bindings import Reference, ref_available

pytestmark = pytest.mark.skipif(not ref_available(np.float64), reason="oracle/_ref not built")
TOL = {np.float64: 1e-11, np.float32: 2e-4}


@pytest.fixture(scope="module")
def refs():
    return {np.float64: Reference(np.float64), np.float32: Reference(np.float32)}


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_operators_live(oracles, refs, dtype):
    O, R = oracles[dtype], refs[dtype]
    m, n, k = 260, 170, 50
    row, col, val = make_coo(m, n, 5000, 31, dtype=dtype, heavy_row=(4, 120), empty_rows=(6,))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    csr_r, _ = R.coo_to_csr_and_csc(row, col, val, m, n)
    for a, b in zip(csr, csr_r):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(5)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B = (rng.standard_normal((n, k)) * 0.3).astype(dtype)
    for mode in ("cg", "pcg", "chol"):
        kw = dict(use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
        Ao, Ar = A0.copy(), A0.copy()
        O.optimizeA_implicit(Ao, B, csr, 5.0, nthreads=3, **kw)
        R.optimizeA_implicit(Ar, B, csr, 5.0, nthreads=2, **kw)
        assert rel_err(Ao, Ar) < TOL[dtype], mode
        Ao, Ar = A0.copy(), A0.copy()
        kw.update(k=k - 1, lam_last=0.3, scale_lam=True)
        O.optimizeA_explicit(Ao, B, csr, 0.05, nthreads=3, **kw)
        R.optimizeA(Ar, B, csr=csr, lam=0.05, nthreads=2, **kw)
        assert rel_err(Ao, Ar) < TOL[dtype], mode


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_determinism_across_threads(refs, dtype):
    """The reference is bit-reproducible across nthreads for a fixed start."""
    R = refs[dtype]
    m, n, k = 200, 150, 16
    row, col, val = make_coo(m, n, 4000, 33, dtype=dtype)
    rng = np.random.default_rng(6)
    A0 = (rng.standard_normal((m, k)) * 0.01).astype(dtype)
    outs = []
    for nt in (1, 4):
        A, B = A0.copy(), np.zeros((n, k), dtype)
        R.fit_collective_implicit_als(A, B, row, col, val, k, lam=3.0, niter=3, nthreads=nt)
        outs.append((A, B))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fits_live(oracles, refs, dtype):
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-10 if dtype is np.float64 else 2e-3
    m, n, k = 300, 200, 12
    rng = np.random.default_rng(7)
    row, col, val = make_coo(m, n, 6000, 34, counts=False, dtype=dtype)
    for mode, ub, ib in (("cg", True, True), ("chol", True, False), ("cg", False, True)):
        A0 = (rng.standard_normal((m, k)) * 0.01).astype(dtype)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        kw = dict(lam=0.05, scale_lam=True, niter=3, use_cg=mode != "chol", finalize_chol=False, user_bias=ub, item_bias=ib)
        Ao, Bo, Ar, Br = A0.copy(), np.zeros((n, k), dtype), A0.copy(), np.zeros((n, k), dtype)
        ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), nthreads=2, **kw)
        assert ro["ret"] == 0 and rr["ret"] == 0
        assert rel_err(Ao, Ar) < tol and rel_err(Bo, Br) < tol, (mode, ub, ib)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fit_implicit_sideinfo_live(oracles, refs, dtype):
    """Implicit model with dense side information, Cholesky (optimizeA_collective_implicit,
    collective.c:5971-6244): k_user / k_item / k_main, m_u < m, n_i < n, w_main folding, one side only."""
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-10 if dtype is np.float64 else 2e-3
    m, n, k = 120, 90, 6
    row, col, val = make_coo(m, n, 1500, 3, counts=True, dtype=dtype, empty_rows=(4, 110))
    rng = np.random.default_rng(7)
    for ku, ki, km, m_u, n_i, wm in ((0, 0, 0, 120, 90, 1.0), (2, 3, 1, 100, 90, 0.5), (0, 2, 0, None, 70, 1.0)):
        U = None if m_u is None else rng.standard_normal((m_u, 5)).astype(dtype)
        II = rng.standard_normal((n_i, 4)).astype(dtype)
        kua = ku if U is not None else 0
        A0 = (rng.standard_normal((m, kua + k + km)) * 0.1).astype(dtype)
        B0 = (rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
        kw = dict(lam=2.0, alpha=1.5, niter=3, use_cg=False, k_main=km, k_user=kua, k_item=ki, w_main=wm, w_user=3.0,
                  w_item=0.7, U=U, II=II)
        a1, b1, a2, b2 = A0.copy(), B0.copy(), A0.copy(), B0.copy()
        r1 = R.fit_collective_implicit_als(a1, b1, row, col, val, k, nthreads=2, **kw)
        r2 = O.fit_implicit_als_sideinfo(a2, b2, row, col, val, k, nthreads=2, **kw)
        assert r1["ret"] == 0 and r2["ret"] == 0
        assert rel_err(a2, a1) < tol and rel_err(b2, b1) < tol and rel_err(r2["D"], r1["D"]) < tol
        if U is not None:
            assert rel_err(r2["C"], r1["C"]) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pcg", [False, True])
def test_fit_block_cg_live(oracles, refs, dtype, pcg):
    """Side information with use_cg=True: collective_block_cg (explicit, collective.c:2134-2903) and
    collective_block_cg_implicit (:2905-3303), CG and Jacobi-PCG, against the compiled reference."""
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-10 if dtype is np.float64 else 5e-3
    m, n, k = 120, 90, 6
    rng = np.random.default_rng(7)
    for ku, ki, km, m_u, n_i, wm in ((0, 0, 0, 120, 90, 1.0), (2, 3, 1, 100, 90, 0.5), (0, 2, 0, None, 70, 1.0)):
        U = None if m_u is None else rng.standard_normal((m_u, 5)).astype(dtype)
        II = rng.standard_normal((n_i, 4)).astype(dtype)
        kua = ku if U is not None else 0
        A0 = (rng.standard_normal((m, kua + k + km)) * 0.1).astype(dtype)
        B0 = (rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
        C0 = None if U is None else (rng.standard_normal((5, kua + k)) * 0.1).astype(dtype)
        D0 = (rng.standard_normal((4, ki + k)) * 0.1).astype(dtype)
        cp = lambda x: None if x is None else x.copy()
        row, col, val = make_coo(m, n, 1500, 3, counts=True, dtype=dtype, empty_rows=(4, 110))
        kw = dict(lam=2.0, alpha=1.5, niter=3, use_cg=True, max_cg_steps=3, precondition_cg=pcg, k_main=km, k_user=kua,
                  k_item=ki, w_main=wm, w_user=3.0, w_item=0.7, U=U, II=II)
        a1, b1, a2, b2 = A0.copy(), B0.copy(), A0.copy(), B0.copy()
        r1 = R.fit_collective_implicit_als(a1, b1, row, col, val, k, nthreads=2, Cm=cp(C0), Dm=cp(D0), **kw)
        r2 = O.fit_implicit_als_sideinfo(a2, b2, row, col, val, k, nthreads=2, Cm=cp(C0), Dm=cp(D0), **kw)
        assert r1["ret"] == 0 and r2["ret"] == 0
        assert rel_err(a2, a1) < tol and rel_err(b2, b1) < tol and rel_err(r2["D"], r1["D"]) < tol, ("implicit", ku, ki, km)
        row, col, val = make_coo(m, n, 1500, 4, counts=False, dtype=dtype, empty_rows=(4, 110))
        kw = dict(lam=0.3, niter=3, use_cg=True, max_cg_steps=3, precondition_cg=pcg, finalize_chol=False, k_main=km,
                  k_user=kua, k_item=ki, w_user=3.0, w_item=0.7, U=U, II=II, scale_lam=True, scale_lam_sideinfo=ku > 0)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        a1, b1, a2, b2 = A0.copy(), B0.copy(), A0.copy(), B0.copy()
        r1 = R.fit_collective_explicit_als(a1, b1, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), nthreads=2,
                                           Cm=cp(C0), Dm=cp(D0), **kw)
        r2 = O.fit_explicit_als(a2, b2, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), nthreads=2, Cm=cp(C0),
                                Dm=cp(D0), **kw)
        assert r1["ret"] == 0 and r2["ret"] == 0
        assert rel_err(a2, a1) < tol and rel_err(b2, b1) < tol and rel_err(r2["biasA"], r1["biasA"]) < tol, ("explicit", ku, ki, km)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sideinfo_beyond_X_live(oracles, refs, dtype):
    """Side information that covers more users / items than X has rows / columns (m_u > m, n_i > n; the m != m_u
    split of optimizeA_collective, collective.c:4832-5101, and its implicit counterpart :6037-6054, 6073-6090):
    A, B have max(m, m_u) / max(n, n_i) rows; explicit model: the extra rows are a separate dense solve on the side
    information with the scale_lam multiplier p, their biases are zero (:8296, :8923)."""
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-10 if dtype is np.float64 else 5e-3
    m, n, k, m_u, n_i = 120, 90, 6, 135, 100
    rng = np.random.default_rng(7)
    for cg, pcg, ku, ki, km, sl, sls in ((False, False, 0, 0, 0, True, False), (True, False, 2, 3, 1, True, True),
                                         (True, True, 0, 0, 0, False, False)):
        U = rng.standard_normal((m_u, 5)).astype(dtype); II = rng.standard_normal((n_i, 4)).astype(dtype)
        A0 = (rng.standard_normal((m_u, ku + k + km)) * 0.1).astype(dtype)
        B0 = (rng.standard_normal((n_i, ki + k + km)) * 0.1).astype(dtype)
        C0 = (rng.standard_normal((5, ku + k)) * 0.1).astype(dtype); D0 = (rng.standard_normal((4, ki + k)) * 0.1).astype(dtype)
        row, col, val = make_coo(m, n, 1500, 3, counts=True, dtype=dtype, empty_rows=(4, 110))
        kw = dict(lam=2.0, alpha=1.5, niter=3, use_cg=cg, precondition_cg=pcg, max_cg_steps=3, k_main=km, k_user=ku, k_item=ki,
                  w_main=0.5, w_user=3.0, w_item=0.7, U=U, II=II, m=m, n=n)
        a1, b1, a2, b2 = A0.copy(), B0.copy(), A0.copy(), B0.copy()
        r1 = R.fit_collective_implicit_als(a1, b1, row, col, val, k, nthreads=2, Cm=C0.copy(), Dm=D0.copy(), **kw)
        r2 = O.fit_implicit_als_sideinfo(a2, b2, row, col, val, k, nthreads=2, Cm=C0.copy(), Dm=D0.copy(), **kw)
        assert r1["ret"] == 0 and r2["ret"] == 0
        assert rel_err(a2, a1) < tol and rel_err(b2, b1) < tol and rel_err(r2["C"], r1["C"]) < tol, ("implicit", cg, pcg)
        row, col, val = make_coo(m, n, 1500, 4, counts=False, dtype=dtype, empty_rows=(4, 110))
        kw = dict(lam=0.3, niter=3, use_cg=cg, precondition_cg=pcg, max_cg_steps=3, finalize_chol=False, k_main=km, k_user=ku,
                  k_item=ki, w_user=3.0, w_item=0.7, U=U, II=II, scale_lam=sl, scale_lam_sideinfo=sls, m=m, n=n)
        bA = (rng.standard_normal(m_u) * 0.1).astype(dtype); bB = (rng.standard_normal(n_i) * 0.1).astype(dtype)
        a1, b1, a2, b2 = A0.copy(), B0.copy(), A0.copy(), B0.copy()
        r1 = R.fit_collective_explicit_als(a1, b1, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), nthreads=2,
                                           Cm=C0.copy(), Dm=D0.copy(), **kw)
        r2 = O.fit_explicit_als(a2, b2, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), nthreads=2, Cm=C0.copy(),
                                Dm=D0.copy(), **kw)
        assert r1["ret"] == 0 and r2["ret"] == 0
        assert rel_err(a2, a1) < tol and rel_err(b2, b1) < tol, ("explicit", cg, pcg)
        assert np.abs(r2["biasA"] - r1["biasA"]).max() < tol and np.abs(r2["biasB"] - r1["biasB"]).max() < tol
        assert not r1["biasA"][m:].any() and not r1["biasB"][n:].any()          # the reference's own behaviour


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_new_rows_live(oracles, refs, dtype):
    """factors_collective_{explicit,implicit}_multiple live (collective.c:10865-11340) on a problem of another seed
    than the committed fixture, 1 vs 3 threads."""
    import golden_cases as gc
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-11 if dtype is np.float64 else 2e-4
    for k in (5, 33):
        d = gc.new_rows_problem(dtype, k, seed=29)
        for name, kind, kw in gc.new_rows_cases(d):
            a1, b1 = gc.run_new_rows(R, kind, dict(kw, nthreads=3))
            a2, b2 = gc.run_new_rows(O, kind, dict(kw, nthreads=1))
            assert not np.isnan(a1).any(), name
            assert gc.maxrel(a2, a1) < tol, (name, k)
            if b1 is not None:
                assert gc.maxrel(b2, b1) < tol, (name, k)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_sideinfo_live(oracles, refs, dtype):
    """Collective half-step with SPARSE side information (U as CSR, missing = absent): optimizeA_collective general
    branch (collective.c:5566-5968 -> :1223-1847) and optimizeA_collective_implicit (:5971-6244 -> :1849-2131), Cholesky.
    Rows with observations but no attributes, attributes but no observations, neither, and rows beyond m_u."""
    O, R = oracles[dtype], refs[dtype]
    tol = 1e-11 if dtype is np.float64 else 2e-4
    rng = np.random.default_rng(4)
    m, n, k, p, m_u = 90, 70, 7, 11, 80
    for ku, ki, km in ((0, 0, 0), (2, 1, 1)):
        B = rng.standard_normal((n, ki + k + km)).astype(dtype); Cm = rng.standard_normal((p, ku + k)).astype(dtype)
        ur, uc, _ = make_coo(m_u, p, 260, 12, counts=False, dtype=dtype, empty_rows=(3, 5))
        uv = rng.standard_normal(len(ur)).astype(dtype)
        ucsr, _ = R.coo_to_csr_and_csc(ur, uc, uv, m_u, p)
        for counts in (False, True):
            row, col, val = make_coo(m, n, 900, 11, counts=counts, dtype=dtype, heavy_row=(9, 50), empty_rows=(3, 7, 85))
            csr, _ = R.coo_to_csr_and_csc(row, col, val, m, n)
            A0 = rng.standard_normal((m, ku + k + km)).astype(dtype)
            if counts:
                a1, a2 = A0.copy(), A0.copy()
                R.optimizeA_collective_implicit_sparse(a1, B, Cm, csr, ucsr, 0.7, w_user=2.5, k=k, k_main=km, k_user=ku,
                                                       k_item=ki, nthreads=2)
                O.optimizeA_collective_sparse(a2, B, Cm, csr, ucsr, 0.7, w_user=2.5, k=k, k_main=km, k_user=ku, k_item=ki,
                                              implicit=True, nthreads=1)
                assert rel_err(a2, a1) < tol, ("implicit", ku)
                assert not a1[3].any() and not a1[85].any()
                continue
            for sl, sls in ((False, False), (True, False), (True, True)):
                a1, a2 = A0.copy(), A0.copy()
                kw = dict(w_user=2.5, lam_last=1.3, k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=sl, scale_lam_sideinfo=sls)
                R.optimizeA_collective(a1, B, Cm, csr, None, 0.7, U_csr=ucsr, nthreads=2, **kw)
                O.optimizeA_collective_sparse(a2, B, Cm, csr, ucsr, 0.7, nthreads=1, **kw)
                assert rel_err(a2, a1) < tol, ("explicit", ku, sl, sls)
                for pcg in (False, True):
                    a1, a2 = A0.copy(), A0.copy()
                    R.optimizeA_collective(a1, B, Cm, csr, None, 0.7, U_csr=ucsr, nthreads=2, use_cg=True, precondition_cg=pcg, **kw)
                    O.optimizeA_collective_sparse(a2, B, Cm, csr, ucsr, 0.7, nthreads=1, use_cg=True, precondition_cg=pcg, **kw)
                    assert rel_err(a2, a1) < tol, ("explicit cg", ku, sl, sls, pcg)
                    assert not a1[3].any() and np.array_equal(a1[85], A0[85])      # no data at all: zeros; beyond m_u: untouched


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fit_sparse_sideinfo_live(oracles, refs, dtype):
    """Whole fits with sparse side information on one or both sides (collective.c:7263-10207, Cholesky updates):
    C / D as optimizeA Case 4 on the attributes' CSC, A / B with the row's attributes as extra rank-1 terms."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.sparse_sideinfo_problem(dtype, seed=43)
    for name, implicit, which, sl, sls in gc.SPARSE_SIDE_CASES:
        exp = gc.sparse_sideinfo_reference(refs[dtype], d, implicit, which, sl, sls, nthreads=3)
        got = gc.sparse_sideinfo_oracle(oracles[dtype], d, implicit, which, sl, sls, nthreads=1)
        assert gc.compare_fits(got, exp) < tol, name
    # block CG / PCG with the attributes as a second gathered term (collective.c:2134-3303, u_vec_sp branches)
    for name, implicit, which, sl, sls, solver in gc.SPARSE_SIDE_CG_CASES:
        exp = gc.sparse_sideinfo_reference(refs[dtype], d, implicit, which, sl, sls, nthreads=3, solver=solver)
        got = gc.sparse_sideinfo_oracle(oracles[dtype], d, implicit, which, sl, sls, nthreads=1, solver=solver)
        assert gc.compare_fits(got, exp) < tol, name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nonneg_live(oracles, refs, dtype):
    """Non-negative factors: solve_nonneg (common.c:2131-2179) in place of the Cholesky solves of A / B (nonneg) and
    C / D (nonneg_C / nonneg_D), CG switched off (collective.c:7474-7479, :9513-9517); sweep limit honoured."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.nonneg_problem(dtype, seed=53)
    for name, implicit, side, opts in gc.NONNEG_CASES:
        exp = gc.nonneg_reference(refs[dtype], d, implicit, side, opts, nthreads=3)
        got = gc.nonneg_oracle(oracles[dtype], d, implicit, side, opts, nthreads=1)
        assert gc.compare_fits(got, exp) < tol, name
        if opts.get("nonneg"):
            assert (exp["A"] >= 0).all() and (exp["B"] >= 0).all() and (exp["A"] == 0).mean() > 0.2


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_implicit_features_live(oracles, refs, dtype):
    """add_implicit_features: the Ai / Bi updates (optimizeA Case 3 on the binary indicator, collective.c:8448-8534) and the
    extra term of the A / B updates (:1704-1707, :1757-1771), with and without side information, biases, scaled lambda."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.nonneg_problem(dtype, seed=57)
    for name, side, opts in gc.IMPLICIT_FEATS_CASES:
        exp = gc.implicit_feats_reference(refs[dtype], d, side, opts, nthreads=3)
        got = gc.implicit_feats_oracle(oracles[dtype], d, side, opts, nthreads=1)
        assert np.abs(exp["Ai"]).sum() > 0 and gc.compare_fits(got, exp) < tol, name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lam_unique_live(oracles, refs, dtype):
    """lam_unique / l1_lam_unique: A / B take [2] / [3] with [0] / [1] on a fitted bias (collective.c:8649-8654, :8820-8825),
    C / D [4] / [5] (:8367, :8418), Bi / Ai [3] / [2] (:8469, :8510); the implicit model ignores the bias slots."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.nonneg_problem(dtype, seed=59)
    for name, implicit, side, opts in gc.LAM_UNIQUE_CASES:
        exp = gc.lam_unique_reference(refs[dtype], d, implicit, side, opts, nthreads=3)
        got = gc.lam_unique_oracle(oracles[dtype], d, implicit, side, opts, nthreads=1)
        assert gc.compare_fits(got, exp) < tol, name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nan_side_info_live(oracles, refs, dtype):
    """Dense side information with NaN: the reference centres the present entries (common.c:4938-4997) and uses only those
    (collective.c:1566-1653; optimizeA Case 2 for C / D) -- identical to its sparse route on the centred values, which is how
    the product runs it.  Sparse and nearly complete matrices, Cholesky and CG."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.nan_side_problem(dtype, seed=67)
    for name, implicit, which, sl, sls, solver in gc.NAN_SIDE_CASES:
        exp = gc.nan_side_reference(refs[dtype], d, implicit, which, sl, sls, nthreads=3, solver=solver)
        got = gc.nan_side_oracle(oracles[dtype], d, implicit, which, sl, sls, nthreads=1, solver=solver)
        assert gc.compare_fits(got, exp) < tol, name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bias_start_values_count_side_information(oracles, refs, dtype):
    """Bias initialisation under scale_lam_sideinfo with dense side information: rows / columns that have attributes scale
    lambda by (entries + attributes) (wsumA / wsumB, collective.c:8071-8104 into common.c:4655-4665, :4812-4823)."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-4
    d = gc.nonneg_problem(dtype, seed=71)
    for sls, side in ((True, True), (True, False), (False, True)):
        U, II = (d["U"][:120], d["I"]) if side else (None, None)
        kw = dict(lam=0.3, U=U, II=II, nthreads=1, use_cg=False, finalize_chol=False, scale_lam=not sls, scale_lam_sideinfo=sls)
        r0 = refs[dtype].fit_collective_explicit_als(d["A0"].copy(), d["B0"].copy(), d["row"], d["col"], d["ratings"], d["k"], niter=0,
                                                     reset_values=True, seed=3, **kw)
        exp = refs[dtype].fit_collective_explicit_als(d["A0"].copy(), d["B0"].copy(), d["row"], d["col"], d["ratings"], d["k"], niter=2,
                                                      reset_values=True, seed=3, **kw)
        got = oracles[dtype].fit_explicit_als(r0["A"].copy(), r0["B"].copy(), d["row"], d["col"], d["ratings"], d["k"], niter=2,
                                              init_biases=True, **kw)
        assert np.abs(r0["biasA"]).max() > 0 and gc.compare_fits(got, exp) < tol, (sls, side)


SCALE_BIAS_CONST_CASES = [("sl chol", False, dict(scale_lam=True)), ("sl cg", False, dict(scale_lam=True, use_cg=True, finalize_chol=True)),
                          ("sls side chol", True, dict(scale_lam_sideinfo=True)), ("sl side cg m_u<m", True, dict(scale_lam=True, use_cg=True)),
                          ("user bias only", False, dict(scale_lam=True, item_bias=False)),
                          ("item bias only cg", False, dict(scale_lam=True, user_bias=False, use_cg=True))]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_scale_bias_const_live(oracles, refs, dtype):
    """scale_bias_const: the biases' lambda takes one constant factor -- the mean over the rows of (entries + attributes counted
    under scale_lam_sideinfo), collective.c:8026-8048, :8071-8160 -- while rows solved without side information stop scaling it
    by their own count (common.c:679-723); the bias start values still multiply by the row's count (:4655-4665)."""
    import golden_cases as gc
    tol = 1e-11 if dtype is np.float64 else 2e-3          # whole fits, CG among them: the single-precision fit tolerance
    d = gc.nonneg_problem(dtype, seed=73)
    for name, side, o in SCALE_BIAS_CONST_CASES:
        o = dict(o)
        U, II = (d["U"][:120], d["I"]) if side else (None, None)
        kw = dict(lam=0.3, U=U, II=II, nthreads=1, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
        args = (d["row"], d["col"], d["ratings"], d["k"])
        r0 = refs[dtype].fit_collective_explicit_als(d["A0"].copy(), d["B0"].copy(), *args, niter=0, reset_values=True, seed=3,
                                                     scale_bias_const=True, **kw)
        exp = refs[dtype].fit_collective_explicit_als(d["A0"].copy(), d["B0"].copy(), *args, niter=2, reset_values=True, seed=3,
                                                      scale_bias_const=True, **kw)
        oracles[dtype].set_scale_bias_const(True)
        try:
            both = kw.get("user_bias", True) == kw.get("item_bias", True)
            got = oracles[dtype].fit_explicit_als(r0["A"].copy(), r0["B"].copy(), *args, niter=2, init_biases=both,
                                                  biasA=r0["biasA"].copy(), biasB=r0["biasB"].copy(), **kw)
        finally:
            oracles[dtype].set_scale_bias_const(False)
        assert max(exp["scaling_biasA"], exp["scaling_biasB"]) > 1 and gc.compare_fits(got, exp) < tol, name


# ---- observation weights (explicit model): weighted row solvers, weighted mean / bias start values, sums of weights under scale_lam
WEIGHT_CASES = [
    ("cg", dict(use_cg=True, finalize_chol=False)),
    ("cg scale_lam finalize", dict(use_cg=True, finalize_chol=True, scale_lam=True)),
    ("pcg scale_lam", dict(use_cg=True, precondition_cg=True, finalize_chol=False, scale_lam=True)),
    ("chol", dict(use_cg=False)),
    ("chol scale_lam no bias", dict(use_cg=False, scale_lam=True, user_bias=False, item_bias=False)),
    ("cg no centring k_main", dict(use_cg=True, finalize_chol=False, center=False, k_main=2)),
]


def weights_problem(dtype, seed=71, m=310, n=190, nnz=6000):
    row, col, val = make_coo(m, n, nnz, seed, counts=False, dtype=dtype, heavy_row=(3, 150), empty_rows=(5, 17))
    # entries ordered by column: the reference hands its B-step the weights in COO order where the CSC order is meant
    # (collective.c:8642, :8689 pass `weight`, not `weightC`, for sparse X -- its wsumB and bias start values do use weightC),
    # so the two only mean the same thing for input sorted by column, which is also what its documentation asks for
    # (cmfrec/__init__.py:3095-3099).  The oracle and the HIP path use the CSC-ordered weights throughout.
    o = np.argsort(col, kind="stable")
    row, col, val = row[o], col[o], val[o]
    rng = np.random.default_rng(seed + 1)
    w = (0.25 + 2.0 * rng.random(len(val))).astype(dtype)
    return m, n, row, col, val, w


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_weighted_operators_live(oracles, refs, dtype):
    """optimizeA Case 4 with weights (common.c:3268-3299): CG, PCG and Cholesky rows; lambda scaled by the driver's sums of
    weights and by the row's own sum."""
    O, R = oracles[dtype], refs[dtype]
    m, n, row, col, val, w = weights_problem(dtype)
    k = 24
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    csr_w, _ = O.coo_to_csr_and_csc(row, col, w, m, n)
    wR = csr_w[2]
    wsum = np.array([wR[int(csr[0][r]):int(csr[0][r + 1])].astype(np.float64).sum() if csr[0][r + 1] > csr[0][r] else 1.0
                     for r in range(m)]).astype(dtype)
    rng = np.random.default_rng(9)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B = (rng.standard_normal((n, k)) * 0.3).astype(dtype)
    for mode in ("cg", "pcg", "chol"):
        for ws in (None, wsum):
            kw = dict(use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3, k=k - 1, lam_last=0.3, scale_lam=True)
            Ao, Ar = A0.copy(), A0.copy()
            O.optimizeA_explicit(Ao, B, csr, 0.05, nthreads=3, weight=wR, wsum=ws, **kw)
            R.optimizeA(Ar, B, csr=csr, lam=0.05, nthreads=2, weight=wR, wsum=ws, **kw)
            assert rel_err(Ao, Ar) < TOL[dtype], (mode, ws is None)
            Au = A0.copy()
            O.optimizeA_explicit(Au, B, csr, 0.05, nthreads=3, **kw)
            assert rel_err(Au, Ar) > 1e-3, "the weights must matter"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_weighted_fit_live(oracles, refs, dtype):
    """fit_collective_explicit_als with weight != NULL against the oracle's restatement: weighted mean, weightR / weightC,
    wsumA / wsumB, weighted two-sided bias start values, weighted row solvers."""
    O, R = oracles[dtype], refs[dtype]
    m, n, row, col, val, w = weights_problem(dtype, seed=73)
    k = 12
    rng = np.random.default_rng(10)
    for name, o in WEIGHT_CASES:
        o = dict(o)
        km = o.get("k_main", 0)
        A0 = (rng.standard_normal((m, k + km)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k + km)) * 0.1).astype(dtype)
        ub, ib = o.get("user_bias", True), o.get("item_bias", True)
        ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=3, nthreads=2, weight=w,
                                init_biases=ub and ib, **o)
        assert ro["ret"] == 0, name
        # the reference initialises the biases itself only with reset_values (its own random start): hand the oracle's over
        rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=3, nthreads=2, weight=w, **o)
        if ub and ib:
            # the reference computes bias start values only together with its own random start (reset_values): compare them
            # after zero iterations, the iterations themselves through the seeded fits of tests/test_gpu_fit.py
            r0 = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=0, nthreads=2, weight=w,
                                               reset_values=True, seed=3, **o)
            o0 = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=0, nthreads=2, weight=w,
                                    init_biases=True, **o)
            assert r0["ret"] == 0 and o0["ret"] == 0, name
            for key in ("biasA", "biasB"):
                assert np.abs(o0[key]).max() > 0.05 and rel_err(o0[key], r0[key]) < 50 * TOL[dtype], (name, key)
            continue
        assert rr["ret"] == 0, name
        for key in ("A", "B"):
            assert rel_err(ro[key], rr[key]) < 50 * TOL[dtype], (name, key)
        assert abs(ro["glob_mean"] - rr["glob_mean"]) < 1e-5 * max(1.0, abs(rr["glob_mean"]))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_global_mean_eight_threads_live(oracles, refs, dtype):
    """calc_mean_and_center at nthreads >= 8 (common.c:3496-3513, :3561-3571), live: zero iterations from a given start, so that
    the mean (and the bias start values computed from the centred entries) is all that happens."""
    O, R = oracles[dtype], refs[dtype]
    m, n, row, col, val, w = weights_problem(dtype, seed=75)
    k = 6
    rng = np.random.default_rng(12)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    for weight in (None, w):
        got = {}
        for nt in (2, 8):
            ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=2, nthreads=nt, weight=weight, use_cg=True)
            rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=2, nthreads=nt, weight=weight,
                                               use_cg=True)
            assert ro["ret"] == 0 and rr["ret"] == 0
            assert abs(ro["glob_mean"] - rr["glob_mean"]) < (1e-13 if dtype is np.float64 else 1e-6), (weight is None, nt)
            for key in ("A", "B"):
                assert rel_err(ro[key], rr[key]) < 50 * TOL[dtype], (weight is None, nt, key)
            got[nt] = float(rr["glob_mean"])
        if weight is not None:
            assert abs(got[2] - got[8]) > 1e-2          # the two branches are different numbers with weights
        else:
            assert abs(got[2] - got[8]) < 1e-5


def test_reference_weight_defects(refs):
    """Two defects of the reference's weighted fit, shown on the reference alone -- the reason the weighted parity cases use
    entries sorted by column and, for the collective closed form, no centring ("Observation weights").

    1. The B-step is handed the weights in COO order where the CSC order is meant (collective.c:8642, :8689: `weight`, not
       `weightC`): the same entries in another order give another model (unweighted, only the summation order changes).
    2. The collective closed form subtracts (w - 1) x glob_mean from the weighted right-hand side without missing-as-zero
       (collective.c:1744-1753): centring inside the fit differs from fitting the centred data, for that solver only."""
    import golden_cases as gc
    R = refs[np.float64]
    d = gc.weights_problem(np.float64)
    k, row, col, x, w = d["k"], d["row"], d["col"], d["ratings"], d["W"]
    fit = lambda r, c, v, wt, **kw: R.fit_collective_explicit_als(d["A0"].copy(), d["B0"].copy(), r, c, v, k, biasA=d["bA"].copy(),
                                                                  biasB=d["bB"].copy(), lam=0.3, niter=2, nthreads=2, weight=wt, **kw)
    # 1. entry order
    perm = np.random.default_rng(0).permutation(len(x))
    plain = dict(use_cg=False, user_bias=False, item_bias=False, center=False)
    a, b = fit(row, col, x, w, **plain), fit(row[perm], col[perm], x[perm], w[perm], **plain)
    assert rel_err(a["A"], b["A"]) > 1e-2
    a, b = fit(row, col, x, None, **plain), fit(row[perm], col[perm], x[perm], None, **plain)
    assert rel_err(a["A"], b["A"]) < 1e-10
    # 2. centring in the collective closed form
    side = dict(U=d["U"], II=d["I"], user_bias=False, item_bias=False)
    for use_cg, differs in ((False, True), (True, False)):
        a = fit(row, col, x, w, use_cg=use_cg, finalize_chol=False, center=True, **side)
        b = fit(row, col, (x - a["glob_mean"]).astype(x.dtype), w, use_cg=use_cg, finalize_chol=False, center=False, **side)
        e = rel_err(a["A"], b["A"])
        assert (e > 1e-2) if differs else (e < 1e-10), (use_cg, e)


# ---- NA_as_zero for the main matrix (sparse X whose absent entries are zeros), model without side information ----------------
NAZ_CASES = [
    ("chol, biases", dict(use_cg=False)),
    ("cg asked for (closed form all the same), scale_lam", dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("no biases", dict(use_cg=False, user_bias=False, item_bias=False)),
    ("no centring, user bias", dict(use_cg=False, center=False, item_bias=False, scale_lam=True)),
    ("item bias, k_main", dict(use_cg=False, user_bias=False, k_main=2)),
    ("no biases, no centring", dict(use_cg=False, user_bias=False, item_bias=False, center=False)),
]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(310, 190), (150, 230)])
def test_NA_as_zero_fit_live(oracles, refs, dtype, shape):
    """fit_collective_explicit_als with NA_as_zero_X on sparse X: mean over all cells, optimizeA Case 3 (one shared matrix) with
    the bias / mean constant of the right-hand side, rows and columns without entries solved like the others; the bias start
    values (one- and two-sided, missing-as-zero branches) after zero iterations of a seeded reference fit.  Both m > n and
    n > m (the item sweep of the two-sided start values averages the user biases over the wrong bound: see the oracle)."""
    O, R = oracles[dtype], refs[dtype]
    m, n = shape
    row, col, val = make_coo(m, n, 5000, 75, counts=False, dtype=dtype, heavy_row=(3, 120), empty_rows=(5, 17))
    keep = col != 11                                     # a column without entries
    row, col, val = row[keep], col[keep], val[keep]
    k = 12
    rng = np.random.default_rng(11)
    for name, o in NAZ_CASES:
        o = dict(o)
        km = o.get("k_main", 0)
        A0 = (rng.standard_normal((m, k + km)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k + km)) * 0.1).astype(dtype)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=0.4, niter=3, nthreads=2,
                                NA_as_zero_X=True, **o)
        rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=0.4, niter=3,
                                           nthreads=2, NA_as_zero_X=True, **o)
        assert ro["ret"] == 0 and rr["ret"] == 0, name
        keys = ("A", "B") + (("biasA",) if o.get("user_bias", True) else ()) + (("biasB",) if o.get("item_bias", True) else ())
        for key in keys:
            assert rel_err(ro[key], rr[key]) < 100 * TOL[dtype], (name, key)
        assert abs(ro["glob_mean"] - rr["glob_mean"]) < 1e-6 * max(1.0, abs(rr["glob_mean"])), name
        ub, ib = o.get("user_bias", True), o.get("item_bias", True)
        if (ub or ib) and not (ib and not ub and not o["use_cg"]) and not (ub and ib and n > m):
            r0 = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=0, nthreads=2, NA_as_zero_X=True,
                                               reset_values=True, seed=3, **o)
            o0 = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, lam=0.4, niter=0, nthreads=2, NA_as_zero_X=True,
                                    init_biases=True, **o)
            for key in (("biasA",) if ub else ()) + (("biasB",) if ib else ()):
                assert np.abs(o0[key]).max() > 1e-3 and rel_err(o0[key], r0[key]) < 100 * TOL[dtype], (name, key)


# ---- NA_as_zero for the main matrix together with dense side information -------------------------------------------------------
NAZ_SIDE_LIVE = [
    dict(), dict(scale_lam=True), dict(scale_lam_sideinfo=True),
    dict(center=False, item_bias=False, k_user=2, k_item=1, k_main=2, w_user=0.7, w_item=1.3),
    dict(user_bias=False, item_bias=False, center=False), dict(U_only=True), dict(I_only=True, scale_lam=True),
]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(310, 190), (150, 230)])
def test_NA_as_zero_sideinfo_fit_live(oracles, refs, dtype, shape):
    """fit_collective_explicit_als with NA_as_zero_X AND dense side information, closed form: optimizeA_collective's factorised
    shared block matrix (collective.c:5607-5617, :5700-5716), right-hand sides X B + w U C + the bias / mean constant
    (:5753-5770, :5815-5821); side information on one or both sides, k_user / k_item / k_main, both lambda scalings."""
    O, R = oracles[dtype], refs[dtype]
    m, n = shape
    rng = np.random.default_rng(5)
    row, col, val = make_coo(m, n, 5000, 75, counts=False, dtype=dtype, heavy_row=(3, 120), empty_rows=(5, 17))
    keep = col != 11
    row, col, val = row[keep], col[keep], val[keep]
    k, p, q = 10, 7, 5
    U = rng.standard_normal((m, p)).astype(dtype); II = rng.standard_normal((n, q)).astype(dtype)
    for o in NAZ_SIDE_LIVE:
        o = dict(o); U_only = o.pop("U_only", False); I_only = o.pop("I_only", False)
        ku, ki, km = o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
        A0 = (rng.standard_normal((m, ku + k + km)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        kw = dict(U=None if I_only else U, II=None if U_only else II, lam=0.4, niter=3, nthreads=2, NA_as_zero_X=True, use_cg=False, **o)
        ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        assert ro["ret"] == 0 and rr["ret"] == 0, o
        keys = ("A", "B") + (() if I_only else ("C",)) + (() if U_only else ("D",)) + (("biasA",) if o.get("user_bias", True) else ()) + \
               (("biasB",) if o.get("item_bias", True) else ())
        for key in keys:
            assert rel_err(ro[key], rr[key]) < 100 * TOL[dtype], (o, key)
    # use_cg changes nothing (the factorised block matrix is taken before the solver is looked at) ...
    kw = dict(U=U, II=II, lam=0.4, niter=2, nthreads=2, NA_as_zero_X=True)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    rc = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, use_cg=True, finalize_chol=False, **kw)
    oc = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, use_cg=True, finalize_chol=False, **kw)
    assert rc["ret"] == 0 and oc["ret"] == 0 and rel_err(oc["A"], rc["A"]) < 100 * TOL[dtype] and rel_err(oc["B"], rc["B"]) < 100 * TOL[dtype]
    # ... and what the restatement does not cover is refused: side information on fewer rows than X
    A0 = np.zeros((m, k), dtype); B0 = np.zeros((n, k), dtype)
    assert O.fit_explicit_als(A0, B0, row, col, val, k, U=U[:m - 9], niter=1, NA_as_zero_X=True, use_cg=False)["ret"] == 2


# ---- NA_as_zero_U / NA_as_zero_I: sparse side information whose absent entries are zeros ------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [77, 78])
def test_NA_as_zero_UI_fit_live(oracles, refs, dtype, seed):
    """The reference's sparse missing-as-zero branches for the side information (collective.c:1277-1457, :5790-5836; C / D by
    optimizeA Case 3 with the column means as a rank-one correction, :8354-8441) against the restatement -- the dense route on the
    zero-filled matrices, rows with neither an entry of X nor of the side information set to zero -- on fresh problems of G21's
    shape (U on 80 of the 90 users, rows without entries on either side): both models, closed form and CG."""
    import golden_cases as gc
    d = gc.sparse_sideinfo_problem(dtype, seed=seed)
    for name, implicit, which, sl, sls, solver in gc.NAZ_UI_CASES:
        ref = gc.naz_ui_reference(refs[dtype], d, implicit, which, sl, sls, solver)
        got = gc.naz_ui_oracle(oracles[dtype], d, implicit, which, sl, sls, solver)
        for key, v in ref.items():
            if v is not None and np.size(v) > 1:
                assert rel_err(got[key], v) < 100 * TOL[dtype], (name, key)


# ---- NA_as_zero for the main matrix together with observation weights -----------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_weighted_operator_live(oracles, refs, dtype):
    """optimizeA Case 4 with NA_as_zero && weight (common.c:3209-3302): per row the shared B^T B plus the correction of the present
    entries -- closed form (factors_closed_form :846-907), CG (factors_explicit_cg_NA_as_zero_weighted :1293-1441) and PCG
    (:1443-1613); with and without the bias / mean constants, both lambda scalings, rows without entries (solved only when bias_BtX
    is given, :3270-3271)."""
    O, R = oracles[dtype], refs[dtype]
    m, n, k = 230, 140, 12
    row, col, val = make_coo(m, n, 4000, 91, counts=False, dtype=dtype, heavy_row=(4, 100), empty_rows=(6, 19))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(17)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B = (rng.standard_normal((n, k)) * 0.3).astype(dtype)
    wt = (0.25 + 2.0 * rng.random(len(csr[2]))).astype(dtype)
    bias_X = (rng.standard_normal(n) * 0.2).astype(dtype)
    glob = 0.37
    bias_BtX = (-(B * (bias_X + dtype(glob))[:, None]).sum(axis=0)).astype(dtype)
    nnz_row = np.diff(csr[0].astype(np.int64))
    wsum = (np.add.reduceat(np.append(wt, 0), np.minimum(csr[0][:-1].astype(np.int64), len(wt))) * (nnz_row > 0) + (n - nnz_row)).astype(dtype)
    tol = 100 * TOL[dtype]
    for mode in ("chol", "cg", "pcg"):
        for const in (True, False):
            for sl, ws in ((False, None), (True, wsum), (True, None)):
                kw = dict(use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3, scale_lam=sl)
                cst = dict(bias_BtX=bias_BtX, bias_X=bias_X, bias_X_glob=glob) if const else {}
                Ao, Ar = A0.copy(), A0.copy()
                O.optimizeA_naz_weighted(Ao, B, csr, wt, 0.3, lam_last=0.7, wsum=ws, nthreads=2, **kw, **cst)
                R.optimizeA(Ar, B, csr=csr, lam=0.3, lam_last=0.7, weight=wt, wsum=ws, NA_as_zero=True, nthreads=2, **kw, **cst)
                assert rel_err(Ao, Ar) < tol, (mode, const, sl, ws is not None)
                if not const:                                   # rows without entries are left as they were
                    assert np.array_equal(Ao[6], A0[6]) and np.array_equal(Ar[19], A0[19])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_weighted_fit_live(oracles, refs, dtype):
    """fit_collective_explicit_als with NA_as_zero_X AND observation weights, no side information (collective.c:8680-8717,
    :8851-8888: optimizeA Case 4 with the bias / mean constants and wsumA / wsumB = sum of the row's weights + the number of its
    absent entries, :7991-8022): closed form and CG, with and without biases / centring, scale_lam."""
    O, R = oracles[dtype], refs[dtype]
    m, n, row, col, val, wt = weights_problem(dtype, seed=92, m=150, n=110, nnz=3000)     # (entries ordered by column, see there)
    k = 8
    rng = np.random.default_rng(18)
    cases = [dict(use_cg=False), dict(use_cg=True, finalize_chol=False), dict(use_cg=False, scale_lam=True),
             dict(use_cg=True, finalize_chol=False, scale_lam=True, user_bias=False), dict(use_cg=False, center=False, item_bias=False),
             dict(use_cg=True, user_bias=False, item_bias=False, center=False, finalize_chol=True)]
    for o in cases:
        A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        kw = dict(lam=0.4, niter=3, nthreads=2, NA_as_zero_X=True, weight=wt, **o)
        ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        assert ro["ret"] == 0 and rr["ret"] == 0, o
        keys = ("A", "B") + (("biasA",) if o.get("user_bias", True) else ()) + (("biasB",) if o.get("item_bias", True) else ())
        for key in keys:
            assert rel_err(ro[key], rr[key]) < 100 * TOL[dtype], (o, key)
        assert abs(ro["glob_mean"] - rr["glob_mean"]) < 1e-6 * max(1.0, abs(rr["glob_mean"])), o


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_weighted_sideinfo_fit_live(oracles, refs, dtype):
    """... and with DENSE side information on top (closed form): the rows with entries leave the factorised block matrix
    (collective.c:1367-1372 wants weight == NULL || nnz == 0) for the general branch of collective_closed_form_block (:1534-1846);
    rows without entries keep the shared factorisation.  Both sides, one side only, k_user / k_item / k_main, scale_lam /
    scale_lam_sideinfo."""
    O, R = oracles[dtype], refs[dtype]
    m, n, row, col, val, wt = weights_problem(dtype, seed=94, m=120, n=90, nnz=2500)      # (entries ordered by column, see there)
    keep = col != 11                                                                       # a column without entries
    row, col, val, wt = row[keep], col[keep], val[keep], wt[keep]
    k, p, q = 6, 7, 5
    rng = np.random.default_rng(19)
    U = rng.standard_normal((m, p)).astype(dtype); II = rng.standard_normal((n, q)).astype(dtype)
    cases = [("UI", dict()), ("UI", dict(scale_lam=True)), ("UI", dict(scale_lam_sideinfo=True)),
             ("UI", dict(user_bias=False, item_bias=False, center=False)),
             ("UI", dict(center=False, item_bias=False, k_user=2, k_item=1, k_main=2, w_user=0.7, w_item=1.3)),
             ("U", dict()), ("I", dict(scale_lam=True))]
    for sides, o in cases:
        ku, ki, km = o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
        A0 = (rng.standard_normal((m, ku + k + km)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
        bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
        kw = dict(U=U if "U" in sides else None, II=II if "I" in sides else None, lam=0.4, niter=3, nthreads=2, NA_as_zero_X=True,
                  use_cg=False, weight=wt, **o)
        ro = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        rr = R.fit_collective_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), **kw)
        assert ro["ret"] == 0 and rr["ret"] == 0, (sides, o)
        keys = ("A", "B") + (("C",) if "U" in sides else ()) + (("D",) if "I" in sides else ())
        keys += (("biasA",) if o.get("user_bias", True) else ()) + (("biasB",) if o.get("item_bias", True) else ())
        for key in keys:
            assert rel_err(ro[key], rr[key]) < 100 * TOL[dtype], (sides, o, key)
        assert abs(ro["glob_mean"] - rr["glob_mean"]) < 1e-6 * max(1.0, abs(rr["glob_mean"])), o


# ---- NA_as_zero for the main matrix together with SPARSE side information --------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [51, 52])
def test_NA_as_zero_sparse_sideinfo_fit_live(oracles, refs, dtype, seed):
    """fit_collective_explicit_als with NA_as_zero_X and sparse U / I (missing = absent), closed form: collective_closed_form_block's
    general branch with prefer_BtB (collective.c:1534-1846) row by row; both lambda scalings, k_user / k_item / k_main, one side only."""
    import golden_cases as gc
    d = gc.naz_sparse_side_problem(dtype, seed=seed)
    for name, which, opts in gc.NAZ_SPARSE_SIDE_CASES:
        ref = gc.naz_sparse_side_reference(refs[dtype], d, which, opts)
        got = gc.naz_sparse_side_oracle(oracles[dtype], d, which, opts)
        for key, v in ref.items():
            if v is not None and np.size(v) > 1:
                assert rel_err(got[key], v) < 100 * TOL[dtype], (name, key)
        assert abs(got["glob_mean"] - ref["glob_mean"]) < 1e-6 * max(1.0, abs(ref["glob_mean"])), name
    # what the restatement does not cover is refused: CG, side information on fewer rows than X
    O = oracles[dtype]
    A0, B0 = d["A0"][:, d["ku"]:].copy(), d["B0"][:, d["ki"]:].copy()
    assert O.fit_als_sparse_sideinfo(A0, B0, d["row"], d["col"], d["ratings"], d["k"], False, k_main=d["km"], U_coo=d["U_coo"], niter=1,
                                     use_cg=True, NA_as_zero_X=True)["ret"] == 2


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("seed", [95, 96])
def test_NA_as_zero_implicit_features_fit_live(oracles, refs, dtype, seed):
    """fit_collective_explicit_als with NA_as_zero_X and add_implicit_features, no side information, closed form: the half-steps of A
    and B are optimizeA_collective's general branch (collective.c:8612 / :8783 -> :1534-1846 with prefer_BtB), Ai / Bi optimizeA
    Case 3 on the indicator (:8448-8534)."""
    import golden_cases as gc
    d = gc.naz_problem(dtype, seed=seed)
    for name, opts in gc.NAZ_IMPF_CASES:
        ref = gc.naz_impf_reference(refs[dtype], d, opts)
        got = gc.naz_impf_oracle(oracles[dtype], d, opts)
        for key, v in ref.items():
            if v is not None and np.size(v) > 1:
                assert rel_err(got[key], v) < 100 * TOL[dtype], (name, key)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_standalone_fixtures_are_the_reference(dtype):
    """Fixtures g29 (precompute_collective_explicit / _implicit) and g30 (topN_old_collective_explicit / _implicit) against the
    compiled reference run live: what the GPU tests of the same names compare the product with is what the reference returns here
    (the summation order of its BLAS may differ between hosts: tolerance of the fixture's precision, ids exact)."""
    import golden_cases as gc
    if not ref_available(dtype):
        pytest.skip("oracle/_ref for this precision not built")
    lib = Reference(dtype).lib
    tol = 1e-10 if dtype is np.float64 else 1e-4
    g = gc.load("g29_precompute_standalone", dtype)
    d = gc.precompute_problem(dtype)
    for tag, cases, call in (("e", gc.PRECOMPUTE_EXPLICIT_CASES, gc.precompute_explicit_call), ("i", gc.PRECOMPUTE_IMPLICIT_CASES, gc.precompute_implicit_call)):
        for ci, (name, opts) in enumerate(cases):
            for key, v in call(lib, d, opts, dtype).items():
                ref = g["%s%d_%s" % (tag, ci, key)]
                assert np.abs(v.astype(np.float64) - ref).max() <= tol * max(np.abs(ref).max(), 1e-30), (name, key)
    g = gc.load("g30_topn_old", dtype)
    d = gc.topn_problem(dtype)
    for ci, (name, opts) in enumerate(gc.TOPN_CASES):
        got = gc.topn_call(lib, d, opts, dtype)
        assert np.array_equal(got["ids"], g["c%d_ids" % ci]), name
        assert np.abs(got["scores"] - g["c%d_scores" % ci]).max() <= tol * np.abs(g["c%d_scores" % ci]).max(), name


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_weights_sparse_side_fixture_normal_equations(ci):
    """Fixture g31 (observation weights + sparse side information, closed form) checked by plain linear algebra: the A-step is the
    last update of an iteration (collective.c:8802), so every row of the fixture's A solves
        (sum_j w_ij b_j b_j^T  (+)  w_user sum_l c_l c_l^T  +  lam mult_i I) a_i = sum_j w_ij (x_ij - biasB_j) b_j  (+)  w_user sum_l u_il c_l
    over the fixture's B, C and item biases, with b_j = [B_j, 1] when the user bias is fitted (its unknown is the last one) and
    mult_i = the sum of the row's weights under scale_lam (collective.c:1285-1292)."""
    import golden_cases as gc
    dtype = np.float64
    g = gc.load("g31_weights_sparse_side", dtype)
    d = gc.weights_sparse_side_problem(dtype)
    name, which, opts = gc.WEIGHT_SPARSE_SIDE_CASES[ci]
    assert not opts.get("use_cg", False) and not opts.get("center", True)
    ku = opts.get("k_user", 0) if "U" in which else 0
    ki = opts.get("k_item", 0) if "I" in which else 0
    km, k = opts.get("k_main", 0), d["k"]
    A, B, Cm = g["c%d_A" % ci], g["c%d_B" % ci], g["c%d_C" % ci]
    ub = opts.get("user_bias", True)
    bB = g["c%d_biasB" % ci] if opts.get("item_bias", True) else np.zeros(d["n"])
    bA = g["c%d_biasA" % ci] if ub else None
    ur, uc, uv, m_u, p = d["U_coo"]
    kt = ku + k + km + (1 if ub else 0)
    lam, w_user = 0.3, 2.0
    worst = 0.0
    for i in range(d["m"]):
        sel = d["row"] == i; usel = ur == i
        if not sel.any() and not usel.any():
            assert not A[i].any()
            continue
        M = np.zeros((kt, kt)); rhs = np.zeros(kt)
        Bt = np.zeros((int(sel.sum()), kt)); Bt[:, ku:ku + k + km] = B[d["col"][sel], ki:]
        if ub: Bt[:, -1] = 1.0
        w = d["W"][sel]
        M += (Bt * w[:, None]).T @ Bt
        rhs += Bt.T @ (w * (d["ratings"][sel] - bB[d["col"][sel]]))
        Ct = np.zeros((int(usel.sum()), kt)); Ct[:, :ku + k] = Cm[uc[usel]]
        M += w_user * Ct.T @ Ct
        rhs += w_user * Ct.T @ uv[usel]
        mult = (w.sum() if sel.any() else 1.0) if opts.get("scale_lam", False) else 1.0
        M += lam * mult * np.eye(kt)
        sol = np.linalg.solve(M, rhs)
        got = np.concatenate([A[i], [bA[i]]]) if ub else A[i]
        worst = max(worst, np.abs(sol - got).max() / max(np.abs(sol).max(), 1e-30))
    assert worst < 1e-9, (name, worst)


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_implicit_features_sparse_side_fixture_normal_equations(ci):
    """Fixture g32 (implicit features + sparse side information, closed form) by plain linear algebra, as for g31: every row of the
    fixture's A solves the system of its X entries, its attributes and the implicit-features term  w_i Bi^T Bi  /  w_i sum_j Bi_j
    on the X block (collective.c:1704-1707, :1757-1771); lambda's multiplier under scale_lam (+ scale_lam_sideinfo) is the number of
    entries (+ of present attributes), :1285-1346."""
    import golden_cases as gc
    dtype = np.float64
    g = gc.load("g32_implicit_features_sparse_side", dtype)
    d = gc.weights_sparse_side_problem(dtype)
    name, which, opts = gc.IMPF_SPARSE_SIDE_CASES[ci]
    assert not opts.get("use_cg", False)
    ku = opts.get("k_user", 0) if "U" in which else 0
    ki = opts.get("k_item", 0) if "I" in which else 0
    km, k = opts.get("k_main", 0), d["k"]
    A, B, Cm, Bi = g["c%d_A" % ci], g["c%d_B" % ci], g["c%d_C" % ci], g["c%d_Bi" % ci]
    ub = opts.get("user_bias", True)
    bB = g["c%d_biasB" % ci] if opts.get("item_bias", True) else np.zeros(d["n"])
    bA = g["c%d_biasA" % ci] if ub else None
    gm = float(g["c%d_glob_mean" % ci]) if opts.get("center", True) else 0.0
    ur, uc, uv, m_u, p = d["U_coo"]
    kt = ku + k + km + (1 if ub else 0)
    lam, w_user, w_imp = 0.3, 2.0, opts.get("w_implicit", 1.0)
    BiTBi = Bi.T @ Bi
    worst = 0.0
    for i in range(d["m"]):
        sel = d["row"] == i; usel = ur == i
        if not sel.any() and not usel.any():
            assert not A[i].any()
            continue
        M = np.zeros((kt, kt)); rhs = np.zeros(kt)
        Bt = np.zeros((int(sel.sum()), kt)); Bt[:, ku:ku + k + km] = B[d["col"][sel], ki:]
        if ub: Bt[:, -1] = 1.0
        M += Bt.T @ Bt
        rhs += Bt.T @ (d["ratings"][sel] - gm - bB[d["col"][sel]])
        Ct = np.zeros((int(usel.sum()), kt)); Ct[:, :ku + k] = Cm[uc[usel]]
        M += w_user * Ct.T @ Ct
        rhs += w_user * Ct.T @ uv[usel]
        M[ku:ku + k + km, ku:ku + k + km] += w_imp * BiTBi
        rhs[ku:ku + k + km] += w_imp * Bi[d["col"][sel]].sum(axis=0)
        mult = 1.0
        if opts.get("scale_lam", False) or opts.get("scale_lam_sideinfo", False):
            mult = float(sel.sum()) if sel.any() else 1.0
            if opts.get("scale_lam_sideinfo", False): mult += float(usel.sum())
        M += lam * mult * np.eye(kt)
        sol = np.linalg.solve(M, rhs)
        got = np.concatenate([A[i], [bA[i]]]) if ub else A[i]
        worst = max(worst, np.abs(sol - got).max() / max(np.abs(sol).max(), 1e-30))
    assert worst < 1e-9, (name, worst)


def test_round6_fixtures_are_the_reference():
    """Fixtures g31 .. g39 (the f4 remainder of round 6: weights / implicit features with sparse side information, dense X with side
    information, NA_as_zero_X under use_cg and with weights + sparse side information) against the compiled reference run live, double
    precision: what the GPU tests compare the product with is what the reference returns on this host too."""
    import golden_cases as gc
    dtype = np.float64
    if not ref_available(dtype):
        pytest.skip("oracle/_ref for this precision not built")
    R = Reference(dtype)
    def check(fixture, cases, run):
        g = gc.load(fixture, dtype)
        for ci, case in enumerate(cases):
            for key, v in run(case).items():
                if v is None:
                    continue
                ref = g["c%d_%s" % (ci, key)]
                assert np.abs(np.asarray(v, np.float64) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-30), (fixture, case[0], key)
    d = gc.weights_sparse_side_problem(dtype)
    check("g31_weights_sparse_side", gc.WEIGHT_SPARSE_SIDE_CASES, lambda c: gc.weights_sparse_side_reference(R, d, c[1], c[2]))
    check("g32_implicit_features_sparse_side", gc.IMPF_SPARSE_SIDE_CASES, lambda c: gc.impf_sparse_side_reference(R, d, c[1], c[2]))
    check("g38_weights_implicit_features", gc.WEIGHT_IMPF_CASES, lambda c: gc.weights_impf_reference(R, d, c[1], c[2]))
    check("g39_na_as_zero_weighted_implicit_features", gc.NAZ_WEIGHTED_IMPF_CASES, lambda c: gc.naz_weighted_impf_reference(R, d, c[1], c[2]))
    check("g36_na_as_zero_weighted_sparse_side", gc.NAZ_WEIGHTED_SPARSE_SIDE_CASES, lambda c: gc.naz_weighted_sparse_side_reference(R, d, c[1], c[2]))
    check("g33_dense_X_sideinfo", gc.DENSE_SIDE_CASES, lambda c: gc.dense_side_reference(R, gc.dense_side_problem(dtype, c[1]), c[2], c[3]))
    dn = gc.naz_sparse_side_problem(dtype)
    check("g34_na_as_zero_sparse_side_cg", gc.NAZ_SPARSE_SIDE_CG_CASES, lambda c: gc.naz_sparse_side_reference(R, dn, c[1], c[2]))
    dw = gc.naz_weighted_problem(dtype)
    check("g35_na_as_zero_weighted_sideinfo_cg", gc.NAZ_WEIGHTED_SIDE_CG_CASES, lambda c: gc.naz_side_reference(R, dw, c[1], c[2], weights=True))
    check("g37_na_as_zero_implicit_features_sideinfo", gc.NAZ_IMPF_SIDE_CASES, lambda c: gc.naz_impf_side_reference(R, c[1], c[2], c[3], dtype))
