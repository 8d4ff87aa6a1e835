"""GPU: a seeded sweep of the newer paths over shapes the fixtures do not cover -- widths that are not multiples of
8 / 16 / 64, one to three 64-lane groups of unknowns, rows heavy enough for the four-wave teams of the generic CG
kernel (>= 129 entries) and for several gather chunks, side information on one or both sides -- against the oracle."""
import numpy as np
import pytest

from conftest import make_coo, rel_err

pytestmark = pytest.mark.gpu


def _coo(rng, rows, cols, cnt, empty=()):
    lin = rng.choice(rows * cols, size=min(cnt, rows * cols), replace=False)
    r = (lin // cols).astype(np.int32); c = (lin % cols).astype(np.int32)
    keep = ~np.isin(r, empty)
    return r[keep], c[keep]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,ku,ki,km", [(3, 0, 0, 0), (21, 1, 2, 0), (70, 0, 3, 1), (126, 2, 0, 1)])
def test_sparse_sideinfo_operator_sweep(oracles, dtype, k, ku, ki, km):
    """cmfrec_hip_optimizeA_collective_sparse (two gather sources, Cholesky) with heavy rows and heavy attribute lists."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    tol = 1e-10 if dtype is np.float64 else 5e-4
    rng = np.random.default_rng(1000 + k)
    m, n, p, m_u = 260, 700, 90, 250
    B = (rng.standard_normal((n, ki + k + km)) * 0.3).astype(dtype); Cm = (rng.standard_normal((p, ku + k)) * 0.3).astype(dtype)
    row, col, val = make_coo(m, n, 9000, k, counts=False, dtype=dtype, heavy_row=(7, 600), empty_rows=(3, 255))
    ur, uc = _coo(rng, m_u, p, 4000, empty=(5, 255))
    ur = np.concatenate([ur[ur != 9], np.full(p, 9, np.int32)]); uc = np.concatenate([uc[:len(ur) - p], np.arange(p, dtype=np.int32)])
    uv = rng.standard_normal(len(ur)).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    ucsr, _ = O.coo_to_csr_and_csc(ur, uc, uv, m_u, p)
    for implicit in (False, True):
        for sl, sls in (((False, False),) if implicit else ((True, True), (False, False))):
            a1 = rng.standard_normal((m, ku + k + km)).astype(dtype); a2 = a1.copy()
            kw = dict(w_user=1.7, lam_last=None if implicit else 0.9, k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=sl,
                      scale_lam_sideinfo=sls, implicit=implicit)
            vals = np.abs(csr[2]) if implicit else csr[2]
            ops.optimizeA_collective_sparse(a1, B, Cm, (csr[0], csr[1], vals), ucsr, 0.6, **kw)
            O.optimizeA_collective_sparse(a2, B, Cm, (csr[0], csr[1], vals), ucsr, 0.6, nthreads=4, **kw)
            assert rel_err(a1, a2) < tol, (implicit, sl, sls)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,ku,ki,km,solver", [(5, 0, 0, 0, dict(use_cg=True)), (40, 2, 1, 1, dict(use_cg=True, precondition_cg=True)),
                                               (70, 0, 2, 0, dict(use_cg=True, finalize_chol=True)), (33, 1, 0, 2, dict(use_cg=False)),
                                               (100, 0, 0, 1, dict(use_cg=False, nonneg=True)), (9, 2, 2, 0, dict(nonneg=True, nonneg_C=True))])
def test_fit_sweep_sparse_sideinfo_and_nonneg(oracles, dtype, k, ku, ki, km, solver):
    """Whole explicit fits through the estimators: sparse U and I, every solver family, rows of several hundred entries."""
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    import golden_cases as gc
    O = oracles[dtype]
    rng = np.random.default_rng(2000 + k)
    m, n, p, q = 300, 240, 30, 25
    row, col, val = make_coo(m, n, 11000, 5 + k, counts=False, dtype=dtype, heavy_row=(11, 200), empty_rows=(3, 290))
    ur, uc = _coo(rng, m - 10, p, 1500, empty=(5,)); uv = rng.standard_normal(len(ur)).astype(dtype)
    ir, ic = _coo(rng, n, q, 1300, empty=(4,)); iv = rng.standard_normal(len(ir)).astype(dtype)
    U_coo, I_coo = (ur, uc, uv, m - 10, p), (ir, ic, iv, n, q)
    A0 = np.abs(rng.standard_normal((m, ku + k + km)) * 0.1).astype(dtype); B0 = np.abs(rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
    bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
    sv = dict(solver)
    nn = dict(nonneg=sv.pop("nonneg", False), nonneg_C=sv.pop("nonneg_C", False), nonneg_D=sv.pop("nonneg_D", False))
    sv.setdefault("finalize_chol", False); sv.setdefault("use_cg", False)
    mdl = CMF(k=k, k_user=ku, k_item=ki, k_main=km, lambda_=0.4, scale_lam=True, w_user=2.0, w_item=0.6, niter=3,
              use_float=dtype is np.float32, precompute_for_predictions=False, **sv, **nn)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    mdl.fit((row, col, val), U=mk(U_coo), I=mk(I_coo), shape=(m, n), A0=A0, B0=B0, biasA0=bA, biasB0=bB)
    O.set_nonneg(nn["nonneg"], nn["nonneg_C"], nn["nonneg_D"], 100)
    try:
        a2, b2 = A0.copy(), B0.copy()
        r = O.fit_als_sparse_sideinfo(a2, b2, row, col, val, k, False, U_coo=U_coo, I_coo=I_coo, biasA=bA.copy(), biasB=bB.copy(),
                                      user_bias=True, item_bias=True, center=True, lam=0.4, scale_lam=True, k_main=km, k_user=ku,
                                      k_item=ki, w_user=2.0, w_item=0.6, niter=3, nthreads=4, **sv)
    finally:
        O.set_nonneg(False, False, False, 100)
    tol = (1e-6 if dtype is np.float64 else 1e-2)
    got = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, biasA=mdl.user_bias_, biasB=mdl.item_bias_)
    assert gc.compare_fits(got, r) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k", [1, 17, 64, 65, 130])
def test_nonneg_width_sweep(oracles, dtype, k):
    """The coordinate-descent phase with one, two and three 64-lane groups of unknowns (implicit model, plain)."""
    from cmfrec_amd import CMF_implicit
    O = oracles[dtype]
    rng = np.random.default_rng(3000 + k)
    m, n = 150, 120
    row, col, val = make_coo(m, n, 3000, 7 + k, counts=True, dtype=dtype, heavy_row=(2, 100), empty_rows=(5,))
    A0 = np.abs(rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = np.abs(rng.standard_normal((n, k)) * 0.1).astype(dtype)
    mdl = CMF_implicit(k=k, lambda_=1.5, alpha=0.8, niter=2, nonneg=True, max_cd_steps=40, use_float=dtype is np.float32,
                       precompute_for_predictions=False).fit((row, col, val), shape=(m, n), A0=A0, B0=B0)
    O.set_nonneg(True, False, False, 40)
    try:
        a2, b2 = A0.copy(), B0.copy()
        O.fit_implicit_als(a2, b2, row, col, val, lam=1.5, alpha=0.8, niter=2, nthreads=4, use_cg=False)
    finally:
        O.set_nonneg(False, False, False, 100)
    tol = 1e-6 if dtype is np.float64 else 1e-2
    assert rel_err(mdl.A_, a2) < tol and rel_err(mdl.B_, b2) < tol and (mdl.A_ >= 0).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,use_cg", [(40, True), (70, True), (70, False), (130, False)])
def test_implicit_features_long_rows(oracles, dtype, k, use_cg):
    """Implicit features on a matrix with long item rows (one beyond 1024 entries, most beyond 128): the sixteen- and
    four-wavefront teams of the generic CG kernel with the Bi gather term, the wide tile counts of the shared-matrix
    (CHOL_NAZ, right-hand sides only + library triangular solves) and two-source Cholesky launches -- against the oracle."""
    from cmfrec_amd import CMF
    rng = np.random.default_rng(91)
    m, n = 1500, 30
    lin = rng.choice(m * n, size=19000, replace=False)
    row = (lin // n).astype(np.int32); col = (lin % n).astype(np.int32)
    extra = np.setdiff1d(np.arange(m, dtype=np.int32), row[col == 0])[:900]          # item 0: > 1024 entries
    row = np.concatenate([row, extra]); col = np.concatenate([col, np.zeros(len(extra), np.int32)])
    assert np.bincount(col, minlength=n)[0] > 1024
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    bA = np.zeros(m, dtype); bB = np.zeros(n, dtype)
    kw = dict(use_cg=use_cg, finalize_chol=False)
    mdl = CMF(k=k, lambda_=2.0, niter=2, use_float=dtype is np.float32, add_implicit_features=True, w_implicit=0.7,
              precompute_for_predictions=False, **kw)
    mdl.fit((row, col, val), shape=(m, n), A0=A0, B0=B0, biasA0=bA, biasB0=bB)
    o = oracles[dtype].fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=2.0, niter=2,
                                        nthreads=4, add_implicit_features=True, w_implicit=0.7, **kw)
    assert o["ret"] == 0
    tol = 1e-7 if dtype is np.float64 else 1e-2
    for got, exp in ((mdl.A_, o["A"]), (mdl.B_, o["B"]), (mdl.Ai_, o["Ai"]), (mdl.Bi_, o["Bi"]), (mdl.user_bias_, o["biasA"]),
                     (mdl.item_bias_, o["biasB"])):
        assert np.isfinite(got).all() and np.abs(got - exp).max() <= tol * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit,pcg", [(True, False), (True, True), (False, False), (False, True)])
def test_block_cg_long_rows(oracles, dtype, implicit, pcg):
    """The block CG / PCG with dense side information on long rows (one item beyond 1024 entries: the sixteen-wavefront team
    of the generic kernel, most items beyond 128: the four-wavefront one) against the oracle, both models."""
    from cmfrec_amd import CMF, CMF_implicit
    rng = np.random.default_rng(93)
    m, n, k = 1500, 30, 24
    lin = rng.choice(m * n, size=19000, replace=False)
    row = (lin // n).astype(np.int32); col = (lin % n).astype(np.int32)
    extra = np.setdiff1d(np.arange(m, dtype=np.int32), row[col == 0])[:900]
    row = np.concatenate([row, extra]); col = np.concatenate([col, np.zeros(len(extra), np.int32)])
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    U = rng.standard_normal((m, 6)).astype(dtype); II = rng.standard_normal((n, 4)).astype(dtype)
    A0 = (rng.standard_normal((m, k + 1)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    kw = dict(use_cg=True, precondition_cg=pcg, finalize_chol=False, k_user=1, w_user=2.0, w_item=0.5)
    O = oracles[dtype]
    tol = 1e-7 if dtype is np.float64 else 1e-2
    if implicit:
        mdl = CMF_implicit(k=k, lambda_=3.0, niter=2, use_float=dtype is np.float32, precompute_for_predictions=False, **kw)
        mdl.fit((row, col, val), U=U, I=II, shape=(m, n), A0=A0, B0=B0)
        A1, B1 = A0.copy(), B0.copy()
        o = O.fit_implicit_als_sideinfo(A1, B1, row, col, val, k, lam=3.0, niter=2, U=U, II=II, nthreads=4, **kw)
        pairs = ((mdl.A_, A1), (mdl.B_, B1), (mdl.C_, o["C"]), (mdl.D_, o["D"]))
    else:
        bA = np.zeros(m, dtype); bB = np.zeros(n, dtype)
        mdl = CMF(k=k, lambda_=3.0, niter=2, use_float=dtype is np.float32, precompute_for_predictions=False, **kw)
        mdl.fit((row, col, val), U=U, I=II, shape=(m, n), A0=A0, B0=B0, biasA0=bA, biasB0=bB)
        o = O.fit_explicit_als(A0.copy(), B0.copy(), row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=3.0, niter=2, U=U, II=II,
                               nthreads=4, **kw)
        pairs = ((mdl.A_, o["A"]), (mdl.B_, o["B"]), (mdl.C_, o["C"]), (mdl.D_, o["D"]), (mdl.user_bias_, o["biasA"]))
    for got, exp in pairs:
        assert np.isfinite(got).all() and np.abs(got - exp).max() <= tol * max(1.0, np.abs(exp).max())
