"""GPU parity: whole fits through the drop-in C entry points (fit_collective_*_als, called via
the CMF / CMF_implicit estimators) against the oracle with injected start values.
Tolerances: relative Frobenius error <= 1e-6 fp64 / 1e-2 fp32 (CG), 1e-3 fp32 (Chol)."""
import numpy as np
import pytest

from conftest import make_coo

pytestmark = pytest.mark.gpu


def frob(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def tol(dtype, mode):
    if dtype is np.float64:
        return 1e-6
    return 1e-2 if "cg" in mode else 1e-3     # "pcg" contains "cg"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", ["cg", "chol", "cg+fin", "pcg"])
def test_fit_implicit(oracles, dtype, mode):
    from cmfrec_amd import CMF_implicit
    O = oracles[dtype]
    m, n, k = 900, 600, 50
    row, col, val = make_coo(m, n, 30000, 21, dtype=dtype, heavy_row=(0, 500))
    rng = np.random.default_rng(1)
    A0 = (rng.standard_normal((m, k)) * 0.01).astype(dtype)
    B0 = np.zeros((n, k), dtype)
    kw = dict(niter=4, use_cg=mode != "chol", finalize_chol=mode == "cg+fin", precondition_cg=mode == "pcg")
    mdl = CMF_implicit(k=k, lambda_=5., alpha=1.5, use_float=dtype is np.float32, **kw).fit(
        (row, col, val), shape=(m, n), A0=A0, B0=B0)
    Ao, Bo = A0.copy(), B0.copy()
    O.fit_implicit_als(Ao, Bo, row, col, val, lam=5., alpha=1.5, nthreads=4, **kw)
    assert frob(mdl.A_, Ao) < tol(dtype, mode) and frob(mdl.B_, Bo) < tol(dtype, mode)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode,ub,ib", [("cg", True, True), ("chol", True, True), ("cg+fin", True, True), ("pcg", True, True),
                                        ("cg", False, False), ("cg", True, False), ("chol", False, True)])
def test_fit_explicit(oracles, dtype, mode, ub, ib):
    from cmfrec_amd import CMF
    O = oracles[dtype]
    m, n, k = 800, 500, 50
    row, col, val = make_coo(m, n, 30000, 22, counts=False, dtype=dtype, heavy_row=(1, 450))
    rng = np.random.default_rng(2)
    A0 = (rng.standard_normal((m, k)) * 0.01).astype(dtype)
    B0 = np.zeros((n, k), dtype)
    bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
    kw = dict(niter=3, use_cg=mode != "chol", finalize_chol=mode == "cg+fin", user_bias=ub, item_bias=ib,
              scale_lam=True, precondition_cg=mode == "pcg")
    mdl = CMF(k=k, lambda_=0.05, use_float=dtype is np.float32, nthreads=1, **kw).fit(
        (row, col, val), shape=(m, n), A0=A0, B0=B0, biasA0=bA, biasB0=bB)
    Ao, Bo = A0.copy(), B0.copy()
    ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=0.05, nthreads=1, **kw)
    t = tol(dtype, mode)
    assert ro["ret"] == 0
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    assert abs(mdl.glob_mean_ - ro["glob_mean"]) <= 1e-6 * abs(ro["glob_mean"])
    if ub:
        assert frob(mdl.user_bias_, ro["biasA"]) < t
    if ib:
        assert frob(mdl.item_bias_, ro["biasB"]) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("useU,useI,ku,ki,km,sls", [(False, True, 0, 0, 0, False), (True, True, 0, 0, 0, False),
                                                     (True, True, 2, 3, 1, True), (True, False, 1, 0, 0, False)])
def test_fit_explicit_sideinfo(oracles, dtype, useU, useI, ku, ki, km, sls):
    from cmfrec_amd import CMF
    O = oracles[dtype]
    m, n, k, p, q = 500, 320, 24, 12, 9
    row, col, val = make_coo(m, n, 12000, 23, counts=False, dtype=dtype)
    rng = np.random.default_rng(3)
    U = (rng.standard_normal((m, p)) + 1).astype(dtype); II = (rng.standard_normal((n, q)) - 2).astype(dtype)
    kA, kB = ku + k + km, ki + k + km
    A0 = (rng.standard_normal((m, kA)) * 0.01).astype(dtype); B0 = (rng.standard_normal((n, kB)) * 0.01).astype(dtype)
    kw = dict(niter=3, use_cg=False, k_user=ku, k_item=ki, k_main=km, w_user=0.5, w_item=2.0, scale_lam=True,
              scale_lam_sideinfo=sls)
    mdl = CMF(k=k, lambda_=0.05, use_float=dtype is np.float32, nthreads=1, **kw).fit(
        (row, col, val), shape=(m, n), U=U if useU else None, I=II if useI else None, A0=A0, B0=B0)
    Ao, Bo = A0.copy(), B0.copy()
    ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, lam=0.05, U=U if useU else None, II=II if useI else None,
                            nthreads=1, **kw)
    t = tol(dtype, "chol")
    assert ro["ret"] == 0
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    if useU:
        assert frob(mdl.C_, ro["C"]) < t
    if useI:
        assert frob(mdl.D_, ro["D"]) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("useU,useI,ku,ki,km,m_u", [(False, True, 0, 0, 0, None), (True, True, 0, 0, 0, 500),
                                                     (True, True, 2, 3, 1, 430), (True, False, 1, 0, 0, 500)])
def test_fit_implicit_sideinfo(oracles, dtype, useU, useI, ku, ki, km, m_u):
    """CMF_implicit with dense side information (Cholesky): optimizeA_collective_implicit +
    collective_closed_form_block_implicit (collective.c:5971-6244, 1849-2131) on the device."""
    from cmfrec_amd import CMF_implicit
    O = oracles[dtype]
    m, n, k, p, q = 500, 320, 24, 12, 9
    row, col, val = make_coo(m, n, 12000, 29, counts=True, dtype=dtype, empty_rows=(3, 480))
    rng = np.random.default_rng(5)
    U = (rng.standard_normal((m_u or m, p)) + 1).astype(dtype); II = (rng.standard_normal((n, q)) - 2).astype(dtype)
    kA, kB = ku + k + km, ki + k + km
    A0 = (rng.standard_normal((m, kA)) * 0.01).astype(dtype); B0 = (rng.standard_normal((n, kB)) * 0.01).astype(dtype)
    kw = dict(niter=3, use_cg=False, k_user=ku, k_item=ki, k_main=km, w_main=0.5, w_user=4.0, w_item=0.8, alpha=2.0)
    mdl = CMF_implicit(k=k, lambda_=3.0, use_float=dtype is np.float32, **kw).fit(
        (row, col, val), shape=(m, n), U=U if useU else None, I=II if useI else None, A0=A0, B0=B0)
    Ao, Bo = A0.copy(), B0.copy()
    ro = O.fit_implicit_als_sideinfo(Ao, Bo, row, col, val, k, lam=3.0, U=U if useU else None, II=II if useI else None,
                                     nthreads=1, **kw)
    t = tol(dtype, "chol")
    assert ro["ret"] == 0
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    if useU:
        assert frob(mdl.C_, ro["C"]) < t
    if useI:
        assert frob(mdl.D_, ro["D"]) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit", [False, True])
@pytest.mark.parametrize("pcg,ku,ki,km,m_u,fin", [(False, 0, 0, 0, 500, False), (False, 2, 3, 1, 430, True),
                                                   (True, 1, 0, 2, 500, False)])
def test_fit_sideinfo_block_cg(oracles, dtype, implicit, pcg, ku, ki, km, m_u, fin):
    """Side information with use_cg=True: the block CG / PCG of the collective system
    (collective_block_cg, collective.c:2134-2903; collective_block_cg_implicit, :2905-3303) on the device,
    optionally finishing with a Cholesky iteration (finalize_chol)."""
    from cmfrec_amd import CMF, CMF_implicit
    O = oracles[dtype]
    m, n, k, p, q = 500, 320, 24, 12, 9
    row, col, val = make_coo(m, n, 12000, 31, counts=implicit, dtype=dtype, empty_rows=(3, 480))
    rng = np.random.default_rng(6)
    U = (rng.standard_normal((m_u, p)) + 1).astype(dtype); II = (rng.standard_normal((n, q)) - 2).astype(dtype)
    kA, kB = ku + k + km, ki + k + km
    A0 = (rng.standard_normal((m, kA)) * 0.01).astype(dtype); B0 = (rng.standard_normal((n, kB)) * 0.01).astype(dtype)
    Ao, Bo = A0.copy(), B0.copy()
    if implicit:
        kw = dict(niter=3, use_cg=True, precondition_cg=pcg, finalize_chol=fin, k_user=ku, k_item=ki, k_main=km, w_main=0.5,
                  w_user=4.0, w_item=0.8, alpha=2.0)
        mdl = CMF_implicit(k=k, lambda_=3.0, use_float=dtype is np.float32, **kw).fit(
            (row, col, val), shape=(m, n), U=U, I=II, A0=A0, B0=B0)
        ro = O.fit_implicit_als_sideinfo(Ao, Bo, row, col, val, k, lam=3.0, U=U, II=II, nthreads=1, **kw)
    else:
        kw = dict(niter=3, use_cg=True, precondition_cg=pcg, finalize_chol=fin, k_user=ku, k_item=ki, k_main=km,
                  w_user=0.5, w_item=2.0, scale_lam=True, scale_lam_sideinfo=ku > 0)
        mdl = CMF(k=k, lambda_=0.05, use_float=dtype is np.float32, nthreads=1, **kw).fit(
            (row, col, val), shape=(m, n), U=U, I=II, A0=A0, B0=B0)
        ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, lam=0.05, U=U, II=II, nthreads=1, **kw)
    t = tol(dtype, "cg")
    assert ro["ret"] == 0
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    assert frob(mdl.C_, ro["C"]) < t and frob(mdl.D_, ro["D"]) < t
    if not implicit:
        assert frob(mdl.user_bias_, ro["biasA"]) < t and frob(mdl.item_bias_, ro["biasB"]) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,km", [(96, 0), (127, 1)])
def test_fit_sideinfo_block_cg_beyond_64_unknowns(oracles, dtype, k, km):
    """The block CG of the collective system with more than 64 unknowns per row (config 3's shape under use_cg): rank-k update by
    the Cholesky producer, CG on the row's Gramian + w C^T C (gram_cg_wide_kernels.hpp), dense side information on both sides, biases,
    both lambda scalings; rows without entries are solved from their side information alone."""
    from cmfrec_amd import CMF
    O = oracles[dtype]
    m, n, p, q = 260, 200, 12, 9
    row, col, val = make_coo(m, n, 9000, 31, counts=False, dtype=dtype, empty_rows=(3, 250))
    rng = np.random.default_rng(6)
    U = (rng.standard_normal((m, p)) + 1).astype(dtype); II = (rng.standard_normal((n, q)) - 2).astype(dtype)
    A0 = (rng.standard_normal((m, k + km)) * 0.01).astype(dtype); B0 = (rng.standard_normal((n, k + km)) * 0.01).astype(dtype)
    Ao, Bo = A0.copy(), B0.copy()
    kw = dict(niter=3, use_cg=True, finalize_chol=False, k_main=km, w_user=0.5, w_item=2.0, scale_lam=True, scale_lam_sideinfo=km > 0)
    mdl = CMF(k=k, lambda_=0.05, use_float=dtype is np.float32, nthreads=1, **kw).fit((row, col, val), shape=(m, n), U=U, I=II, A0=A0, B0=B0)
    ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, lam=0.05, U=U, II=II, nthreads=1, **kw)
    t = tol(dtype, "cg")
    assert ro["ret"] == 0
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    assert frob(mdl.C_, ro["C"]) < t and frob(mdl.D_, ro["D"]) < t
    assert frob(mdl.user_bias_, ro["biasA"]) < t and frob(mdl.item_bias_, ro["biasB"]) < t
    assert np.abs(mdl.A_[3]).max() > 0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit", [False, True])
@pytest.mark.parametrize("cg,ku,ki,km", [(False, 0, 0, 0), (False, 2, 3, 1), (True, 1, 0, 2)])
def test_fit_sideinfo_beyond_X(oracles, dtype, implicit, cg, ku, ki, km):
    """m_u > m and n_i > n: users / items known only from their side information (collective.c:4832-5101,
    6037-6090): factor matrices with max(m, m_u) / max(n, n_i) rows, zero biases beyond X."""
    from cmfrec_amd import CMF, CMF_implicit
    O = oracles[dtype]
    m, n, k, p, q, m_u, n_i = 500, 320, 24, 12, 9, 540, 350
    row, col, val = make_coo(m, n, 12000, 33, counts=implicit, dtype=dtype, empty_rows=(3, 480))
    rng = np.random.default_rng(8)
    U = (rng.standard_normal((m_u, p)) + 1).astype(dtype); II = (rng.standard_normal((n_i, q)) - 2).astype(dtype)
    A0 = (rng.standard_normal((m_u, ku + k + km)) * 0.01).astype(dtype)
    B0 = (rng.standard_normal((n_i, ki + k + km)) * 0.01).astype(dtype)
    Ao, Bo = A0.copy(), B0.copy()
    if implicit:
        kw = dict(niter=3, use_cg=cg, k_user=ku, k_item=ki, k_main=km, w_main=0.5, w_user=4.0, w_item=0.8, alpha=2.0)
        mdl = CMF_implicit(k=k, lambda_=3.0, use_float=dtype is np.float32, **kw).fit(
            (row, col, val), shape=(m, n), U=U, I=II, A0=A0, B0=B0)
        ro = O.fit_implicit_als_sideinfo(Ao, Bo, row, col, val, k, lam=3.0, U=U, II=II, nthreads=1, m=m, n=n, **kw)
    else:
        kw = dict(niter=3, use_cg=cg, finalize_chol=False, k_user=ku, k_item=ki, k_main=km, w_user=0.5, w_item=2.0,
                  scale_lam=True, scale_lam_sideinfo=ku > 0)
        mdl = CMF(k=k, lambda_=0.05, use_float=dtype is np.float32, nthreads=1, **kw).fit(
            (row, col, val), shape=(m, n), U=U, I=II, A0=A0, B0=B0)
        ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, lam=0.05, U=U, II=II, nthreads=1, m=m, n=n, **kw)
    t = tol(dtype, "cg" if cg else "chol")
    assert ro["ret"] == 0
    assert mdl.A_.shape[0] == m_u and mdl.B_.shape[0] == n_i
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    assert frob(mdl.A_[m:], Ao[m:]) < t and frob(mdl.B_[n:], Bo[n:]) < t              # the rows beyond X on their own
    assert frob(mdl.C_, ro["C"]) < t and frob(mdl.D_, ro["D"]) < t
    if not implicit:
        assert frob(mdl.user_bias_, ro["biasA"]) < t and frob(mdl.item_bias_, ro["biasB"]) < t
        assert not mdl.user_bias_[m:].any() and not mdl.item_bias_[n:].any()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_seeded_fit_matches_reference_seed(dtype):
    """reset_values=true: the start values come from the seed exactly as in the reference
    (xoshiro256++ / ziggurat, helpers.c:927-1043), so a seeded fit() reproduces the reference's own
    seeded fit -- checked against the real reference where oracle/_ref is available."""
    from oracle.bindings import Reference, ref_available
    if not ref_available(dtype):
        pytest.skip("oracle/_ref not built")
    from cmfrec_amd import CMF, CMF_implicit
    R = Reference(dtype)
    uf = dtype is np.float32
    m, n, k = 700, 500, 24
    row, col, val = make_coo(m, n, 20000, 31, dtype=dtype)
    mdl = CMF_implicit(k=k, lambda_=3., niter=3, random_state=123, use_float=uf).fit((row, col, val), shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    R.fit_collective_implicit_als(Ar, Br, row, col, val, k, lam=3., niter=3, nthreads=2, reset_values=True, seed=123)
    t = 1e-6 if dtype is np.float64 else 1e-2
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t
    row, col, val = make_coo(m, n, 20000, 32, counts=False, dtype=dtype)
    mdl = CMF(k=k, lambda_=0.05, scale_lam=True, niter=3, random_state=7, use_float=uf, nthreads=1).fit(
        (row, col, val), shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, scale_lam=True, niter=3, nthreads=2,
                                       reset_values=True, seed=7)
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t
    assert frob(mdl.user_bias_, rr["biasA"]) < t and frob(mdl.item_bias_, rr["biasB"]) < t
    # a single bias: one-sided bias initialisation (common.c:4130-4289)
    for ub, ib in ((True, False), (False, True)):
        mdl = CMF(k=k, lambda_=0.05, scale_lam=True, niter=3, random_state=9, use_float=uf, nthreads=1, user_bias=ub,
                  item_bias=ib).fit((row, col, val), shape=(m, n))
        Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
        rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, scale_lam=True, niter=3, nthreads=2,
                                           reset_values=True, seed=9, user_bias=ub, item_bias=ib)
        assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t, (ub, ib)
        if ub:
            assert frob(mdl.user_bias_, rr["biasA"]) < t
        if ib:
            assert frob(mdl.item_bias_, rr["biasB"]) < t
    # non-negative factors: the seeded start values are made non-negative (collective.c:8256-8263) and the CG is off
    mdl = CMF(k=k, lambda_=0.05, niter=2, random_state=11, use_float=uf, nthreads=1, nonneg=True, user_bias=False,
              item_bias=False, center=False).fit((row, col, val), shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, niter=2, nthreads=2, reset_values=True, seed=11, nonneg=True,
                                  user_bias=False, item_bias=False, center=False)
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t and (mdl.A_ >= 0).all()
    # sparse item side information with the CG solvers: B gets seeded values, D starts at zero (:8243-8273)
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    ir = rng.integers(0, n, 3000).astype(np.int32); ic = rng.integers(0, 12, 3000).astype(np.int32)
    key = np.unique(ir.astype(np.int64) * 12 + ic); ir = (key // 12).astype(np.int32); ic = (key % 12).astype(np.int32)
    iv = rng.standard_normal(len(ir)).astype(dtype)
    mdl = CMF(k=k, lambda_=0.05, scale_lam=True, niter=3, random_state=13, use_float=uf, nthreads=1, w_item=2.0).fit(
        (row, col, val), I=sp.coo_matrix((iv, (ir, ic)), shape=(n, 12)), shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, scale_lam=True, niter=3, nthreads=2, reset_values=True,
                                       seed=13, w_item=2.0, I_coo=(ir, ic, iv, n, 12))
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t and frob(mdl.D_, rr["D"]) < t
    # dense side information under scale_lam_sideinfo: the bias start values count the attributes of the rows that have
    # them on top of their entries (wsumA / wsumB, collective.c:8071-8104); U covers fewer users than X
    U = rng.standard_normal((m - 60, 5)).astype(dtype); II = rng.standard_normal((n, 4)).astype(dtype)
    mdl = CMF(k=k, lambda_=0.05, scale_lam_sideinfo=True, niter=2, random_state=17, use_float=uf, nthreads=1, use_cg=False,
              w_user=2.0).fit((row, col, val), U=U, I=II, shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, scale_lam_sideinfo=True, niter=2, nthreads=2,
                                       reset_values=True, seed=17, use_cg=False, finalize_chol=False, w_user=2.0, U=U, II=II)
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t and frob(mdl.C_, rr["C"]) < t
    assert frob(mdl.user_bias_, rr["biasA"]) < t and frob(mdl.item_bias_, rr["biasB"]) < t
    # the same data with the CG solver (C, D start at zero, B is seeded because there is item side information, :8243-8273)
    mdl = CMF(k=k, lambda_=0.05, scale_lam=True, niter=3, random_state=19, use_float=uf, nthreads=1, use_cg=True, finalize_chol=True,
              w_user=2.0, k_user=2).fit((row, col, val), U=U, I=II, shape=(m, n))
    Ar, Br = np.zeros((m, k + 2), dtype), np.zeros((n, k), dtype)
    rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, scale_lam=True, niter=3, nthreads=2, reset_values=True,
                                       seed=19, use_cg=True, finalize_chol=True, w_user=2.0, k_user=2, U=U, II=II)
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t and frob(mdl.C_, rr["C"]) < t and frob(mdl.D_, rr["D"]) < t
    # implicit model with dense side information, seeded (uniform start values, B seeded too, collective.c:9756-9785)
    cnt = np.ceil(np.abs(val) * 3 + 1).astype(dtype)
    mdl = CMF_implicit(k=k, lambda_=2., niter=3, random_state=23, use_float=uf, w_user=3.0, w_item=0.5, k_item=1).fit(
        (row, col, cnt), U=U, I=II, shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k + 1), dtype)
    rr = R.fit_collective_implicit_als(Ar, Br, row, col, cnt, k, lam=2., niter=3, nthreads=2, reset_values=True, seed=23, w_user=3.0,
                                       w_item=0.5, k_item=1, U=U, II=II)
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t and frob(mdl.C_, rr["C"]) < t and frob(mdl.D_, rr["D"]) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_scale_bias_const_matches_reference(dtype):
    """scale_bias_const through the estimator, seeded like the reference (the scaling constants are outputs of the fit):
    Cholesky and CG, with and without dense side information (fewer users in U than in X), single biases."""
    from oracle.bindings import Reference, ref_available
    if not ref_available(dtype):
        pytest.skip("oracle/_ref not built")
    from cmfrec_amd import CMF
    from test_oracle_vs_ref import SCALE_BIAS_CONST_CASES
    R = Reference(dtype)
    uf = dtype is np.float32
    m, n, k = 700, 500, 24
    row, col, val = make_coo(m, n, 20000, 35, counts=False, dtype=dtype)
    rng = np.random.default_rng(8)
    Ud = rng.standard_normal((m - 60, 5)).astype(dtype); Id = rng.standard_normal((n, 4)).astype(dtype)
    t = 1e-6 if dtype is np.float64 else 1e-2
    for name, side, o in SCALE_BIAS_CONST_CASES:
        o = dict(o); U, II = (Ud, Id) if side else (None, None)
        kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
        mdl = CMF(k=k, lambda_=0.05, niter=2, random_state=29, use_float=uf, nthreads=1, scale_bias_const=True,
                  precompute_for_predictions=False, **kw).fit((row, col, val), U=U, I=II, shape=(m, n))
        Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
        rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, niter=2, nthreads=2, reset_values=True, seed=29,
                                           scale_bias_const=True, U=U, II=II, **kw)
        assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t, name
        if mdl.user_bias:
            assert frob(mdl.user_bias_, rr["biasA"]) < t and abs(mdl._scaling_biasA - rr["scaling_biasA"]) < 1e-5 * rr["scaling_biasA"], name
        if mdl.item_bias:
            assert frob(mdl.item_bias_, rr["biasB"]) < t and abs(mdl._scaling_biasB - rr["scaling_biasB"]) < 1e-5 * rr["scaling_biasB"], name
        if name == "sl chol":
            # the scaling constant travels to the new-row function: a fit that ends on a Cholesky A-step leaves the closed-form
            # factors of its own rows, so the training data handed back reproduces A_ and the user biases
            import scipy.sparse as sp
            A2, b2 = mdl.factors_multiple(sp.csr_matrix((val, (row, col)), shape=(m, n)), return_bias=True)
            ne = np.bincount(row, minlength=m) > 0
            t2 = 1e-8 if dtype is np.float64 else 5e-3
            assert np.abs(A2[ne] - mdl.A_[ne]).max() < t2 * max(1.0, np.abs(mdl.A_).max()) and np.abs(b2[ne] - mdl.user_bias_[ne]).max() < t2 * 10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cu,ci", [(False, False), (False, True), (True, False)])
def test_fit_explicit_uncentred_sideinfo(dtype, cu, ci):
    """center_U / center_I = False: the side information is used as given (no column means are handed to the C function, as the
    reference's estimator does); seeded fit against the compiled reference, Cholesky and CG, on matrices whose columns are far
    from zero mean; the new-row function afterwards works without the means."""
    from oracle.bindings import Reference, ref_available
    if not ref_available(dtype):
        pytest.skip("oracle/_ref not built")
    from cmfrec_amd import CMF
    R = Reference(dtype)
    m, n, k, p, q = 420, 300, 16, 6, 5
    row, col, val = make_coo(m, n, 9000, 41, counts=False, dtype=dtype)
    rng = np.random.default_rng(5)
    U = (rng.standard_normal((m, p)) + 1.5).astype(dtype); II = (rng.standard_normal((n, q)) - 2.0).astype(dtype)
    t = 1e-6 if dtype is np.float64 else 1e-2
    for use_cg in (False, True):
        kw = dict(niter=2, use_cg=use_cg, finalize_chol=False, w_user=0.5, w_item=2.0)
        mdl = CMF(k=k, lambda_=0.05, random_state=17, use_float=dtype is np.float32, nthreads=1, center_U=cu, center_I=ci,
                  precompute_for_predictions=False, **kw).fit((row, col, val), U=U, I=II, shape=(m, n))
        Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
        rr = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, lam=0.05, nthreads=2, reset_values=True, seed=17, U=U, II=II,
                                           center_U=cu, center_I=ci, **kw)
        assert rr["ret"] == 0
        assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t, use_cg
        assert frob(mdl.C_, rr["C"]) < t and frob(mdl.D_, rr["D"]) < t, use_cg
        assert len(mdl._U_colmeans) == (p if cu else 0) and len(mdl._I_colmeans) == (q if ci else 0)
        # and it matters: the centred fit of the same data is a different model
        ctr = CMF(k=k, lambda_=0.05, random_state=17, use_float=dtype is np.float32, nthreads=1, precompute_for_predictions=False,
                  **kw).fit((row, col, val), U=U, I=II, shape=(m, n))
        assert frob(mdl.C_ if not cu else mdl.D_, ctr.C_ if not cu else ctr.D_) > 1e-2


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("side", [False, True])
def test_factors_multiple_after_fit(oracles, dtype, side):
    """``factors_multiple`` of the estimators (factors_collective_*_multiple underneath).  A fit that ends on a Cholesky
    A-step leaves exactly the closed-form factors of its own training rows, so handing the training data back must
    reproduce A_ (and the user biases); then genuinely new rows are checked against the oracle."""
    import scipy.sparse as sp
    import golden_cases as gc
    from cmfrec_amd import CMF, CMF_implicit
    O = oracles[dtype]
    t = 1e-9 if dtype is np.float64 else 2e-3
    m, n, k, p = 900, 500, 24, 6
    rng = np.random.default_rng(5)
    U = rng.standard_normal((m, p)).astype(dtype) if side else None
    row, col, val = make_coo(m, n, 20000, 8, counts=False, dtype=dtype, heavy_row=(5, 300), empty_rows=() if side else (9,))
    X = sp.coo_matrix((val, (row, col)), shape=(m, n))
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    mdl = CMF(k=k, lambda_=0.8, niter=3, use_cg=False, use_float=dtype is np.float32, scale_lam=True, w_user=2.0,
              precompute_for_predictions=side).fit(X, U=U, A0=A0, B0=B0)
    A, bias = mdl.factors_multiple(X, U=U, return_bias=True)
    ne = np.ones(m, bool) if side else (np.arange(m) != 9)      # the fit leaves a row without data at its start value
    assert gc.maxrel(A[ne], mdl.A_[ne]) < t and gc.maxrel(bias[ne], mdl.user_bias_[ne]) < t
    assert side or (not A[9].any() and bias[9] == 0)            # ... new rows without data are zero (collective.c:3634)
    # new rows: a slice of other users, fewer rows of side information than of X
    row2, col2, val2 = make_coo(300, n, 5000, 9, counts=False, dtype=dtype, empty_rows=(2, 250))
    U2 = rng.standard_normal((260, p)).astype(dtype) if side else None
    A, bias = mdl.factors_multiple((row2, col2, val2), U=U2, return_bias=True)
    kw = dict(B=mdl.B_, row=row2, col=col2, val=val2, m=300, k=k, biasB=mdl.item_bias_, glob_mean=mdl.glob_mean_,
              user_bias=True, lam=0.8, scale_lam=True, w_user=2.0, nthreads=4)
    if side:
        kw.update(Cm=mdl.C_, U=U2, U_colmeans=mdl._U_colmeans, TransCtCinvCt=mdl._TransCtCinvCt)
    Ao, bo = O.factors_explicit_multiple(**kw)
    assert A.shape == (300, k) and gc.maxrel(A, Ao) < t and gc.maxrel(bias, bo) < t

    row, col, val = make_coo(m, n, 20000, 10, counts=True, dtype=dtype, heavy_row=(5, 300), empty_rows=() if side else (9,))
    X = sp.coo_matrix((val, (row, col)), shape=(m, n))
    mi = CMF_implicit(k=k, lambda_=2.0, alpha=1.5, niter=3, use_cg=False, use_float=dtype is np.float32, w_user=2.0,
                      ).fit(X, U=U, A0=A0, B0=B0)
    assert gc.maxrel(mi.factors_multiple(X, U=U)[ne], mi.A_[ne]) < t
    row2, col2, val2 = make_coo(300, n, 5000, 11, counts=True, dtype=dtype, empty_rows=(2, 250))
    A = mi.factors_multiple((row2, col2, val2), U=U2)
    kw = dict(B=mi.B_, row=row2, col=col2, val=val2, m=300, k=k, lam=2.0, alpha=1.5, w_user=2.0, nthreads=4, BtB=mi._BtB)
    if side:
        kw.update(Cm=mi.C_, U=U2, U_colmeans=mi._U_colmeans)
    assert gc.maxrel(A, O.factors_implicit_multiple(**kw)) < t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_factors_multiple_l1_after_fit(dtype):
    """New rows under an L1 penalty (solve_elasticnet behind factors_collective_*_multiple; the reference's outputs are the
    g22 fixture, tests/test_gpu_golden.py::test_new_rows_l1): a fit whose last A-step ran the same elastic-net sweeps leaves
    exactly the factors the new-rows entry point computes for its training rows -- explicit model with biases and both penalties
    per matrix, implicit model without side information (solved on the symmetric system, see tests/golden_cases.py)."""
    import scipy.sparse as sp
    import golden_cases as gc
    from cmfrec_amd import CMF, CMF_implicit
    t = 1e-8 if dtype is np.float64 else 2e-3
    m, n, k = 500, 300, 12
    rng = np.random.default_rng(15)
    row, col, val = make_coo(m, n, 12000, 18, counts=False, dtype=dtype, heavy_row=(5, 200))
    X = sp.coo_matrix((val, (row, col)), shape=(m, n))
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    mdl = CMF(k=k, lambda_=0.5, l1_lambda=[0.01, 0.02, 0.05, 0.04, 0.03, 0.03], niter=3, use_float=dtype is np.float32,
              scale_lam=True, max_cd_steps=1000, precompute_for_predictions=False).fit(X, A0=A0, B0=B0)
    A, bias = mdl.factors_multiple(X, return_bias=True)
    assert (mdl.A_ == 0).mean() > 0.02                       # the penalty is at work
    assert gc.maxrel(A, mdl.A_) < t and gc.maxrel(bias, mdl.user_bias_) < t
    row, col, val = make_coo(m, n, 12000, 19, counts=True, dtype=dtype, heavy_row=(5, 200))
    X = sp.coo_matrix((val, (row, col)), shape=(m, n))
    mi = CMF_implicit(k=k, lambda_=2.0, l1_lambda=3.0, alpha=1.5, niter=3, use_float=dtype is np.float32,
                      max_cd_steps=1000).fit(X, A0=A0, B0=B0)
    A = mi.factors_multiple(X)
    assert np.isfinite(A).all()
    assert gc.maxrel(A, mi.A_) < t
    assert (mi.A_ == 0).mean() > 0.02                         # the penalty is at work


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("side", [False, True])
def test_implicit_adjust_weight_matches_reference(dtype, side):
    """adjust_weight of the implicit model (collective.c:9776-9811): w_main is multiplied by nnz / (m n), which rescales lambda
    and the side-information weights; the multiplier is an output.  Seeded like the reference."""
    from oracle.bindings import Reference, ref_available
    if not ref_available(dtype):
        pytest.skip("oracle/_ref not built")
    from cmfrec_amd import CMF_implicit
    R = Reference(dtype)
    m, n, k = 600, 400, 16
    row, col, val = make_coo(m, n, 15000, 37, counts=True, dtype=dtype)
    rng = np.random.default_rng(9)
    U = rng.standard_normal((m, 5)).astype(dtype) if side else None
    II = rng.standard_normal((n, 4)).astype(dtype) if side else None
    mdl = CMF_implicit(k=k, lambda_=2.0, niter=2, random_state=31, use_float=dtype is np.float32, nthreads=1, use_cg=False,
                       w_user=3.0, w_item=0.5, precompute_for_predictions=False)
    mdl._adjust_weight = True
    mdl.fit((row, col, val), U=U, I=II, shape=(m, n))
    Ar, Br = np.zeros((m, k), dtype), np.zeros((n, k), dtype)
    rr = R.fit_collective_implicit_als(Ar, Br, row, col, val, k, lam=2.0, niter=2, nthreads=2, reset_values=True, seed=31, use_cg=False,
                                       U=U, II=II, w_user=3.0, w_item=0.5, adjust_weight=True)
    t = 1e-6 if dtype is np.float64 else 1e-2
    assert abs(mdl._w_main_multiplier - len(val) / (m * n)) < 1e-6 and abs(mdl._w_main_multiplier - float(rr["w_main_multiplier"])) < 1e-7
    assert frob(mdl.A_, Ar) < t and frob(mdl.B_, Br) < t
    plain = CMF_implicit(k=k, lambda_=2.0, niter=2, random_state=31, use_float=dtype is np.float32, nthreads=1, use_cg=False,
                         w_user=3.0, w_item=0.5, precompute_for_predictions=False).fit((row, col, val), U=U, I=II, shape=(m, n))
    assert frob(plain.A_, Ar) > 1e-2


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("pattern,use_cg", [("full", True), ("sparse-ish", True), ("sparse-ish", False)])
def test_fit_dense_X_medium(oracles, dtype, pattern, use_cg):
    """Dense X at a size where rows fall into every length bin (1600 x 1200: complete rows of 1200 entries are split rows, the
    70 %-missing pattern has rows of ~360): CMF.fit(X = array with NaN) against the oracle on the present entries -- closed form
    for the complete matrix whatever use_cg says (optimizeA Case 1), the solver asked for where every row misses many entries
    (Case 2).  (Small patterns against the reference itself: tests/test_gpu_golden.py::test_dense_X.)"""
    from cmfrec_amd import CMF
    O = oracles[dtype]
    m, n, k = 1600, 1200, 16
    rng = np.random.default_rng(8)
    X = (0.5 * rng.integers(1, 11, (m, n))).astype(dtype)
    if pattern != "full":
        X[rng.random((m, n)) < 0.7] = np.nan
    row, col = [a.astype(np.int32) for a in np.nonzero(~np.isnan(X))]
    val = X[row, col]
    A0 = (rng.standard_normal((m, k)) * 0.01).astype(dtype); B0 = (rng.standard_normal((n, k)) * 0.01).astype(dtype)
    bA = (rng.standard_normal(m) * 0.1).astype(dtype); bB = (rng.standard_normal(n) * 0.1).astype(dtype)
    mdl = CMF(k=k, lambda_=2.0, niter=2, use_cg=use_cg, finalize_chol=False, use_float=dtype is np.float32, nthreads=1,
              precompute_for_predictions=False).fit(X, A0=A0, B0=B0, biasA0=bA, biasB0=bB)
    Ao, Bo = A0.copy(), B0.copy()
    cg_o = use_cg and pattern != "full"
    ro = O.fit_explicit_als(Ao, Bo, row, col, val, k, biasA=bA.copy(), biasB=bB.copy(), lam=2.0, niter=2, nthreads=2, use_cg=cg_o,
                            finalize_chol=False)
    assert ro["ret"] == 0
    t = tol(dtype, "cg" if cg_o else "chol")
    assert frob(mdl.A_, Ao) < t and frob(mdl.B_, Bo) < t
    assert frob(mdl.user_bias_, ro["biasA"]) < t and frob(mdl.item_bias_, ro["biasB"]) < t
    assert abs(mdl.glob_mean_ - ro["glob_mean"]) <= 1e-6 * abs(ro["glob_mean"])
