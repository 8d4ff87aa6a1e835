"""Shared drivers for the golden-fixture tests: every case runs an engine (oracle on CPU, HIP on
GPU) on the stored inputs and returns (name, got, expected) triples."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = {np.float64: "f64", np.float32: "f32"}


def load(name, dtype):
    return np.load(os.path.join(GOLD, "%s_%s.npz" % (name, TAGS[dtype])))


def csr_from(g, O):
    return O.coo_to_csr_and_csc(g["row"], g["col"], g["val"], int(g["m"]), int(g["n"]))


def implicit_cases(g, O, op_implicit, modes=("cg", "pcg", "chol")):
    csr, _ = csr_from(g, O)
    for k in (8, 50, 64):
        for mode in modes:
            A = g["A0_k%d" % k].copy()
            op_implicit(A, g["B_k%d" % k], csr, float(g["lam"]), use_cg=mode != "chol", precondition_cg=mode == "pcg",
                        max_cg_steps=3)
            yield "implicit %s k=%d" % (mode, k), A, g["A_%s_k%d" % (mode, k)]


def explicit_cases(g, O, op_explicit, modes=("cg", "pcg", "chol")):
    csr, _ = csr_from(g, O)
    for k in (51, 17):
        for mode in modes:
            A = g["A0_k%d" % k].copy()
            op_explicit(A, g["B_k%d" % k], csr, float(g["lam"]), lam_last=float(g["lam_last"]), k=k, scale_lam=True,
                        use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
            yield "explicit %s k=%d" % (mode, k), A, g["A_%s_k%d" % (mode, k)]


def collective_cases(g, O, op_collective):
    csr, _ = csr_from(g, O)
    for ci in (0, 1):
        p, k, ku, ki, km, sls = [int(x) for x in g["cfg_%d" % ci]]
        A = g["A0_%d" % ci].copy()
        op_collective(A, g["B_%d" % ci], g["C_%d" % ci], csr, g["U_%d" % ci], float(g["lam"]), w_user=float(g["w_user"]),
                      lam_last=float(g["lam_last"]), k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=True,
                      scale_lam_sideinfo=bool(sls))
        kA = ku + k + km
        yield "collective cfg %d" % ci, A[:, :kA], g["A_%d" % ci][:, :kA]


def frob(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def maxrel(a, b):
    dt = np.asarray(a).dtype
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    log = os.environ.get("CMFREC_TEST_RELERR_LOG")       # same log as conftest.rel_err
    if log:
        with open(log, "a") as f:
            f.write("%s %s %.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], dt, e))
    return e


# ---- result metrics of the reference's benchmarks (SURVEY.md 8c G7, 8d "results parity") ----
def rmse(A, B, biasA, biasB, glob_mean, urow, icol, y):
    """Root mean squared error of  glob_mean + biasA[u] + biasB[i] + A_u . B_i  on held-out pairs
    (reference predict_multiple, src/common.c:5098-5106; benchmark_explicit_cmfrec.ipynb)."""
    pred = np.einsum("ij,ij->i", np.asarray(A, np.float64)[urow], np.asarray(B, np.float64)[icol]) + float(glob_mean)
    if biasA is not None and len(biasA):
        pred = pred + np.asarray(biasA, np.float64)[urow]
    if biasB is not None and len(biasB):
        pred = pred + np.asarray(biasB, np.float64)[icol]
    return float(np.sqrt(np.mean((pred - np.asarray(y, np.float64)) ** 2)))


def precision_at_k(A, B, train_row, train_col, test_row, test_col, k=10):
    """P@k as the reference's implicit benchmark computes it (benchmark_implicit_cmfrec.ipynb, cell 3): for every
    user with held-out items, rank all items by A_u . B_i, drop the user's training items, take the top k and
    count the fraction that are held-out items of that user; mean over those users."""
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    n = B.shape[0]
    seen = {}
    for u, i in zip(train_row, train_col):
        seen.setdefault(int(u), []).append(int(i))
    held = {}
    for u, i in zip(test_row, test_col):
        held.setdefault(int(u), set()).add(int(i))
    hits = []
    for u, items in sorted(held.items()):
        s = B @ A[u]
        s[seen.get(u, [])] = -np.inf
        top = np.argpartition(-s, min(k, n - 1))[:k]
        hits.append(len(items.intersection(top.tolist())) / float(k))
    return float(np.mean(hits))


# ---- factors of new rows (factors_collective_*_multiple) -------------------------------------------------
def new_rows_problem(dtype, k, seed=11):
    """Seeded model (B, C, biases, column means) + new rows (COO with empty rows, dense U) for a given k."""
    rng = np.random.default_rng(seed + k)
    n, m, p = 70, 48, 5
    ku, ki, km = 2, 1, 1
    d = dict(n=n, m=m, p=p, k=k, ku=ku, ki=ki, km=km)
    d["B_plain"] = (rng.standard_normal((n, k)) * 0.3).astype(dtype)
    d["B_full"] = (rng.standard_normal((n, ki + k + km)) * 0.3).astype(dtype)
    d["C_plain"] = (rng.standard_normal((p, k)) * 0.3).astype(dtype)
    d["C_full"] = (rng.standard_normal((p, ku + k)) * 0.3).astype(dtype)
    d["biasB"] = (rng.standard_normal(n) * 0.2).astype(dtype)
    d["U_more"] = rng.standard_normal((m + 7, p)).astype(dtype)          # m_u > m
    d["U_less"] = rng.standard_normal((m - 9, p)).astype(dtype)          # m_u < m
    d["colmeans"] = (rng.standard_normal(p) * 0.1).astype(dtype)
    lin = rng.choice(m * n, size=14 * m, replace=False)
    row = (lin // n).astype(np.int32); col = (lin % n).astype(np.int32)
    keep = (row != 3) & (row != 17) & (row != m - 2)                      # empty rows, with and without U
    d["row"], d["col"] = row[keep], col[keep]
    d["ratings"] = (0.5 * rng.integers(1, 11, keep.sum())).astype(dtype)
    d["counts"] = np.ceil(rng.lognormal(1, 1, keep.sum())).astype(dtype)
    return d


def new_rows_cases(d):
    """(name, kind, kwargs) for every configuration; kwargs fit Oracle/Reference.factors_{explicit,implicit}_multiple."""
    k, ku, ki, km = d["k"], d["ku"], d["ki"], d["km"]
    X = (d["row"], d["col"])
    lam_c = 0.7 / 2.5
    Cf = d["C_full"].astype(np.float64)
    T = np.linalg.solve(Cf.T @ Cf + lam_c * np.eye(ku + k), Cf.T).T.astype(d["C_full"].dtype)   # [p, ku+k]
    ex = [
        ("e0 bias scale_lam lam_bias", dict(B=d["B_plain"], biasB=d["biasB"], glob_mean=3.1, user_bias=True, lam=0.6,
                                            lam_bias=1.1, scale_lam=True)),
        ("e1 plain w_main", dict(B=d["B_plain"], lam=2.0, w_main=1.5)),
        ("e2 U>m bias both scalings k_*", dict(B=d["B_full"], Cm=d["C_full"], U=d["U_more"], U_colmeans=d["colmeans"],
                                                biasB=d["biasB"], glob_mean=3.1, user_bias=True, lam=0.7, lam_bias=1.3,
                                                k_main=km, k_user=ku, k_item=ki, scale_lam=True, scale_lam_sideinfo=True,
                                                w_main=1.5, w_user=2.5)),
        ("e3 U<m scale_lam", dict(B=d["B_plain"], Cm=d["C_plain"], U=d["U_less"], glob_mean=-0.4, lam=0.7, scale_lam=True,
                                  w_user=0.8)),
        ("e4 bias scale_bias_const", dict(B=d["B_plain"], biasB=d["biasB"], user_bias=True, lam=0.6, lam_bias=0.9,
                                          scale_lam=True, scale_bias_const=True, scaling_biasA=0.4)),
        ("e5 U TransCtCinvCt", dict(B=d["B_full"], Cm=d["C_full"], U=d["U_more"], U_colmeans=d["colmeans"], user_bias=True,
                                    lam=0.7, k_main=km, k_user=ku, k_item=ki, w_user=2.5, TransCtCinvCt=T)),
        ("e6 U only sideinfo scaling", dict(B=d["B_plain"], Cm=d["C_plain"], U=d["U_more"], lam=0.7,
                                            scale_lam_sideinfo=True, scale_bias_const=True, user_bias=True, scaling_biasA=0.5,
                                            w_user=1.7)),
    ]
    Bf = d["B_full"].astype(np.float64)[:, ki:]
    BtB = (Bf.T @ Bf + 0.45 * np.eye(k + km)).astype(d["B_full"].dtype)
    im = [
        ("i0 plain alpha", dict(B=d["B_plain"], lam=0.7, alpha=2.0)),
        ("i1 U>m w_main k_*", dict(B=d["B_full"], Cm=d["C_full"], U=d["U_more"], U_colmeans=d["colmeans"], lam=0.7, alpha=2.0,
                                   k_main=km, k_user=ku, k_item=ki, w_main=1.5, w_user=2.5, w_main_multiplier=0.8)),
        # (apply_log_transf cannot be pinned: the reference's row function takes the logarithm of a freshly allocated,
        #  uninitialised buffer, collective.c:10802-10810)
        ("i2 U<m BtB", dict(B=d["B_full"], Cm=d["C_full"], U=d["U_less"], lam=0.45, k_main=km, k_user=ku, k_item=ki,
                            w_user=4.0, BtB=BtB)),
        ("i3 plain w_main", dict(B=d["B_plain"], lam=0.7, w_main=2.0)),
    ]
    # sparse side information for the new rows (missing = absent): attributes of U_less' shape, two thirds dropped.
    rng = np.random.default_rng(5)
    mask = rng.random(d["U_less"].shape) < 0.35
    mask[4] = False; mask[3, :2] = True                     # row 4: observations only; row 3: attributes only
    ur, uc = np.nonzero(mask)
    U_coo = (ur.astype(np.int32), uc.astype(np.int32), d["U_less"][mask], d["U_less"].shape[0], d["U_less"].shape[1])
    ex.append(("e7 sparse U bias scale_lam", dict(B=d["B_full"], Cm=d["C_full"], U_coo=U_coo, biasB=d["biasB"], glob_mean=3.1,
                                                  user_bias=True, lam=0.7, lam_bias=1.3, k_main=km, k_user=ku, k_item=ki,
                                                  scale_lam=True, w_main=1.5, w_user=2.5)))
    im.append(("i4 sparse U", dict(B=d["B_full"], Cm=d["C_full"], U_coo=U_coo, lam=0.7, alpha=2.0, k_main=km, k_user=ku,
                                   k_item=ki, w_user=2.5)))
    for name, kw in ex:
        yield name, "explicit", dict(row=X[0], col=X[1], val=d["ratings"], m=d["m"], k=k, **kw)
    for name, kw in im:
        yield name, "implicit", dict(row=X[0], col=X[1], val=d["counts"], m=d["m"], k=k, **kw)


def new_rows_l1_cases(d):
    """New rows under an L1 penalty (solve_elasticnet behind factors_collective_*_multiple): pinned by fixtures from the real
    reference only (g22) -- the oracle does not restate the penalty for new rows."""
    k, ku, ki, km = d["k"], d["ku"], d["ki"], d["km"]
    X = (d["row"], d["col"])
    ex = [
        ("l0 bias scale_lam l1_bias", dict(B=d["B_plain"], biasB=d["biasB"], glob_mean=3.1, user_bias=True, lam=0.6, lam_bias=1.1,
                                           scale_lam=True, l1_lam=0.05, l1_lam_bias=0.02)),
        ("l1 plain w_main", dict(B=d["B_plain"], lam=2.0, w_main=1.5, l1_lam=0.4)),
        ("l2 U>m bias both scalings k_*", dict(B=d["B_full"], Cm=d["C_full"], U=d["U_more"], U_colmeans=d["colmeans"],
                                                biasB=d["biasB"], glob_mean=3.1, user_bias=True, lam=0.7, lam_bias=1.3, k_main=km,
                                                k_user=ku, k_item=ki, scale_lam=True, scale_lam_sideinfo=True, w_main=1.5,
                                                w_user=2.5, l1_lam=0.03, l1_lam_bias=0.01)),
        ("l3 U<m", dict(B=d["B_plain"], Cm=d["C_plain"], U=d["U_less"], glob_mean=-0.4, lam=0.7, w_user=0.8, l1_lam=0.2)),
    ]
    # Not pinned, implicit model: without side information factors_implicit_chol hands solve_elasticnet a matrix whose lower
    # triangle was never filled (common.c:2106-2115, fill_lower = false; the reference's rows end in +-inf for k = 50), with side
    # information its sweeps diverge on the block system (+-inf in most rows).  The HIP entry point solves the symmetric system
    # for the first (what its own fit does, tests/test_gpu_fit.py::test_factors_multiple_l1_after_fit) and refuses the second.
    for name, kw in ex:
        yield name, "explicit", dict(row=X[0], col=X[1], val=d["ratings"], m=d["m"], k=k, **kw)


def run_new_rows(engine, kind, kw):
    """Returns (A, biasA or None)."""
    if kind == "explicit":
        return engine.factors_explicit_multiple(**kw)
    return engine.factors_implicit_multiple(**kw), None


class HipNewRows:
    """The product's factors_collective_{explicit,implicit}_multiple (the reference's positional C signatures) behind
    the keyword interface of Oracle / Reference."""

    def __init__(self, dtype):
        import ctypes as C
        from cmfrec_amd import _lib
        self.C, self._lib, self.dtype = C, _lib, dtype
        self.lib = _lib.load(dtype)
        self.R = _lib.real(dtype)

    def factors_explicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, biasB=None, glob_mean=0.0,
                                  user_bias=False, lam=1.0, lam_bias=None, k_main=0, k_user=0, k_item=0, scale_lam=False,
                                  scale_lam_sideinfo=False, scale_bias_const=False, scaling_biasA=1.0, w_main=1.0,
                                  w_user=1.0, nthreads=1, TransCtCinvCt=None, csr=None, U_coo=None, l1_lam=0.0, l1_lam_bias=None):
        C, P, R = self.C, self._lib.ptr, self.R
        n = B.shape[0]
        m_u, p = (0, 0) if U is None else U.shape
        su = (None, None, None, C.c_size_t(0))
        if U_coo is not None:
            keep_u = (np.ascontiguousarray(U_coo[0], np.int32), np.ascontiguousarray(U_coo[1], np.int32),
                      np.ascontiguousarray(U_coo[2], self.dtype))
            su = (P(keep_u[0]), P(keep_u[1]), P(keep_u[2]), C.c_size_t(len(keep_u[2]))); m_u, p = U_coo[3], U_coo[4]
        mm = max(m, m_u)
        A = np.full((mm, k_user + k + k_main), np.nan, self.dtype)
        biasA = np.full(mm, np.nan, self.dtype) if user_bias else None
        lam_unique = None
        if lam_bias is not None and lam_bias != lam:
            lam_unique = np.zeros(6, self.dtype); lam_unique[0] = lam_bias; lam_unique[2] = lam
        l1_unique = None
        if l1_lam_bias is not None and l1_lam_bias != l1_lam:
            l1_unique = np.zeros(6, self.dtype); l1_unique[0] = l1_lam_bias; l1_unique[2] = l1_lam
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        coo = (P(val), P(row), P(col), C.c_size_t(len(val)), None, None, None)
        if csr is not None:
            coo = (None, None, None, C.c_size_t(0), P(csr[0]), P(csr[1]), P(csr[2]))
        rc = self.lib.factors_collective_explicit_multiple(
            P(A), P(biasA), C.c_int(m), P(U), C.c_int(m_u), C.c_int(p), C.c_bool(False), C.c_bool(False), C.c_bool(False),
            *su, None, None, None, None, C.c_int(0), C.c_int(0),
            P(Cm), None, R(glob_mean), P(biasB), P(U_colmeans), *coo,
            None, C.c_int(n), None, P(B), None, C.c_bool(False),
            C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            R(lam), P(lam_unique), R(l1_lam), P(l1_unique), C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo),
            C.c_bool(scale_bias_const), R(scaling_biasA), R(w_main), R(w_user), R(1.), C.c_int(n), C.c_bool(True),
            None, None, None, None, None, P(TransCtCinvCt), None, None, None, C.c_int(nthreads))
        assert rc == 0, (rc, self.lib.cmfrec_hip_last_error())
        return A, biasA

    def factors_implicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, lam=1.0, alpha=1.0,
                                  k_main=0, k_user=0, k_item=0, w_main=1.0, w_user=1.0, w_main_multiplier=1.0,
                                  apply_log_transf=False, nthreads=1, BtB=None, csr=None, U_coo=None, l1_lam=0.0):
        C, P, R = self.C, self._lib.ptr, self.R
        n = B.shape[0]
        m_u, p = (0, 0) if U is None else U.shape
        su = (None, None, None, C.c_size_t(0))
        if U_coo is not None:
            keep_u = (np.ascontiguousarray(U_coo[0], np.int32), np.ascontiguousarray(U_coo[1], np.int32),
                      np.ascontiguousarray(U_coo[2], self.dtype))
            su = (P(keep_u[0]), P(keep_u[1]), P(keep_u[2]), C.c_size_t(len(keep_u[2]))); m_u, p = U_coo[3], U_coo[4]
        A = np.full((max(m, m_u), k_user + k + k_main), np.nan, self.dtype)
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        coo = (P(val), P(row), P(col), C.c_size_t(len(val)), None, None, None)
        if csr is not None:
            coo = (None, None, None, C.c_size_t(0), P(csr[0]), P(csr[1]), P(csr[2]))
        rc = self.lib.factors_collective_implicit_multiple(
            P(A), C.c_int(m), P(U), C.c_int(m_u), C.c_int(p), C.c_bool(False), C.c_bool(False),
            *su, None, None, None, *coo,
            P(B), C.c_int(n), P(Cm), P(U_colmeans), C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            R(lam), R(l1_lam), R(alpha), R(w_main), R(w_user), R(w_main_multiplier), C.c_bool(apply_log_transf),
            None, P(BtB), None, None, C.c_int(nthreads))
        assert rc == 0, (rc, self.lib.cmfrec_hip_last_error())
        return A


def new_rows_vs_golden(engine, dtype, extra=None):
    """Yields (label, error) of an engine against the g11 fixture (the reference's outputs)."""
    g = load("g11_new_rows", dtype)
    for k in (6, 50):
        d = new_rows_problem(dtype, k)
        for name, kind, kw in new_rows_cases(d):
            A, bA = run_new_rows(engine, kind, dict(kw, **(extra or {})))
            key = "k%d_%s" % (k, name.split()[0])
            yield "k=%d %s" % (k, name), maxrel(A, g["A_" + key])
            if bA is not None:
                yield "k=%d %s (bias)" % (k, name), maxrel(bA, g["biasA_" + key])


def new_rows_l1_vs_golden(engine, dtype):
    """Yields (label, error) of an engine against the g22 fixture (new rows under an L1 penalty, the reference's outputs)."""
    g = load("g22_new_rows_l1", dtype)
    for k in (6, 50):
        d = new_rows_problem(dtype, k)
        for name, kind, kw in new_rows_l1_cases(d):
            A, bA = run_new_rows(engine, kind, kw)
            key = "k%d_%s" % (k, name.split()[0])
            yield "k=%d %s" % (k, name), maxrel(A, g["A_" + key])
            if bA is not None:
                yield "k=%d %s (bias)" % (k, name), maxrel(bA, g["biasA_" + key])


# ---- sparse side information (missing = absent), Cholesky updates -----------------------------------------------------
def sparse_sideinfo_problem(dtype, seed=41):
    """Seeded problem for fits with sparse U / I: returns dict with X triplets (ratings and counts), side-information COO
    tuples (row, col, val, rows, cols) and start values."""
    rng = np.random.default_rng(seed)
    m, n, k, p, q, m_u, n_i = 90, 70, 6, 9, 7, 80, 70
    ku, ki, km = 2, 1, 1
    d = dict(m=m, n=n, k=k, ku=ku, ki=ki, km=km)
    def coo(rows, cols, cnt, empty):
        lin = rng.choice(rows * cols, size=cnt, replace=False)
        r = (lin // cols).astype(np.int32); c = (lin % cols).astype(np.int32)
        keep = ~np.isin(r, empty)
        return r[keep], c[keep]
    ur, uc = coo(m_u, p, 280, (3, 5)); ir, ic = coo(n_i, q, 220, (4,))
    d["U_coo"] = (ur, uc, rng.standard_normal(len(ur)).astype(dtype), m_u, p)
    d["I_coo"] = (ir, ic, rng.standard_normal(len(ir)).astype(dtype), n_i, q)
    xr, xc = coo(m, n, 1100, (3, 7, 85))
    d["row"], d["col"] = xr, xc
    d["ratings"] = (0.5 * rng.integers(1, 11, len(xr))).astype(dtype)
    d["counts"] = np.ceil(rng.lognormal(1, 1, len(xr))).astype(dtype)
    d["A0"] = (rng.standard_normal((m, ku + k + km)) * 0.1).astype(dtype); d["B0"] = (rng.standard_normal((n, ki + k + km)) * 0.1).astype(dtype)
    d["C0"] = (rng.standard_normal((p, ku + k)) * 0.1).astype(dtype); d["D0"] = (rng.standard_normal((q, ki + k)) * 0.1).astype(dtype)
    d["bA"] = (rng.standard_normal(m) * 0.1).astype(dtype); d["bB"] = (rng.standard_normal(n) * 0.1).astype(dtype)
    return d


SPARSE_SIDE_CASES = [("implicit UI", True, "UI", False, False), ("explicit UI", False, "UI", False, False),
                     ("explicit UI scaled", False, "UI", True, True), ("explicit U", False, "U", True, False),
                     ("implicit I", True, "I", False, False)]
# the same problems with the CG solvers: (name, implicit, sides, scale_lam, scale_lam_sideinfo, solver kwargs)
SPARSE_SIDE_CG_CASES = [("implicit UI cg", True, "UI", False, False, dict(use_cg=True)),
                        ("explicit UI cg+finalize", False, "UI", True, True, dict(use_cg=True, finalize_chol=True)),
                        ("explicit U pcg", False, "U", True, False, dict(use_cg=True, precondition_cg=True)),
                        # (implicit PCG with k_item > 0: rows without attributes divide by a zero preconditioner in the reference,
                        #  collective.c:2993-3003, NaN on both sides -- so this one runs without item-only factors)
                        ("implicit I pcg", True, "I", False, False, dict(use_cg=True, precondition_cg=True, no_k_side=True))]


def sparse_sideinfo_reference(R, d, implicit, which, sl, sls, nthreads=2, solver=None):
    """The real reference on the problem; returns dict(A, B, C, D, biasA, biasB, glob_mean)."""
    solver = dict(solver or {}); nks = solver.pop("no_k_side", False)
    ku = d["ku"] if ("U" in which and not nks) else 0; ki = d["ki"] if ("I" in which and not nks) else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver or {})
    cz = 0 if sv["use_cg"] else 1          # CG warm-starts C / D: the estimators start them at zero, so do these runs
    kw = dict(k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads, **sv,
              U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
              Cm=(d["C0"][:, d["ku"] - ku:] * cz).copy() if "U" in which else None,
              Dm=(d["D0"][:, d["ki"] - ki:] * cz).copy() if "I" in which else None)
    if implicit:
        r = R.fit_collective_implicit_als(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, w_main=0.5, **kw)
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"])
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **kw)
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])


def sparse_sideinfo_oracle(O, d, implicit, which, sl, sls, nthreads=2, solver=None):
    solver = dict(solver or {}); nks = solver.pop("no_k_side", False)
    ku = d["ku"] if ("U" in which and not nks) else 0; ki = d["ki"] if ("I" in which and not nks) else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    cz = 0 if (solver or {}).get("use_cg") else 1
    kw = dict(k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads, **(solver or {}),
              U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
              Cm=(d["C0"][:, d["ku"] - ku:] * cz).copy() if "U" in which else None,
              Dm=(d["D0"][:, d["ki"] - ki:] * cz).copy() if "I" in which else None)
    if implicit:
        return O.fit_als_sparse_sideinfo(A0, B0, d["row"], d["col"], d["counts"], d["k"], True, lam=2.0, alpha=1.5, w_main=0.5, **kw)
    return O.fit_als_sparse_sideinfo(A0, B0, d["row"], d["col"], d["ratings"], d["k"], False, biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                     user_bias=True, item_bias=True, center=True, lam=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **kw)


def sparse_sideinfo_hip(d, implicit, which, sl, sls, dtype, solver=None):
    """The product: the estimators with SciPy sparse side information."""
    import scipy.sparse as sp
    from cmfrec_amd import CMF, CMF_implicit
    solver = dict(solver or {}); nks = solver.pop("no_k_side", False)
    ku = d["ku"] if ("U" in which and not nks) else 0; ki = d["ki"] if ("I" in which and not nks) else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    U = mk(d["U_coo"]) if "U" in which else None; I = mk(d["I_coo"]) if "I" in which else None
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver or {})
    common = dict(k=d["k"], k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3,
                  use_float=dtype is np.float32, precompute_for_predictions=False, **sv)
    shape = (d["m"], d["n"])
    if implicit:
        mdl = CMF_implicit(lambda_=2.0, alpha=1.5, w_main=0.5, **common)
        # start values of C / D cannot be injected through the estimator: the first C / D update overwrites them anyway
        mdl.fit((d["row"], d["col"], d["counts"]), U=U, I=I, shape=shape, A0=A0, B0=B0)
        return dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_)
    mdl = CMF(lambda_=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **common)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=I, shape=shape, A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    return dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, biasA=mdl.user_bias_, biasB=mdl.item_bias_, glob_mean=mdl.glob_mean_)


# ---- NA_as_zero_X with precompute_for_predictions (model without side information): fixture g27 -- the fit's factors and the matrices
# B_plus_bias, BtB, TransBtBinvBt, BtXbias (collective.c:8936-9082)
NAZ_PRE_CASES = [
    ("chol, biases", dict(use_cg=False)),
    ("cg + finalize (the defaults), scale_lam", dict(use_cg=True, finalize_chol=True, scale_lam=True)),
    ("no centring, item bias only", dict(use_cg=False, center=False, user_bias=False)),
    ("no biases, centred", dict(use_cg=False, user_bias=False, item_bias=False)),
]


def naz_pre_reference(R, d, opts, nthreads=2):
    o = dict(opts)
    A0, B0 = _impf_start(d, o)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, nthreads=nthreads, NA_as_zero_X=True, precompute=True, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], BtXbias=r["pre"]["BtXbias"], BtB=r["pre"]["BtB"], TransBtBinvBt=r["pre"]["TransBtBinvBt"])
    if o.get("user_bias", True): out["B_plus_bias"] = r["pre"]["B_plus_bias"]
    return out


def naz_pre_hip(d, opts, dtype):
    from cmfrec_amd import CMF
    o = dict(opts)
    A0, B0 = _impf_start(d, o)
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, use_float=dtype is np.float32, NA_as_zero=True, nthreads=1, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    return dict(A=mdl.A_, B=mdl.B_, BtXbias=mdl._BtXbias, BtB=mdl._BtB, TransBtBinvBt=mdl._TransBtBinvBt, B_plus_bias=mdl._B_plus_bias)


# ---- NA_as_zero_X together with implicit features, no side information (optimizeA_collective's general branch on a matrix all rows
# share: B^T B + w_i Bi^T Bi + lam mult I; collective.c:8612 / :8783 -> :1534-1846) -- fixture g26, closed form, the problem of g18
NAZ_IMPF_CASES = [
    ("biases", dict()),
    ("scale_lam", dict(scale_lam=True)),
    ("no biases, no centring", dict(user_bias=False, item_bias=False, center=False)),
    ("user bias, no centring, k_main, w_implicit", dict(center=False, item_bias=False, k_main=2, w_implicit=0.6)),
    ("item bias", dict(user_bias=False)),
    ("per-matrix lambdas", dict(lam_unique=[0.7, 0.2, 0.4, 0.25, 1.5, 0.6])),
]


def naz_impf_reference(R, d, opts, nthreads=2):
    o = dict(opts)
    A0, B0 = _impf_start(d, o)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, nthreads=nthreads, NA_as_zero_X=True, use_cg=False, add_implicit_features=True, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], Ai=r["Ai"], Bi=r["Bi"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_impf_oracle(O, d, opts, nthreads=2):
    o = dict(opts)
    lam6 = o.pop("lam_unique", None)
    if lam6 is not None:
        O.set_lam_unique(np.asarray(lam6, np.float64), None)
    try:
        A0, B0 = _impf_start(d, o)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=3, nthreads=nthreads, NA_as_zero_X=True, use_cg=False, add_implicit_features=True, **o)
    finally:
        O.set_lam_unique(None, None)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], Ai=r["Ai"], Bi=r["Bi"], glob_mean=r["glob_mean"])
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_impf_hip(d, opts, dtype, **ctor):
    from cmfrec_amd import CMF
    o = dict(opts)
    o.setdefault("w_implicit", 1.0)           # (the estimator's default is 0.5, the C default used by the reference calls above 1)
    if "lam_unique" in o:
        o["lambda_"] = o.pop("lam_unique")
    else:
        o["lambda_"] = 0.3
    A0, B0 = _impf_start(d, o)
    kw = dict(k=d["k"], niter=3, use_float=dtype is np.float32, precompute_for_predictions=False, NA_as_zero=True, use_cg=False,
              add_implicit_features=True, nthreads=1)
    kw.update(o); kw.update(ctor)
    mdl = CMF(**kw)
    mdl.fit((d["row"], d["col"], d["ratings"]), shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, Ai=mdl.Ai_, Bi=mdl.Bi_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero_X together with SPARSE side information (collective_closed_form_block's general branch with prefer_BtB,
# collective.c:1534-1846: the shared B^T B plus the rank-1 terms of the row's own attributes) -- fixture g25.  Side information on
# exactly the rows / columns of X, closed form.
def naz_sparse_side_problem(dtype, seed=47):
    d = sparse_sideinfo_problem(dtype, seed)
    rng = np.random.default_rng(seed + 1)
    m, n = d["m"], d["n"]
    p, q = d["U_coo"][4], d["I_coo"][4]
    def coo(rows, cols, cnt, empty):
        lin = rng.choice(rows * cols, size=cnt, replace=False)
        r = (lin // cols).astype(np.int32); c = (lin % cols).astype(np.int32)
        keep = ~np.isin(r, empty)
        return r[keep], c[keep]
    ur, uc = coo(m, p, 300, (3, 5)); ir, ic = coo(n, q, 220, (4,))           # (row 3: neither an entry of X nor an attribute)
    d["U_coo"] = (ur, uc, rng.standard_normal(len(ur)).astype(dtype), m, p)
    d["I_coo"] = (ir, ic, rng.standard_normal(len(ir)).astype(dtype), n, q)
    return d


# (name, sides with side information, options)
NAZ_SPARSE_SIDE_CASES = [
    ("both sides, biases", "UI", dict()),
    ("both sides, scale_lam", "UI", dict(scale_lam=True)),
    ("both sides, scale_lam_sideinfo, item bias", "UI", dict(scale_lam_sideinfo=True, user_bias=False)),
    ("no biases, no centring", "UI", dict(user_bias=False, item_bias=False, center=False)),
    ("user side only", "U", dict()),
    ("item side only, scale_lam, no k_item", "I", dict(scale_lam=True, no_k_side=True)),
    ("user bias, no centring", "UI", dict(center=False, item_bias=False)),
]


# ... under use_cg (round 6; fixture g34): collective_block_cg's NA_as_zero_X branches with the row's attributes as a sparse vector
NAZ_SPARSE_SIDE_CG_CASES = [
    ("cg, both sides", "UI", dict(use_cg=True, finalize_chol=False)),
    ("cg, scaled, finalize", "UI", dict(use_cg=True, finalize_chol=True, scale_lam=True, scale_lam_sideinfo=True)),
    ("pcg, user side, no biases", "U", dict(use_cg=True, precondition_cg=True, finalize_chol=False, user_bias=False, item_bias=False, center=False)),
    ("cg, item side, user bias", "I", dict(use_cg=True, finalize_chol=False, item_bias=False)),
]


def _naz_sparse_args(d, which, opts):
    o = dict(opts); nks = o.pop("no_k_side", False)
    ku = d["ku"] if ("U" in which and not nks) else 0; ki = d["ki"] if ("I" in which and not nks) else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    return o, ku, ki, A0, B0


def naz_sparse_side_reference(R, d, which, opts, nthreads=2):
    o, ku, ki, A0, B0 = _naz_sparse_args(d, which, opts)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                                      k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads,
                                      use_cg=o.pop("use_cg", False),
                                      U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
                                      NA_as_zero_X=True, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if "U" in which: out["C"] = r["C"]
    if "I" in which: out["D"] = r["D"]
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_sparse_side_oracle(O, d, which, opts, nthreads=2):
    o, ku, ki, A0, B0 = _naz_sparse_args(d, which, opts)
    ub, ib, ce = o.pop("user_bias", True), o.pop("item_bias", True), o.pop("center", True)
    r = O.fit_als_sparse_sideinfo(A0, B0, d["row"], d["col"], d["ratings"], d["k"], False, biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                  user_bias=ub, item_bias=ib, center=ce, lam=0.3, k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0,
                                  w_item=0.7, niter=3, nthreads=nthreads, use_cg=False,
                                  U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
                                  NA_as_zero_X=True, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if "U" in which: out["C"] = r["C"]
    if "I" in which: out["D"] = r["D"]
    if ub: out["biasA"] = r["biasA"]
    if ib: out["biasB"] = r["biasB"]
    return out


def naz_sparse_side_hip(d, which, opts, dtype, **ctor):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o, ku, ki, A0, B0 = _naz_sparse_args(d, which, opts)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    U = mk(d["U_coo"]) if "U" in which else None; I = mk(d["I_coo"]) if "I" in which else None
    kw = dict(k=d["k"], lambda_=0.3, k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, use_cg=False,
              use_float=dtype is np.float32, precompute_for_predictions=False, NA_as_zero=True, nthreads=1)
    kw.update(o); kw.update(ctor)
    mdl = CMF(**kw)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=I, shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, glob_mean=mdl.glob_mean_)
    if U is not None: out["C"] = mdl.C_
    if I is not None: out["D"] = mdl.D_
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero_U / NA_as_zero_I: sparse side information whose absent entries are zeros (collective.c:1277-1457, :5790-5836,
# C / D by optimizeA Case 3 with the column means as a rank-one correction, :8354-8441) ----------------------------------------
# (name, implicit, sides, scale_lam, scale_lam_sideinfo, solver kwargs); the problem of G12 (U covers 80 of the 90 users)
NAZ_UI_CASES = [("implicit UI", True, "UI", False, False, dict()),
                ("implicit UI cg", True, "UI", False, False, dict(use_cg=True)),
                ("explicit UI", False, "UI", False, False, dict()),
                ("explicit UI scaled, cg + finalize", False, "UI", True, True, dict(use_cg=True, finalize_chol=True)),
                ("explicit U scaled", False, "U", True, False, dict()),
                ("explicit I cg", False, "I", False, False, dict(use_cg=True)),
                ("implicit I", True, "I", False, False, dict())]


def zero_filled(coo, rows):
    """The dense matrix a COO tuple (row, col, val, rows, cols) with absent = 0 denotes, on `rows` rows."""
    M = np.zeros((rows, coo[4]), coo[2].dtype)
    np.add.at(M, (coo[0], coo[1]), coo[2])
    return M


def naz_ui_reference(R, d, implicit, which, sl, sls, solver, nthreads=2):
    """The real reference on G12's problem with NA_as_zero_U / NA_as_zero_I on the sides that are given."""
    solver = dict(solver or {})
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver)
    cz = 0 if sv["use_cg"] else 1
    kw = dict(k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads, **sv,
              U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
              NA_as_zero_U="U" in which, NA_as_zero_I="I" in which,
              Cm=(d["C0"][:, d["ku"] - ku:] * cz).copy() if "U" in which else None,
              Dm=(d["D0"][:, d["ki"] - ki:] * cz).copy() if "I" in which else None)
    if implicit:
        r = R.fit_collective_implicit_als(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, w_main=0.5, **kw)
        assert r["ret"] == 0
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"])
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **kw)
    assert r["ret"] == 0
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])


def naz_ui_oracle(O, d, implicit, which, sl, sls, solver, nthreads=2):
    """The restatement: the dense route on the zero-filled matrices (rows / columns of X)."""
    solver = dict(solver or {})
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver)
    cz = 0 if sv["use_cg"] else 1
    U = zero_filled(d["U_coo"], d["m"]) if "U" in which else None
    II = zero_filled(d["I_coo"], d["n"]) if "I" in which else None
    # ... except that a row with neither an entry of X nor of the side information is set to zero, not solved
    def neither(ix, coo, rows):
        has = np.zeros(rows, bool); has[ix] = True; has[coo[0]] = True
        return np.nonzero(~has)[0]
    O.set_zero_rows(neither(d["row"], d["U_coo"], d["m"]) if "U" in which else None,
                    neither(d["col"], d["I_coo"], d["n"]) if "I" in which else None)
    kw = dict(k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads, U=U, II=II, **sv,
              Cm=(d["C0"][:, d["ku"] - ku:] * cz).copy() if "U" in which else None,
              Dm=(d["D0"][:, d["ki"] - ki:] * cz).copy() if "I" in which else None)
    if implicit:
        r = O.fit_implicit_als_sideinfo(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, w_main=0.5, **kw)
        assert r["ret"] == 0
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"])
    r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                           scale_lam=sl, scale_lam_sideinfo=sls, **kw)
    assert r["ret"] == 0
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])


def naz_ui_hip(d, implicit, which, sl, sls, solver, dtype, flags=True, precompute=False):
    """The product: the estimators with SciPy sparse side information and NA_as_zero_user / NA_as_zero_item."""
    import scipy.sparse as sp
    from cmfrec_amd import CMF, CMF_implicit
    solver = dict(solver or {})
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    U = mk(d["U_coo"]) if "U" in which else None; I = mk(d["I_coo"]) if "I" in which else None
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver)
    common = dict(k=d["k"], k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, use_float=dtype is np.float32,
                  precompute_for_predictions=precompute, NA_as_zero_user=flags and "U" in which, NA_as_zero_item=flags and "I" in which, **sv)
    shape = (d["m"], d["n"])
    if implicit:
        mdl = CMF_implicit(lambda_=2.0, alpha=1.5, w_main=0.5, **common)
        mdl.fit((d["row"], d["col"], d["counts"]), U=U, I=I, shape=shape, A0=A0, B0=B0)
        out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_)
    else:
        mdl = CMF(lambda_=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **common)
        mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=I, shape=shape, A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
        out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, biasA=mdl.user_bias_, biasB=mdl.item_bias_, glob_mean=mdl.glob_mean_)
    if precompute:
        out["_model"] = mdl
    return out


def compare_fits(got, exp):
    """Largest relative error over the factor matrices / biases both sides carry."""
    err = 0.0
    for key in ("A", "B", "C", "D", "biasA", "biasB", "Ai", "Bi", "TransBtBinvBt", "BeTBeChol", "U_colmeans", "I_colmeans"):
        if key in exp and exp[key] is not None and got.get(key) is not None and np.size(exp[key]):
            e = maxrel(got[key], exp[key])
            err = max(err, e) if np.isfinite(e) else float("inf")        # NaN anywhere is a failure, never silently dropped
    if "glob_mean" in exp and "glob_mean" in got:
        err = max(err, abs(float(got["glob_mean"]) - float(exp["glob_mean"])))
    return err


# ---- non-negative factors (solve_nonneg instead of the Cholesky / CG solves) -----------------------------------------
def nonneg_problem(dtype, seed=51):
    rng = np.random.default_rng(seed)
    m, n, k = 140, 100, 8
    d = dict(m=m, n=n, k=k)
    lin = rng.choice(m * n, size=2200, replace=False)
    row = (lin // n).astype(np.int32); col = (lin % n).astype(np.int32)
    keep = ~np.isin(row, (4, 120))
    d["row"], d["col"] = row[keep], col[keep]
    d["ratings"] = (0.5 * rng.integers(1, 11, keep.sum())).astype(dtype)
    d["counts"] = np.ceil(rng.lognormal(1, 1, keep.sum())).astype(dtype)
    d["A0"] = np.abs(rng.standard_normal((m, k)) * 0.1).astype(dtype); d["B0"] = np.abs(rng.standard_normal((n, k)) * 0.1).astype(dtype)
    d["bA"] = (rng.standard_normal(m) * 0.1).astype(dtype); d["bB"] = (rng.standard_normal(n) * 0.1).astype(dtype)
    d["U"] = np.abs(rng.standard_normal((m, 5))).astype(dtype); d["I"] = np.abs(rng.standard_normal((n, 4))).astype(dtype)
    return d


# (name, implicit, side information, constructor / fit options)
NONNEG_CASES = [
    ("implicit", True, False, dict(nonneg=True, use_cg=True)),                       # the CG request is overridden
    ("explicit biases", False, False, dict(nonneg=True, scale_lam=True)),
    ("explicit side info, all constrained", False, True, dict(nonneg=True, nonneg_C=True, nonneg_D=True, user_bias=False,
                                                             item_bias=False, center=False)),
    ("implicit side info, C only", True, True, dict(nonneg_C=True, use_cg=True)),      # implicit: any constraint switches the CG off
    ("explicit few sweeps", False, False, dict(nonneg=True, max_cd_steps=3, user_bias=False, item_bias=False, center=False)),
    # L1 penalty: solve_elasticnet (common.c:2228-2294), or solve_nonneg with the penalty on the right-hand side.
    # (Not pinned: the implicit model WITHOUT side information and without nonneg -- factors_implicit_chol hands
    #  solve_elasticnet a matrix whose lower triangle was never filled, common.c:2106-2115, fill_lower = false.)
    ("explicit l1, biases, scale_lam", False, False, dict(l1_lam=0.02, scale_lam=True)),
    ("explicit l1 + nonneg, side info", False, True, dict(l1_lam=0.05, nonneg=True, nonneg_C=True, user_bias=False, item_bias=False,
                                                         center=False)),
    ("implicit l1, side info", True, True, dict(l1_lam=0.3)),
    ("explicit l1, side info, both scalings", False, True, dict(l1_lam=0.01, scale_lam=True, scale_lam_sideinfo=True)),
    ("implicit l1 + nonneg", True, False, dict(l1_lam=0.2, nonneg=True)),
]


def nonneg_reference(R, d, implicit, side, opts, nthreads=2):
    o = dict(opts)
    A0, B0 = d["A0"].copy(), d["B0"].copy()
    U, II = (d["U"], d["I"]) if side else (None, None)
    if implicit:
        r = R.fit_collective_implicit_als(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, niter=3,
                                          U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                          use_cg=o.pop("use_cg", False), **o)
        r = r if isinstance(r, dict) else dict(A=A0, B=B0, C=None, D=None)
        return dict(A=A0, B=B0, C=r.get("C"), D=r.get("D"))
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                      use_cg=o.pop("use_cg", False), finalize_chol=False, **o)
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])


def nonneg_oracle(O, d, implicit, side, opts, nthreads=2):
    o = dict(opts)
    O.set_nonneg(o.pop("nonneg", False), o.pop("nonneg_C", False), o.pop("nonneg_D", False), o.get("max_cd_steps", 100))
    O.set_l1(o.pop("l1_lam", 0.0), o.pop("max_cd_steps", 100))
    try:
        A0, B0 = d["A0"].copy(), d["B0"].copy()
        U, II = (d["U"], d["I"]) if side else (None, None)
        if implicit:
            r = O.fit_implicit_als_sideinfo(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, niter=3, U=U, II=II,
                                            w_user=2.0, w_item=0.5, nthreads=nthreads, use_cg=o.pop("use_cg", False), **o)
            return dict(A=A0, B=B0, C=r.get("C"), D=r.get("D"))
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=3, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads, use_cg=o.pop("use_cg", False),
                               finalize_chol=False, **o)
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])
    finally:
        O.set_nonneg(False, False, False, 100)
        O.set_l1(0.0, 100)


def nonneg_hip(d, implicit, side, opts, dtype):
    from cmfrec_amd import CMF, CMF_implicit
    o = dict(opts)
    if "l1_lam" in o:
        o["l1_lambda"] = o.pop("l1_lam")
    U, II = (d["U"], d["I"]) if side else (None, None)
    common = dict(k=d["k"], niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False)
    shape = (d["m"], d["n"])
    if implicit:
        mdl = CMF_implicit(lambda_=2.0, alpha=1.5, use_cg=o.pop("use_cg", False), **common, **o)
        mdl.fit((d["row"], d["col"], d["counts"]), U=U, I=II, shape=shape, A0=d["A0"], B0=d["B0"])
        return dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_)
    mdl = CMF(lambda_=0.3, use_cg=o.pop("use_cg", False), finalize_chol=False, **common, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=shape, A0=d["A0"], B0=d["B0"], biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- implicit features of the explicit model (add_implicit_features, collective.c:8448-8534) -------------------------
# (name, side information, options)
IMPLICIT_FEATS_CASES = [
    ("plain", False, dict()),
    ("scale_lam k_main", False, dict(scale_lam=True, k_main=2, w_implicit=0.7)),
    ("side info", True, dict(k_user=2, k_item=1, k_main=1, scale_lam_sideinfo=True, w_implicit=1.5)),
    ("no biases w_main", False, dict(user_bias=False, item_bias=False, center=False, w_main=2.0)),
    ("user bias only", True, dict(item_bias=False, w_implicit=0.25)),
    # block CG with the implicit-features term (collective.c:2301-2304, :2624-2643, :2862-2868); Ai / Bi stay closed-form
    ("cg", False, dict(use_cg=True, k_main=1, w_implicit=0.6)),
    ("cg side info finalize", True, dict(use_cg=True, finalize_chol=True, k_user=1, k_item=1, scale_lam=True)),
    ("pcg side info", True, dict(use_cg=True, precondition_cg=True, w_implicit=1.3, user_bias=False)),
    # not pinned: nonneg / L1 together with implicit features -- the reference segfaults on them (verified here with
    # nonneg=True and with l1_lam=0.05, one and two threads), so the product rejects the combination
]


def _impf_start(d, o):
    """Start values wide enough for the case's k_user / k_item / k_main (deterministic extension of the problem's A0, B0)."""
    rng = np.random.default_rng(77)
    dt = d["A0"].dtype
    ku, ki, km = o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
    ext = lambda M, left, right: np.ascontiguousarray(np.hstack(
        [np.abs(rng.standard_normal((M.shape[0], left)) * 0.1).astype(dt), M,
         np.abs(rng.standard_normal((M.shape[0], right)) * 0.1).astype(dt)]))
    return ext(d["A0"], ku, km), ext(d["B0"], ki, km)


def implicit_feats_reference(R, d, side, opts, nthreads=2):
    o = dict(opts); niter = o.pop("niter", 3)
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=niter, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      add_implicit_features=True, **o)
    assert r["ret"] == 0
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"],
                Ai=r["Ai"], Bi=r["Bi"])


def implicit_feats_oracle(O, d, side, opts, nthreads=2):
    o = dict(opts)
    niter = o.pop("niter", 3)
    O.set_nonneg(o.pop("nonneg", False), False, False, 100)
    O.set_l1(o.pop("l1_lam", 0.0) / o.get("w_main", 1.0), 100)
    try:
        A0, B0 = _impf_start(d, o)
        U, II = (d["U"], d["I"]) if side else (None, None)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=niter, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads, use_cg=o.pop("use_cg", False),
                               finalize_chol=o.pop("finalize_chol", False), add_implicit_features=True, **o)
        assert r["ret"] == 0
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"],
                    Ai=r["Ai"], Bi=r["Bi"])
    finally:
        O.set_nonneg(False, False, False, 100)
        O.set_l1(0.0, 100)


def implicit_feats_hip(d, side, opts, dtype):
    from cmfrec_amd import CMF
    o = dict(opts)
    niter = o.pop("niter", 3)
    o.setdefault("w_implicit", 1.0)           # the estimator's default is 0.5 (cmfrec/__init__.py), the C default used above 1
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    mdl = CMF(k=d["k"], niter=niter, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, lambda_=0.3, use_cg=o.pop("use_cg", False),
              finalize_chol=o.pop("finalize_chol", False), add_implicit_features=True, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_, Ai=mdl.Ai_, Bi=mdl.Bi_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- per-matrix penalties (lam_unique / l1_lam_unique: user bias, item bias, A, B, C, D; collective.c:430) ------------
LAM6 = [0.7, 0.2, 0.4, 0.25, 1.5, 0.6]
L16 = [0.02, 0.0, 0.05, 0.03, 0.04, 0.01]
# (name, implicit, side information, options)
LAM_UNIQUE_CASES = [
    ("explicit chol", False, False, dict(lam_unique=LAM6, use_cg=False)),
    ("explicit cg finalize", False, False, dict(lam_unique=LAM6, use_cg=True, finalize_chol=True, scale_lam=True)),
    ("explicit side info", False, True, dict(lam_unique=LAM6, use_cg=False, k_user=1, k_item=2, w_main=2.0)),
    ("explicit side info cg", False, True, dict(lam_unique=LAM6, use_cg=True, item_bias=False, scale_lam_sideinfo=True)),
    ("explicit l1 per matrix", False, True, dict(lam_unique=LAM6, l1_lam_unique=L16)),
    ("explicit implicit features", False, False, dict(lam_unique=LAM6, use_cg=False, add_implicit_features=True, w_implicit=0.8)),
    ("implicit cg", True, False, dict(lam_unique=LAM6, use_cg=True)),
    ("implicit side info chol", True, True, dict(lam_unique=LAM6, use_cg=False, k_user=1, w_main=0.5)),
    ("implicit side info l1 + nonneg", True, True, dict(lam_unique=LAM6, l1_lam_unique=L16, nonneg=True)),
    # the matrices for predictions: lam_unique[2] on the diagonals, the bias' lam_unique[0] as a correction of the last
    # entry (collective.c:9066-9074, :9225-9238)
    ("explicit precompute", False, True, dict(lam_unique=LAM6, use_cg=False, scale_lam=True, precompute=True)),
]


def lam_unique_reference(R, d, implicit, side, opts, nthreads=2):
    o = dict(opts)
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    if implicit:
        r = R.fit_collective_implicit_als(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, niter=3,
                                          U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                          use_cg=o.pop("use_cg", False), **o)
        r = r if isinstance(r, dict) else dict(C=None, D=None)
        return dict(A=A0, B=B0, C=r.get("C"), D=r.get("D"))
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])
    if r["Ai"] is not None: out.update(Ai=r["Ai"], Bi=r["Bi"])
    if r.get("pre"): out.update(TransBtBinvBt=r["pre"]["TransBtBinvBt"], BeTBeChol=np.triu(r["pre"]["BeTBeChol"]))
    return out


def lam_unique_oracle(O, d, implicit, side, opts, nthreads=2):
    o = dict(opts)
    wm = o.get("w_main", 1.0); o.pop("precompute", None)
    lam6 = np.asarray(o.pop("lam_unique"), np.float64); l16 = o.pop("l1_lam_unique", None)
    O.set_nonneg(o.pop("nonneg", False), False, False, 100)
    try:
        A0, B0 = _impf_start(d, o)
        U, II = (d["U"], d["I"]) if side else (None, None)
        if implicit:          # the implicit driver rescales by w_main itself
            O.set_lam_unique(lam6, l16)
            r = O.fit_implicit_als_sideinfo(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, niter=3, U=U, II=II,
                                            w_user=2.0, w_item=0.5, nthreads=nthreads, use_cg=o.pop("use_cg", False), **o)
            return dict(A=A0, B=B0, C=r.get("C"), D=r.get("D"))
        O.set_lam_unique(lam6 / wm, None if l16 is None else np.asarray(l16, np.float64) / wm)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=3, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads, use_cg=o.pop("use_cg", False),
                               finalize_chol=o.pop("finalize_chol", False), **o)
        assert r["ret"] == 0
        out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"])
        if r["Ai"] is not None: out.update(Ai=r["Ai"], Bi=r["Bi"])
        return out
    finally:
        O.set_nonneg(False, False, False, 100)
        O.set_lam_unique(None, None)


def lam_unique_hip(d, implicit, side, opts, dtype):
    from cmfrec_amd import CMF, CMF_implicit
    o = dict(opts)
    o["lambda_"] = o.pop("lam_unique")
    if "l1_lam_unique" in o:
        o["l1_lambda"] = o.pop("l1_lam_unique")
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    common = dict(k=d["k"], niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32,
                  precompute_for_predictions=o.pop("precompute", False))
    shape = (d["m"], d["n"])
    if implicit:
        mdl = CMF_implicit(alpha=1.5, use_cg=o.pop("use_cg", False), **common, **o)
        mdl.fit((d["row"], d["col"], d["counts"]), U=U, I=II, shape=shape, A0=A0, B0=B0)
        return dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_)
    mdl = CMF(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **common, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=shape, A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    if mdl.add_implicit_features: out.update(Ai=mdl.Ai_, Bi=mdl.Bi_)
    if mdl.precompute_for_predictions: out.update(TransBtBinvBt=mdl._TransBtBinvBt, BeTBeChol=np.triu(mdl._BeTBeChol))
    return out


# ---- dense side information with missing values (NaN): present entries only, centred column-wise ----------------------
def nan_side_problem(dtype, seed=61):
    d = sparse_sideinfo_problem(dtype, seed)
    rng = np.random.default_rng(seed + 1)
    mk = lambda c: (lambda M: (M.__setitem__((c[0], c[1]), c[2]), M)[1])(np.full((c[3], c[4]), np.nan, dtype))
    d["U_nan"], d["I_nan"] = mk(d["U_coo"]), mk(d["I_coo"])            # dense matrices, NaN where the sparse ones have nothing
    # nearly complete matrices (the reference's near_dense branches): 4 % missing
    few = lambda rows, cols: (lambda M: (M.__setitem__(rng.random((rows, cols)) < 0.04, np.nan), M)[1])(
        rng.standard_normal((rows, cols)).astype(dtype))
    d["U_few"], d["I_few"] = few(*d["U_nan"].shape), few(*d["I_nan"].shape)
    # (round 5) attributes of both kinds in one matrix -- some miss a handful of values, some most of them, one is complete -- and
    # matrices with at least 75 % complete attributes (optimizeA Case 1 with its fix-up loop, common.c:2905-2985)
    def mixed(rows, cols, complete):
        M = rng.standard_normal((rows, cols)).astype(dtype)
        for c in range(cols):
            if c < complete: continue
            frac = 0.6 if (c % 2) else 0.05
            M[rng.random(rows) < frac, c] = np.nan
        return M
    d["U_mixed"], d["I_mixed"] = mixed(*d["U_nan"].shape, 1), mixed(*d["I_nan"].shape, 1)
    d["U_near"], d["I_near"] = mixed(*d["U_nan"].shape, 7), mixed(*d["I_nan"].shape, 6)      # 7 of 9 / 6 of 7 attributes complete
    return d


def _nan_mats(d, solver):
    solver = dict(solver or {})
    few = solver.pop("few", False)
    variant = solver.pop("variant", "few" if few else "nan")
    return (d["U_" + variant], d["I_" + variant]), (solver or None)


def centred_coo(M):
    """(row, col, centred value, rows, cols) of the present entries + the column means (center_by_cols, common.c:4938-4997)."""
    means = np.nanmean(M.astype(np.float64), axis=0).astype(M.dtype)
    r, c = np.nonzero(~np.isnan(M))
    return (r.astype(np.int32), c.astype(np.int32), (M[r, c] - means[c]).astype(M.dtype), M.shape[0], M.shape[1]), means


# Cholesky solver, unscaled lambda: there the reference's shortcut for rows with few missing values (a precomputed Gramian
# minus the missing rows, common.c:762-790) solves the same system.  Under scale_lam that Gramian already carries
# lam x (all rows), and under CG such attributes are solved in closed form (or from zero with k steps) -- the per-attribute rules
# of nan_side_rules below (round 5; without them the nearly complete matrices come out 3e-2 .. 1e-1 apart).
NAN_SIDE_CASES = [("implicit UI", True, "UI", False, False, None), ("explicit UI", False, "UI", False, False, None),
                  ("explicit U", False, "U", False, False, None), ("implicit I", True, "I", False, False, None),
                  ("explicit UI nearly complete", False, "UI", False, False, dict(few=True)),
                  ("implicit UI nearly complete", True, "UI", False, False, dict(few=True)),
                  # round 5: scale_lam and the CG solvers (per-attribute rules of the dense C / D update, nan_side_rules)
                  ("explicit UI nearly complete, scale_lam", False, "UI", True, False, dict(few=True)),
                  ("explicit UI nearly complete, scale_lam_sideinfo", False, "UI", True, True, dict(few=True)),
                  ("explicit UI nearly complete, cg", False, "UI", False, False, dict(few=True, use_cg=True)),
                  ("implicit UI nearly complete, cg + finalize", True, "UI", False, False, dict(few=True, use_cg=True, finalize_chol=True)),
                  ("explicit UI mixed attributes, scale_lam, cg", False, "UI", True, False, dict(variant="mixed", use_cg=True)),
                  ("explicit U mixed attributes, scale_lam_sideinfo", False, "U", True, True, dict(variant="mixed")),
                  ("implicit UI mixed attributes, cg", True, "UI", False, False, dict(variant="mixed", use_cg=True)),
                  ("explicit UI 75 % complete, scale_lam, cg", False, "UI", True, False, dict(variant="near", use_cg=True)),
                  ("explicit I 75 % complete, scale_lam", False, "I", True, False, dict(variant="near")),
                  ("implicit UI 75 % complete, cg", True, "UI", False, False, dict(variant="near", use_cg=True)),
                  ("explicit UI sparse-like, scale_lam_sideinfo, cg + finalize", False, "UI", True, True, dict(use_cg=True, finalize_chol=True))]


def nan_side_reference(R, d, implicit, which, sl, sls, nthreads=2, solver=None):
    """The real reference on DENSE U / I with NaN."""
    (Un, In), solver = _nan_mats(d, solver)
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver or {})
    cz = 0 if sv["use_cg"] else 1
    kw = dict(k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3, nthreads=nthreads, **sv,
              U=Un if "U" in which else None, II=In if "I" in which else None,
              Cm=(d["C0"][:, d["ku"] - ku:] * cz).copy() if "U" in which else None,
              Dm=(d["D0"][:, d["ki"] - ki:] * cz).copy() if "I" in which else None)
    if implicit:
        r = R.fit_collective_implicit_als(A0, B0, d["row"], d["col"], d["counts"], d["k"], lam=2.0, alpha=1.5, w_main=0.5, **kw)
        return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], U_colmeans=r["U_colmeans"], I_colmeans=r["I_colmeans"])
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **kw)
    return dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"],
                U_colmeans=r["U_colmeans"], I_colmeans=r["I_colmeans"])


def nan_side_rules(M, kc, scale_lam):
    """What the dense C / D update does differently from the sparse one, per attribute (column of M [rows, p], NaN = missing;
    optimizeA Cases 1-2 on the transposed matrix, common.c:2793-3116): (closed-form mask, lambda multipliers or None).  An
    attribute that misses fewer than 2 kc values is solved from the precomputed Gramian minus the missing rows -- in closed form
    whatever use_cg says (mask 1), under scale_lam with the `rows` x lam of a complete one (:759-790, :3031-3032); when at least
    75 % of the attributes are complete (helpers.c:151-250) those share one factorisation (mask 1) and the ones that miss 2 kc
    values or more are redone by CG from zero with kc steps (mask 2; :2953-2985)."""
    rows, p = M.shape
    na = np.isnan(M).sum(0)
    near = (p - int((na > 0).sum())) >= int(0.75 * p)
    cf = (na < 2 * kc).astype(np.uint8)
    if near: cf[na >= 2 * kc] = 2
    mult = None
    if scale_lam and ((na > 0) & (na < 2 * kc)).any():
        mult = np.where(na < 2 * kc, rows, np.where(na < rows, rows - na, 1)).astype(M.dtype)
    return cf, mult


def nan_side_oracle(O, d, implicit, which, sl, sls, nthreads=2, solver=None):
    """The oracle's sparse-side-information fit on the centred present entries, with the per-attribute rules of the dense C / D
    update (nan_side_rules)."""
    d2 = dict(d)
    (Un, In), solver = _nan_mats(d, solver)
    d2["U_coo"], _ = centred_coo(Un); d2["I_coo"], _ = centred_coo(In)
    scaled = (sl or sls) and not implicit
    cfC, multC = nan_side_rules(Un, d["ku"] + d["k"], scaled) if "U" in which else (None, None)
    cfD, multD = nan_side_rules(In, d["ki"] + d["k"], scaled) if "I" in which else (None, None)
    O.set_sideinfo_dense_rules(cfC, multC, cfD, multD)
    try:
        return sparse_sideinfo_oracle(O, d2, implicit, which, sl, sls, nthreads=nthreads, solver=solver)
    finally:
        O.set_sideinfo_dense_rules()


def nan_side_hip(d, implicit, which, sl, sls, dtype, solver=None):
    """The product: the estimators with dense U / I that contain NaN."""
    from cmfrec_amd import CMF, CMF_implicit
    (Un, In), solver = _nan_mats(d, solver)
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    U = Un if "U" in which else None; I = In if "I" in which else None
    sv = dict(use_cg=False, finalize_chol=False); sv.update(solver or {})
    common = dict(k=d["k"], k_main=d["km"], k_user=ku, k_item=ki, w_user=3.0, w_item=0.7, niter=3,
                  use_float=dtype is np.float32, precompute_for_predictions=False, **sv)
    shape = (d["m"], d["n"])
    if implicit:
        mdl = CMF_implicit(lambda_=2.0, alpha=1.5, w_main=0.5, **common)
        mdl.fit((d["row"], d["col"], d["counts"]), U=U, I=I, shape=shape, A0=A0, B0=B0)
        out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_)
    else:
        mdl = CMF(lambda_=0.3, scale_lam=sl, scale_lam_sideinfo=sls, **common)
        mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=I, shape=shape, A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
        out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, biasA=mdl.user_bias_, biasB=mdl.item_bias_, glob_mean=mdl.glob_mean_)
    if U is not None: out["U_colmeans"] = mdl._U_colmeans
    if I is not None: out["I_colmeans"] = mdl._I_colmeans
    return out


# ---- observation weights of the explicit model (fit(..., W=...); fit_collective_explicit_als with weight != NULL) ------------
def weights_problem(dtype, seed=81):
    """A ratings problem with one weight per entry.  The entries are ordered by column: the reference hands its B-step the
    weights in COO order where the CSC order is meant (collective.c:8642, :8689 pass `weight`, not `weightC`, for sparse X,
    while its wsumB and its bias start values read weightC), so only for input sorted by column does it compute what it
    documents -- the ordering its own docs ask for (cmfrec/__init__.py:3095-3099).  The oracle and the HIP path use the
    CSC-ordered weights throughout."""
    d = nonneg_problem(dtype, seed)
    rng = np.random.default_rng(seed + 1)
    # one long row and one long column (more than a 64-entry tile; the split-row path has its own test)
    extra_c = rng.choice(d["n"], 70, replace=False); extra_r = rng.choice(d["m"], 90, replace=False)
    extra_r = extra_r[~np.isin(extra_r, (4, 120))]
    row = np.concatenate([d["row"], np.full(len(extra_c), 9, np.int32), extra_r.astype(np.int32)])
    col = np.concatenate([d["col"], extra_c.astype(np.int32), np.full(len(extra_r), 11, np.int32)])
    lin = row.astype(np.int64) * d["n"] + col
    _, first = np.unique(lin, return_index=True)
    first.sort()
    row, col = row[first], col[first]
    o = np.argsort(col, kind="stable")
    d["row"], d["col"] = row[o], col[o]
    d["ratings"] = (0.5 * rng.integers(1, 11, len(o))).astype(dtype)
    d["W"] = (0.2 + 2.5 * rng.random(len(o)) ** 2).astype(dtype)
    d.pop("counts")
    d["A0"] = (rng.standard_normal((d["m"], d["k"])) * 0.1).astype(dtype); d["B0"] = (rng.standard_normal((d["n"], d["k"])) * 0.1).astype(dtype)
    d["U"] = rng.standard_normal((d["m"], 5)).astype(dtype); d["I"] = rng.standard_normal((d["n"], 4)).astype(dtype)
    return d


# (name, side information, options).  seed: the reference's own random start + its weighted bias start values (reset_values).
WEIGHT_CASES = [
    ("cg", False, dict(use_cg=True, finalize_chol=False)),
    ("cg scale_lam finalize", False, dict(use_cg=True, finalize_chol=True, scale_lam=True)),
    ("pcg scale_lam", False, dict(use_cg=True, precondition_cg=True, finalize_chol=False, scale_lam=True)),
    ("chol", False, dict(use_cg=False)),
    ("chol scale_lam no bias", False, dict(use_cg=False, scale_lam=True, user_bias=False, item_bias=False)),
    ("cg seeded, both biases", False, dict(use_cg=True, finalize_chol=False, scale_lam=True, seed=5)),
    ("chol seeded, both biases", False, dict(use_cg=False, seed=6)),
    ("cg seeded, user bias", False, dict(use_cg=True, finalize_chol=False, item_bias=False, scale_lam=True, seed=7)),
    ("cg seeded, item bias", False, dict(use_cg=True, finalize_chol=False, user_bias=False, seed=8)),
    # side information + weights, closed form: without centring.  The reference's collective closed form subtracts
    # (w - 1) x glob_mean from every weighted entry of the right-hand side whether or not X is missing-as-zero
    # (collective.c:1744-1753 lacks the NA_as_zero_X condition its twin common.c:826-845 has), so with centring its Cholesky
    # and its CG solvers minimise different objectives; tests/test_oracle_vs_ref.py::test_reference_weight_defects shows it.
    ("side info chol", True, dict(use_cg=False, k_user=1, k_item=2, center=False)),
    ("side info chol scale_lam", True, dict(use_cg=False, scale_lam=True, k_main=1, center=False)),
    ("side info cg", True, dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("side info pcg finalize", True, dict(use_cg=True, precondition_cg=True, finalize_chol=True, center=False)),
    ("nonneg", False, dict(nonneg=True, scale_lam=True, use_cg=False)),
    ("l1", False, dict(l1_lam=0.02, use_cg=False)),
]


def weights_reference(R, d, side, opts, nthreads=2):
    o = dict(opts)
    seed = o.pop("seed", None)
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    if seed is not None:
        A0[:] = 0; B0[:] = 0
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], lam=0.3, niter=3, U=U, II=II, w_user=2.0,
                                          w_item=0.5, nthreads=nthreads, weight=d["W"], reset_values=True, seed=seed, **kw)
    else:
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                          lam=0.3, niter=3, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"], **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def weights_oracle(O, d, side, opts, nthreads=2):
    """None where the oracle has no restatement (side information, a one-sided or seeded start, nonneg / L1 with weights)."""
    o = dict(opts)
    if side or "seed" in o or o.get("nonneg") or o.get("l1_lam"):
        return None
    A0, B0 = _impf_start(d, o)
    r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                           niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"], use_cg=o.pop("use_cg", False),
                           finalize_chol=o.pop("finalize_chol", False), **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def weights_hip(d, side, opts, dtype, weights=True, nthreads=1):
    from cmfrec_amd import CMF
    o = dict(opts)
    seed = o.pop("seed", None)
    if "l1_lam" in o:
        o["l1_lambda"] = o.pop("l1_lam")
    A0, B0 = _impf_start(d, o)
    U, II = (d["U"], d["I"]) if side else (None, None)
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=nthreads,
              **(dict(random_state=seed) if seed is not None else {}), **o)
    start = {} if seed is not None else dict(A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"] if weights else None, **start)
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- observation weights together with SPARSE side information (round 6): collective_closed_form_block's weight branches with
#      u_vec_sp (collective.c:1636-1653, :1673-1699, :1738-1753) and collective_block_cg's (:2187-2208, :2292-2298) -- fixture g31
def weights_sparse_side_problem(dtype, seed=83):
    d = weights_problem(dtype, seed)
    rng = np.random.default_rng(seed + 7)
    m, n, p, q = d["m"], d["n"], 7, 6
    def coo(rows, cols, cnt, empty):
        lin = rng.choice(rows * cols, size=cnt, replace=False)
        r = (lin // cols).astype(np.int32); c = (lin % cols).astype(np.int32)
        keep = ~np.isin(r, empty)
        return r[keep], c[keep]
    ur, uc = coo(m, p, 3 * m, (2, 9)); ir, ic = coo(n - 3, q, 2 * n, (11,))
    d["U_coo"] = (ur, uc, rng.standard_normal(len(ur)).astype(dtype), m, p)
    d["I_coo"] = (ir, ic, rng.standard_normal(len(ir)).astype(dtype), n - 3, q)
    d["p"], d["q"] = p, q
    return d


# (name, sides, options); closed form without centring for the reason given at WEIGHT_CASES
WEIGHT_SPARSE_SIDE_CASES = [
    ("chol", "UI", dict(use_cg=False, center=False, k_user=1, k_item=2)),
    ("chol scale_lam", "UI", dict(use_cg=False, scale_lam=True, center=False, k_main=1)),
    ("chol U, no biases", "U", dict(use_cg=False, center=False, user_bias=False, item_bias=False)),
    ("cg scale_lam", "UI", dict(use_cg=True, finalize_chol=False, scale_lam=True, k_user=1)),
    ("cg I", "I", dict(use_cg=True, finalize_chol=False, k_item=1, k_main=1)),
    ("pcg finalize", "UI", dict(use_cg=True, precondition_cg=True, finalize_chol=True, center=False)),
]


def weights_sparse_side_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which: o["k_user"] = 0
    if "I" not in which: o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    ku, ki = o.get("k_user", 0), o.get("k_item", 0)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"],
                                      U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None,
                                      Cm=np.zeros((d["p"], ku + d["k"]), d["A0"].dtype) if "U" in which else None,
                                      Dm=np.zeros((d["q"], ki + d["k"]), d["A0"].dtype) if "I" in which else None, **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def weights_sparse_side_hip(d, which, opts, dtype, weights=True):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which: o["k_user"] = 0
    if "I" not in which: o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    U = mk(d["U_coo"]) if "U" in which else None; II = mk(d["I_coo"]) if "I" in which else None
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"] if weights else None,
            A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- implicit features together with SPARSE side information (round 6): collective_closed_form_block with u_vec_sp and
#      add_implicit_features (collective.c:1636-1653 beside :1704-1707, :1757-1771), collective_block_cg (:2292-2304) -- fixture g32
IMPF_SPARSE_SIDE_CASES = [
    ("chol UI", "UI", dict(k_user=2, k_item=1, k_main=1, w_implicit=1.5)),
    ("chol scaled U", "U", dict(scale_lam=True, scale_lam_sideinfo=True, w_implicit=0.7)),
    ("chol no biases", "UI", dict(user_bias=False, item_bias=False, center=False)),
    ("cg UI", "UI", dict(use_cg=True, k_user=1, k_main=1, w_implicit=0.6)),
    ("cg finalize I", "I", dict(use_cg=True, finalize_chol=True, k_item=1, scale_lam=True)),
    ("pcg UI", "UI", dict(use_cg=True, precondition_cg=True, w_implicit=1.3, user_bias=False)),
    # DENSE side information that covers fewer rows than X ("u" / "i": the first m - 4 users / n - 3 items): with implicit features
    # the reference solves the remaining rows on the whole vector, not on the X block only (collective.c:4906-4960)
    ("cg dense short U", "u", dict(use_cg=True, k_user=2, w_implicit=0.8)),
    ("cg dense short UI finalize", "ui", dict(use_cg=True, finalize_chol=True, k_user=1, k_item=2, scale_lam=True)),
    ("chol dense short I", "i", dict(k_item=1, k_main=1)),
]


def _impf_sides(d, which):
    """(dense U, dense I, sparse U, sparse I) of a case: capitals = the sparse triplets, small letters = dense, fewer rows than X."""
    return (d["U"][:d["m"] - 4] if "u" in which else None, d["I"][:d["n"] - 3] if "i" in which else None,
            d["U_coo"] if "U" in which else None, d["I_coo"] if "I" in which else None)


def impf_sparse_side_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    ku, ki = o.get("k_user", 0), o.get("k_item", 0)
    Ud, Id, Us, Is = _impf_sides(d, which)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads,
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      U=Ud, II=Id, U_coo=Us, I_coo=Is,
                                      Cm=np.zeros((d["p"], ku + d["k"]), d["A0"].dtype) if Us is not None else None,
                                      Dm=np.zeros((d["q"], ki + d["k"]), d["A0"].dtype) if Is is not None else None,
                                      add_implicit_features=True, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], glob_mean=r["glob_mean"], Ai=r["Ai"], Bi=r["Bi"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def impf_sparse_side_hip(d, which, opts, dtype):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    o.setdefault("w_implicit", 1.0)
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    Ud, Id, Us, Is = _impf_sides(d, which)
    U = mk(Us) if Us is not None else Ud; II = mk(Is) if Is is not None else Id
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), add_implicit_features=True, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_, Ai=mdl.Ai_, Bi=mdl.Bi_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero_X with observation weights AND sparse side information (round 6; fixture g36): the weight branches of
#      collective_closed_form_block / collective_block_cg with the row's attributes as a sparse vector; the problem of g31 (entries
#      ordered by column), side information on exactly the rows / columns of X
NAZ_WEIGHTED_SPARSE_SIDE_CASES = [
    ("chol, both sides", "UI", dict(use_cg=False)),
    # (scale_lam_sideinfo is refused with weights + sparse side information: the reference's multipliers there add
    #  `U_csr_p[row+1] - U_csr[row]`, a value of U in place of its row pointer, collective.c:8087, :8106)
    ("chol, scale_lam, k_user / k_item", "UI", dict(use_cg=False, scale_lam=True, k_user=1, k_item=2)),
    ("chol, user side, no biases, no centring", "U", dict(use_cg=False, user_bias=False, item_bias=False, center=False, k_main=1)),
    ("cg, both sides", "UI", dict(use_cg=True, finalize_chol=False)),
    ("pcg, item side, item bias", "I", dict(use_cg=True, precondition_cg=True, finalize_chol=False, user_bias=False, k_item=1)),
    ("cg + finalize, scale_lam", "UI", dict(use_cg=True, finalize_chol=True, scale_lam=True)),
]


def _nwss_sides(d, which):
    Uc, Ic = d["U_coo"], d["I_coo"]
    return ((Uc[0], Uc[1], Uc[2], d["m"], d["p"]) if "U" in which else None, (Ic[0], Ic[1], Ic[2], d["n"], d["q"]) if "I" in which else None)


def naz_weighted_sparse_side_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which: o["k_user"] = 0
    if "I" not in which: o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    Us, Is = _nwss_sides(d, which)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"], NA_as_zero_X=True,
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      U_coo=Us, I_coo=Is, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_weighted_sparse_side_hip(d, which, opts, dtype):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which: o["k_user"] = 0
    if "I" not in which: o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    Us, Is = _nwss_sides(d, which)
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              NA_as_zero=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=mk(Us) if Us is not None else None, I=mk(Is) if Is is not None else None,
            shape=(d["m"], d["n"]), W=d["W"], A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero_X with implicit features AND side information (round 6; fixture g37): dense side information -- the shared block
#      matrix with w_i Bi^T Bi on its X block, whatever the solver asked for (collective.c:5121-5230) --, sparse -- row by row, closed form
#      or block CG (:1534-1846, :2134-2903)
NAZ_IMPF_SIDE_CASES = [
    # (name, "dense" / "sparse", sides, options)
    ("dense UI", "dense", "UI", dict()),
    ("dense UI, cg asked for, scaled", "dense", "UI", dict(use_cg=True, finalize_chol=False, scale_lam=True, scale_lam_sideinfo=True, w_implicit=0.6)),
    ("dense U, k_user, no centring, user bias", "dense", "U", dict(k_user=2, k_main=1, center=False, item_bias=False)),
    ("sparse UI, chol", "sparse", "UI", dict()),
    ("sparse I, chol, scale_lam, w_implicit", "sparse", "I", dict(scale_lam=True, w_implicit=1.4)),
    ("sparse UI, cg", "sparse", "UI", dict(use_cg=True, finalize_chol=False)),
    ("sparse U, pcg, no biases", "sparse", "U", dict(use_cg=True, precondition_cg=True, finalize_chol=False, user_bias=False, item_bias=False, center=False)),
]


def _naz_impf_side_inputs(kind, which, opts, dtype):
    """problem, start values, side-information keyword arguments of the reference call, and the estimator's U / I."""
    import scipy.sparse as sp
    o = dict(opts)
    if kind == "dense":
        d = naz_problem(dtype)
        A0, B0 = _impf_start(d, o)
        U, II = _naz_side(d, which)
        return d, o, A0, B0, dict(U=U, II=II), U, II
    d = naz_sparse_side_problem(dtype)
    ku = d["ku"] if "U" in which else 0; ki = d["ki"] if "I" in which else 0
    A0 = d["A0"][:, d["ku"] - ku:].copy(); B0 = d["B0"][:, d["ki"] - ki:].copy()
    o.update(k_main=d["km"], k_user=ku, k_item=ki)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    kw = dict(U_coo=d["U_coo"] if "U" in which else None, I_coo=d["I_coo"] if "I" in which else None)
    return d, o, A0, B0, kw, (mk(d["U_coo"]) if "U" in which else None), (mk(d["I_coo"]) if "I" in which else None)


def naz_impf_side_reference(R, kind, which, opts, dtype, nthreads=2):
    d, o, A0, B0, kw, _, _ = _naz_impf_side_inputs(kind, which, opts, dtype)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=3.0, w_item=0.7, nthreads=nthreads, NA_as_zero_X=True,
                                      add_implicit_features=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      **kw, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], Ai=r["Ai"], Bi=r["Bi"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_impf_side_hip(kind, which, opts, dtype):
    from cmfrec_amd import CMF
    d, o, A0, B0, _, U, II = _naz_impf_side_inputs(kind, which, opts, dtype)
    o.setdefault("w_implicit", 1.0)
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=3.0, w_item=0.7, use_float=dtype is np.float32, precompute_for_predictions=False,
              NA_as_zero=True, add_implicit_features=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
              nthreads=1, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, Ai=mdl.Ai_, Bi=mdl.Bi_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- observation weights together with implicit features (round 6; fixture g38): the weighted row solvers with the implicit-features
#      term (collective.c:1673-1699 beside :1704-1707, :1757-1771; block CG :2187-2208 beside :2301-2304, :2624-2643).  The problem of
#      g31 (entries ordered by column); closed form without centring for the reason given at WEIGHT_CASES
WEIGHT_IMPF_CASES = [
    # (name, side information: "" none / small letters dense / capitals sparse, options)
    ("chol", "", dict(use_cg=False, center=False)),
    ("chol, scale_lam, k_main", "", dict(use_cg=False, center=False, scale_lam=True, k_main=2, w_implicit=0.7)),
    ("cg", "", dict(use_cg=True, finalize_chol=False, w_implicit=0.6)),
    ("pcg, no biases", "", dict(use_cg=True, precondition_cg=True, finalize_chol=False, user_bias=False, item_bias=False)),
    ("dense side info, chol", "ui", dict(use_cg=False, center=False, k_user=1, k_item=2)),
    ("dense side info, cg + finalize", "ui", dict(use_cg=True, finalize_chol=True, scale_lam=True, center=False)),
    ("sparse side info, chol", "UI", dict(use_cg=False, center=False, k_item=1)),
    ("sparse side info, cg", "UI", dict(use_cg=True, finalize_chol=False, k_user=1, w_implicit=1.3)),
]


def _wimpf_sides(d, which):
    return (d["U"] if "u" in which else None, d["I"] if "i" in which else None,
            d["U_coo"] if "U" in which else None, d["I_coo"] if "I" in which else None)


def weights_impf_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    ku, ki = o.get("k_user", 0), o.get("k_item", 0)
    Ud, Id, Us, Is = _wimpf_sides(d, which)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"], add_implicit_features=True,
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      U=Ud, II=Id, U_coo=Us, I_coo=Is,
                                      Cm=np.zeros((d["p"], ku + d["k"]), d["A0"].dtype) if Us is not None else None,
                                      Dm=np.zeros((d["q"], ki + d["k"]), d["A0"].dtype) if Is is not None else None, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], Ai=r["Ai"], Bi=r["Bi"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def weights_impf_hip(d, which, opts, dtype, weights=True):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    o.setdefault("w_implicit", 1.0)
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    Ud, Id, Us, Is = _wimpf_sides(d, which)
    U = mk(Us) if Us is not None else Ud; II = mk(Is) if Is is not None else Id
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              add_implicit_features=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1, **o)
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"] if weights else None,
            A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, Ai=mdl.Ai_, Bi=mdl.Bi_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero_X with observation weights AND implicit features (round 6; fixture g39), without and with side information on exactly
#      the rows / columns of X: the reference's collective route row by row (collective.c:1534-1846 / :2134-2903 with weight, NA_as_zero_X
#      and add_implicit_features); a row without entries is zero unless the bias / mean constant exists (:1258-1268)
NAZ_WEIGHTED_IMPF_CASES = [
    # (name, side information: "" none / small letters dense / capitals sparse, options)
    ("chol", "", dict(use_cg=False)),
    ("chol, no biases, no centring, scale_lam", "", dict(use_cg=False, user_bias=False, item_bias=False, center=False, scale_lam=True, w_implicit=0.7)),
    ("cg", "", dict(use_cg=True, finalize_chol=False, k_main=1)),
    ("cg, no biases, no centring", "", dict(use_cg=True, finalize_chol=False, user_bias=False, item_bias=False, center=False)),
    ("pcg, user bias", "", dict(use_cg=True, precondition_cg=True, finalize_chol=False, item_bias=False, w_implicit=1.3)),
    ("dense side info, chol", "ui", dict(use_cg=False, k_user=1, k_item=1)),
    ("dense side info, cg", "ui", dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("sparse side info, chol", "UI", dict(use_cg=False, k_item=2)),
    ("sparse side info, cg + finalize", "UI", dict(use_cg=True, finalize_chol=True, k_user=1)),
]


def _nwi_sides(d, which):
    Uc, Ic = d["U_coo"], d["I_coo"]
    return (d["U"] if "u" in which else None, d["I"] if "i" in which else None,
            (Uc[0], Uc[1], Uc[2], d["m"], d["p"]) if "U" in which else None, (Ic[0], Ic[1], Ic[2], d["n"], d["q"]) if "I" in which else None)


def naz_weighted_impf_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    Ud, Id, Us, Is = _nwi_sides(d, which)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, w_user=2.0, w_item=0.5, nthreads=nthreads, weight=d["W"], NA_as_zero_X=True,
                                      add_implicit_features=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      U=Ud, II=Id, U_coo=Us, I_coo=Is, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], Ai=r["Ai"], Bi=r["Bi"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_weighted_impf_hip(d, which, opts, dtype):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    o.setdefault("w_implicit", 1.0)
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    Ud, Id, Us, Is = _nwi_sides(d, which)
    U = mk(Us) if Us is not None else Ud; II = mk(Is) if Is is not None else Id
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              NA_as_zero=True, add_implicit_features=True, use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
              nthreads=1, **o)
    start = dict(A0=A0, B0=B0)
    if mdl.user_bias or mdl.item_bias: start.update(biasA0=d["bA"], biasB0=d["bB"])
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"], **start)
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, Ai=mdl.Ai_, Bi=mdl.Bi_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- the global mean a caller with nthreads >= 8 receives (calc_mean_and_center, common.c:3496-3513 unweighted: sum / count;
#      :3561-3571 weighted: the UNWEIGHTED sum over the sum of the weights) -- fixture g23 -----------------------------------
# (name, weighted, options): centred fits through the 82-argument entry point with nthreads = 8, given start values and seeded
NTHREADS8_CASES = [
    ("cg, both biases", False, dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("chol, both biases", False, dict(use_cg=False)),
    ("cg seeded", False, dict(use_cg=True, finalize_chol=False, seed=9)),
    ("weighted cg, both biases", True, dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("weighted chol", True, dict(use_cg=False)),
    ("weighted cg seeded", True, dict(use_cg=True, finalize_chol=False, scale_lam=True, seed=10)),
]


def nthreads8_reference(R, d, weighted, opts):
    e = dict(d)
    if not weighted:
        e["W"] = None
    return weights_reference(R, e, False, opts, nthreads=8)


def nthreads8_oracle(O, d, weighted, opts):
    """None for the seeded starts (the oracle has no restatement of the weighted random start)."""
    if "seed" in opts:
        return None
    e = dict(d)
    if not weighted:
        e["W"] = None
    return weights_oracle(O, e, False, opts, nthreads=8)


def nthreads8_hip(d, weighted, opts, dtype, nthreads=8):
    return weights_hip(d, False, opts, dtype, weights=weighted, nthreads=nthreads)


# ---- NA_as_zero for the main matrix (CMF(NA_as_zero=True); fit_collective_explicit_als with NA_as_zero_X) -------------------
def naz_problem(dtype, seed=91):
    d = nonneg_problem(dtype, seed)
    rng = np.random.default_rng(seed + 1)
    keep = d["col"] != 7                                  # a column without entries next to the rows (4, 120) without
    d["row"], d["col"], d["ratings"] = d["row"][keep], d["col"][keep], d["ratings"][keep]
    d["A0"] = (rng.standard_normal((d["m"], d["k"])) * 0.1).astype(dtype); d["B0"] = (rng.standard_normal((d["n"], d["k"])) * 0.1).astype(dtype)
    return d


# (name, options).  seed: the reference's own random start + its missing-as-zero bias start values (m > n here, where the
# item sweep's average over the wrong bound stays inside the array)
NAZ_CASES = [
    ("chol, biases", dict(use_cg=False)),
    ("cg asked for, scale_lam", dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    ("no biases", dict(use_cg=False, user_bias=False, item_bias=False)),
    ("no centring, user bias, scale_lam", dict(use_cg=False, center=False, item_bias=False, scale_lam=True)),
    ("item bias, k_main", dict(use_cg=False, user_bias=False, k_main=2)),
    ("no biases, no centring", dict(use_cg=False, user_bias=False, item_bias=False, center=False)),
    ("per-matrix lambdas", dict(use_cg=False, lam_unique=LAM6)),
    ("seeded, both biases", dict(use_cg=False, scale_lam=True, seed=5)),
    ("seeded, user bias", dict(use_cg=False, item_bias=False, seed=6)),
    ("seeded, item bias (cg asked for)", dict(use_cg=True, finalize_chol=False, user_bias=False, scale_lam=True, seed=7)),
]


def naz_reference(R, d, opts, nthreads=2):
    o = dict(opts)
    seed = o.pop("seed", None)
    A0, B0 = _impf_start(d, o)
    kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    if seed is not None:
        A0[:] = 0; B0[:] = 0
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], lam=0.3, niter=3, nthreads=nthreads,
                                          NA_as_zero_X=True, reset_values=True, seed=seed, **kw)
    else:
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                          lam=0.3, niter=3, nthreads=nthreads, NA_as_zero_X=True, **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_oracle(O, d, opts, nthreads=2):
    """None for the seeded cases (the oracle has no random start)."""
    o = dict(opts)
    if "seed" in o:
        return None
    lam6 = o.pop("lam_unique", None)
    if lam6 is not None:
        O.set_lam_unique(np.asarray(lam6, np.float64), None)
    try:
        A0, B0 = _impf_start(d, o)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=3, nthreads=nthreads, NA_as_zero_X=True, use_cg=o.pop("use_cg", False),
                               finalize_chol=o.pop("finalize_chol", False), **o)
    finally:
        O.set_lam_unique(None, None)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_hip(d, opts, dtype, NA_as_zero=True):
    from cmfrec_amd import CMF
    o = dict(opts)
    seed = o.pop("seed", None)
    if "lam_unique" in o:
        o["lambda_"] = o.pop("lam_unique")
    else:
        o["lambda_"] = 0.3
    A0, B0 = _impf_start(d, o)
    mdl = CMF(k=d["k"], niter=3, use_float=dtype is np.float32, precompute_for_predictions=False, NA_as_zero=NA_as_zero,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1,
              **(dict(random_state=seed) if seed is not None else {}), **o)
    start = {} if seed is not None else dict(A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    mdl.fit((d["row"], d["col"], d["ratings"]), shape=(d["m"], d["n"]), **start)
    out = dict(A=mdl.A_, B=mdl.B_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero for the main matrix WITH observation weights (optimizeA Case 4, NA_as_zero && weight: common.c:3209-3302, per row
# :846-907 closed form, :1293-1441 CG) -- fixture g24.  Given start values only: the reference's bias start values are not defined
# for this combination (common.c:4727-4731 index the item biases by row).  Entries ordered by column, like every weighted case.
def naz_weighted_problem(dtype, seed=93):
    d = weights_problem(dtype, seed)
    keep = d["col"] != 7                                  # a column without entries next to the rows (4, 120) without
    for key in ("row", "col", "ratings", "W"):
        d[key] = d[key][keep]
    return d


NAZ_WEIGHTED_CASES = [
    ("chol, biases", dict(use_cg=False)),
    ("cg, biases", dict(use_cg=True, finalize_chol=False)),
    ("chol, scale_lam", dict(use_cg=False, scale_lam=True)),
    ("cg + finalize, scale_lam, item bias", dict(use_cg=True, finalize_chol=True, scale_lam=True, user_bias=False)),
    ("chol, no centring, user bias", dict(use_cg=False, center=False, item_bias=False)),
    ("cg, no biases, no centring: rows without entries stay", dict(use_cg=True, finalize_chol=False, user_bias=False, item_bias=False, center=False)),
    ("chol, no biases, centred, k_main", dict(use_cg=False, user_bias=False, item_bias=False, k_main=2)),
    ("cg, per-matrix lambdas", dict(use_cg=True, finalize_chol=False, lam_unique=LAM6)),
    # precondition_cg (round 6): factors_explicit_pcg_NA_as_zero_weighted, common.c:1443-1613
    ("pcg, biases", dict(use_cg=True, finalize_chol=False, precondition_cg=True)),
    ("pcg, scale_lam, item bias", dict(use_cg=True, finalize_chol=False, precondition_cg=True, scale_lam=True, user_bias=False)),
    ("pcg, no biases, no centring: rows without entries stay", dict(use_cg=True, finalize_chol=False, precondition_cg=True, user_bias=False, item_bias=False, center=False)),
    ("pcg + finalize, per-matrix lambdas, k_main", dict(use_cg=True, finalize_chol=True, precondition_cg=True, lam_unique=LAM6, k_main=2)),
]


def naz_weighted_reference(R, d, opts, nthreads=2):
    o = dict(opts)
    A0, B0 = _impf_start(d, o)
    kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                      lam=0.3, niter=3, nthreads=nthreads, NA_as_zero_X=True, weight=d["W"], **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_weighted_oracle(O, d, opts, nthreads=2):
    o = dict(opts)
    lam6 = o.pop("lam_unique", None)
    if lam6 is not None:
        O.set_lam_unique(np.asarray(lam6, np.float64), None)
    try:
        A0, B0 = _impf_start(d, o)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3,
                               niter=3, nthreads=nthreads, NA_as_zero_X=True, weight=d["W"], use_cg=o.pop("use_cg", False),
                               finalize_chol=o.pop("finalize_chol", False), **o)
    finally:
        O.set_lam_unique(None, None)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_weighted_hip(d, opts, dtype, weights=True, **fit_kw):
    from cmfrec_amd import CMF
    o = dict(opts)
    if "lam_unique" in o:
        o["lambda_"] = o.pop("lam_unique")
    else:
        o["lambda_"] = 0.3
    A0, B0 = _impf_start(d, o)
    mdl = CMF(k=d["k"], niter=3, use_float=dtype is np.float32, precompute_for_predictions=False, NA_as_zero=True,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1, **o)
    start = dict(A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    start.update(fit_kw)
    mdl.fit((d["row"], d["col"], d["ratings"]), shape=(d["m"], d["n"]), W=d["W"] if weights else None, **start)
    out = dict(A=mdl.A_, B=mdl.B_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- NA_as_zero for the main matrix together with dense side information (optimizeA_collective with the factorised shared
# block matrix, collective.c:5566-5968 / :5607-5617, :5700-5716) ---------------------------------------------------------------
# (name, which sides carry side information, options).  Side information on exactly the rows / columns of X (the reference's own
# build corrupts its heap with fewer).
NAZ_SIDE_CASES = [
    ("both sides, biases", "UI", dict()),
    ("both sides, scale_lam", "UI", dict(scale_lam=True)),
    ("both sides, scale_lam_sideinfo", "UI", dict(scale_lam_sideinfo=True)),
    ("k_user / k_item / k_main, weights of the sides, user bias", "UI", dict(center=False, item_bias=False, k_user=2, k_item=1, k_main=2,
                                                                               w_user=0.7, w_item=1.3)),
    ("no biases, no centring", "UI", dict(user_bias=False, item_bias=False, center=False)),
    ("user side only", "U", dict()),
    ("item side only, scale_lam", "I", dict(scale_lam=True)),
    ("per-matrix lambdas", "UI", dict(lam_unique=LAM6)),
    ("seeded, both biases", "UI", dict(scale_lam=True, seed=5)),
    # use_cg: the reference takes the factorised block matrix before it looks at the solver (collective.c:1364-1460): closed-form numbers
    ("cg, both sides, biases", "UI", dict(use_cg=True, finalize_chol=False)),
    ("cg + finalize, scale_lam_sideinfo", "UI", dict(use_cg=True, finalize_chol=True, scale_lam_sideinfo=True)),
    ("cg, k_user / k_item / k_main, user bias", "UI", dict(use_cg=True, finalize_chol=False, center=False, item_bias=False, k_user=2, k_item=1, k_main=2,
                                                              w_user=0.7, w_item=1.3)),
    ("cg, user side only, scale_lam", "U", dict(use_cg=True, finalize_chol=False, scale_lam=True)),
]


def _naz_side(d, sides):
    return (d["U"] if "U" in sides else None), (d["I"] if "I" in sides else None)


# ... and with observation weights on top (round 5): the rows with entries leave the shared factorisation for the general branch
# of collective_closed_form_block (collective.c:1367-1372 -> :1534-1846) -- fixture g28, closed form, the problem of g24 (entries
# ordered by column, see weights_problem)
NAZ_WEIGHTED_SIDE_CASES = [
    ("both sides, biases", "UI", dict()),
    ("both sides, scale_lam", "UI", dict(scale_lam=True)),
    ("both sides, scale_lam_sideinfo", "UI", dict(scale_lam_sideinfo=True)),
    ("no biases, no centring", "UI", dict(user_bias=False, item_bias=False, center=False)),
    ("k_user / k_item / k_main, weights of the sides, user bias", "UI", dict(center=False, item_bias=False, k_user=2, k_item=1, k_main=2,
                                                                               w_user=0.7, w_item=1.3)),
    ("user side only", "U", dict()),
    ("item side only, scale_lam", "I", dict(scale_lam=True)),
    ("per-matrix lambdas, item bias", "UI", dict(lam_unique=LAM6, user_bias=False)),
]


# ... under use_cg (round 6; fixture g35): collective_block_cg's NA_as_zero_X + weight branches (collective.c:2446-2493, :2700-2760) for
# the rows with entries, the shared factorisation for the rows without (:1367-1372)
NAZ_WEIGHTED_SIDE_CG_CASES = [
    ("cg, both sides", "UI", dict(use_cg=True, finalize_chol=False)),
    ("cg, scaled", "UI", dict(use_cg=True, finalize_chol=False, scale_lam=True, scale_lam_sideinfo=True)),
    ("pcg, user side, k_user, user bias", "U", dict(use_cg=True, precondition_cg=True, finalize_chol=False, k_user=2, center=False, item_bias=False)),
    ("cg + finalize, item side", "I", dict(use_cg=True, finalize_chol=True)),
]


def naz_side_reference(R, d, sides, opts, nthreads=2, weights=False):
    o = dict(opts)
    seed = o.pop("seed", None)
    A0, B0 = _impf_start(d, o)
    U, II = _naz_side(d, sides)
    kw = dict(U=U, II=II, lam=0.3, niter=o.pop("niter", 3), nthreads=nthreads, NA_as_zero_X=True, use_cg=o.pop("use_cg", False),
              finalize_chol=o.pop("finalize_chol", False), **o)
    if weights: kw["weight"] = d["W"]
    if seed is not None:
        A0[:] = 0; B0[:] = 0
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], reset_values=True, seed=seed, **kw)
    else:
        r = R.fit_collective_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if U is not None: out["C"] = r["C"]
    if II is not None: out["D"] = r["D"]
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_side_oracle(O, d, sides, opts, nthreads=2, weights=False):
    """None for the seeded cases (the oracle has no random start)."""
    o = dict(opts)
    if "seed" in o:
        return None
    lam6 = o.pop("lam_unique", None)
    if lam6 is not None:
        O.set_lam_unique(np.asarray(lam6, np.float64), None)
    U, II = _naz_side(d, sides)
    try:
        A0, B0 = _impf_start(d, o)
        r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), U=U, II=II,
                               lam=0.3, niter=3, nthreads=nthreads, NA_as_zero_X=True, use_cg=o.pop("use_cg", False),
                               finalize_chol=o.pop("finalize_chol", False), weight=d["W"] if weights else None, **o)
    finally:
        O.set_lam_unique(None, None)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if U is not None: out["C"] = r["C"]
    if II is not None: out["D"] = r["D"]
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def naz_side_hip(d, sides, opts, dtype, weights=False, **ctor):
    from cmfrec_amd import CMF
    o = dict(opts)
    seed = o.pop("seed", None)
    o["lambda_"] = o.pop("lam_unique") if "lam_unique" in o else 0.3
    A0, B0 = _impf_start(d, o)
    U, II = _naz_side(d, sides)
    args = dict(k=d["k"], niter=3, use_float=dtype is np.float32, precompute_for_predictions=False, NA_as_zero=True, use_cg=False,
                finalize_chol=False, nthreads=1)
    args.update(dict(random_state=seed) if seed is not None else {})
    args.update(o); args.update(ctor)
    mdl = CMF(**args)
    start = {} if seed is not None else dict(A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"] if weights else None, **start)
    out = dict(A=mdl.A_, B=mdl.B_, glob_mean=mdl.glob_mean_)
    if U is not None: out["C"] = mdl.C_
    if II is not None: out["D"] = mdl.D_
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- dense X (NaN = missing): optimizeA Cases 1-2, common.c:2787-3116 ------------------------------------------------
def dense_problem(dtype, variant, seed=131):
    """variant: 'full' no missing entry; 'near' 13 % of the rows and 10 % of the columns have missing entries (both half-steps
    are Case 1); 'holes' half of the cells missing, one row and one column entirely (Case 2, every row misses many entries: the
    solver asked for); 'mixed' 10 % of the rows miss half of their cells, so the rows are near dense (Case 1) and the columns are
    not (Case 2) -- but every column misses fewer than 2 k entries, which the reference solves in closed form from the
    precomputed B^T B whatever use_cg says (factors_closed_form, common.c:662, :759-790)."""
    rng = np.random.default_rng(seed)
    m, n, k = 90, 60, 6
    X = (0.5 * rng.integers(1, 11, (m, n))).astype(dtype)
    if variant == "near":
        rows = rng.choice(m, 12, replace=False); cols = rng.choice(n, 6, replace=False)
        for r in rows:
            X[r, rng.choice(cols, 3, replace=False)] = np.nan
    elif variant == "holes":
        X[rng.random((m, n)) < 0.5] = np.nan
        X[4, :] = np.nan; X[:, 7] = np.nan
    elif variant == "mixed":
        for r in rng.choice(m, 9, replace=False):
            X[r, rng.random(n) < 0.5] = np.nan
    elif variant in ("split", "split2"):
        # half of the rows miss 4 entries, the other half 30 of 60: neither complete nor near dense, and one half-step holds rows on
        # both sides of the 2 k boundary (closed form next to the solver asked for).  split2: ten columns miss 3 entries only, so the
        # columns split as well
        for r in range(m):
            X[r, rng.choice(n, 4 if r % 2 == 0 else 30, replace=False)] = np.nan
        if variant == "split2":
            for c in rng.choice(n, 10, replace=False):
                X[:, c] = (0.5 * rng.integers(1, 11, m)).astype(dtype)
                X[rng.choice(m, 3, replace=False), c] = np.nan
    d = dict(m=m, n=n, k=k, X=X)
    d["row"], d["col"] = [a.astype(np.int32) for a in np.nonzero(~np.isnan(X))]     # row-major order of the present entries
    d["ratings"] = X[d["row"], d["col"]]
    d["Wfull"] = (0.25 * rng.integers(1, 9, (m, n))).astype(dtype)
    d["W"] = d["Wfull"][d["row"], d["col"]]
    d["A0"] = (rng.standard_normal((m, k)) * 0.1).astype(dtype); d["B0"] = (rng.standard_normal((n, k)) * 0.1).astype(dtype)
    d["bA"] = (rng.standard_normal(m) * 0.1).astype(dtype); d["bB"] = (rng.standard_normal(n) * 0.1).astype(dtype)
    return d


# (name, variant, options).  weights: dense weights (Case 2 whatever the pattern).  seed: the reference's own random start and
# its dense bias start values.
DENSE_CASES = [
    ("full, cg asked for", "full", dict(use_cg=True, finalize_chol=True)),
    ("full, chol, scale_lam", "full", dict(use_cg=False, scale_lam=True)),
    ("near dense, chol", "near", dict(use_cg=False)),
    ("near dense, cg asked for", "near", dict(use_cg=True, finalize_chol=False)),
    ("holes, cg", "holes", dict(use_cg=True, finalize_chol=True, scale_lam=True)),
    ("holes, chol, no biases", "holes", dict(use_cg=False, user_bias=False, item_bias=False)),
    ("mixed, cg", "mixed", dict(use_cg=True, finalize_chol=False)),
    ("holes, weights, cg", "holes", dict(use_cg=True, finalize_chol=False, weights=True)),
    ("full, weights, chol", "full", dict(use_cg=False, weights=True, scale_lam=True)),
    ("near dense, nonneg", "near", dict(use_cg=False, nonneg=True, user_bias=False, item_bias=False, center=False)),
    ("holes, seeded", "holes", dict(use_cg=True, finalize_chol=True, seed=5)),
    ("near dense, seeded, user bias", "near", dict(use_cg=False, item_bias=False, seed=6)),
    ("full, seeded", "full", dict(use_cg=True, finalize_chol=False, seed=7)),
    ("rows on both sides of the 2 k boundary, cg", "split", dict(use_cg=True, finalize_chol=False)),
    ("rows and columns on both sides, cg + finalize, user bias", "split2", dict(use_cg=True, finalize_chol=True, item_bias=False)),
    ("rows and columns on both sides, chol", "split2", dict(use_cg=False)),
    ("near dense, chol, scale_lam", "near", dict(use_cg=False, scale_lam=True)),
    ("rows and columns on both sides, cg, scale_lam", "split2", dict(use_cg=True, finalize_chol=False, scale_lam=True)),
    # (under use_cg the reference runs k CG steps from zero on the rows of a near-dense half-step that miss many entries, common.c:2944-2983;
    #  the closed form here is the same solution until that CG leaves through its 1e-8 exit early -- with scale_lam's larger lambda it does,
    #  1e-5 apart -- so the scaled mixed pattern is pinned with the closed form)
    ("mixed, chol, scale_lam, no biases", "mixed", dict(use_cg=False, scale_lam=True, user_bias=False, item_bias=False)),
]


def dense_reference(R, d, opts, nthreads=2):
    o = dict(opts)
    seed = o.pop("seed", None)
    W = d["Wfull"] if o.pop("weights", False) else None
    kw = dict(use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), **o)
    A0, B0 = d["A0"].copy(), d["B0"].copy()
    if seed is not None:
        A0[:] = 0; B0[:] = 0
        r = R.fit_collective_explicit_als(A0, B0, None, None, None, d["k"], lam=0.3, niter=3, nthreads=nthreads, Xfull=d["X"], weight=W,
                                          reset_values=True, seed=seed, **kw)
    else:
        r = R.fit_collective_explicit_als(A0, B0, None, None, None, d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3, niter=3,
                                          nthreads=nthreads, Xfull=d["X"], weight=W, **kw)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def dense_oracle(O, d, variant, opts, nthreads=2):
    """The present entries as a sparse X.  Without weights the 'full' / 'near' / 'mixed' patterns are closed-form solves
    whatever use_cg says (Case 1, and Case 2 with few missing entries per row).  None for seeded starts and non-negative factors."""
    o = dict(opts)
    if "seed" in o or o.get("nonneg"):
        return None
    W = d["W"] if o.pop("weights", False) else None
    use_cg, fin = o.pop("use_cg", False), o.pop("finalize_chol", False)
    if W is None and variant in ("full", "near", "mixed"):
        use_cg = fin = False
    if W is None and use_cg and variant in ("split", "split2"):
        # Case 2 with rows on both sides of the boundary: closed form below 2 k missing entries, the solver asked for above
        na_r = np.isnan(d["X"]).sum(1); na_c = np.isnan(d["X"]).sum(0)
        few_r = 2 * (d["k"] + o.get("k_main", 0) + int(o.get("user_bias", True))); few_c = 2 * (d["k"] + o.get("k_main", 0) + int(o.get("item_bias", True)))
        O.set_closed_form_rows(na_r < few_r, na_c < few_c)
    if W is None and o.get("scale_lam") and variant != "full":
        # under scale_lam a row that misses fewer than 2 k entries keeps the n lam of a complete row, the others lam times their
        # present entries: unit weights + per-row multipliers
        na_r = np.isnan(d["X"]).sum(1); na_c = np.isnan(d["X"]).sum(0)
        few_r = 2 * (d["k"] + o.get("k_main", 0) + int(o.get("user_bias", True))); few_c = 2 * (d["k"] + o.get("k_main", 0) + int(o.get("item_bias", True)))
        if ((na_r > 0) & (na_r < few_r)).any() or ((na_c > 0) & (na_c < few_c)).any():
            dt = d["X"].dtype
            mult_r = np.where(na_r < few_r, d["n"], np.where(na_r < d["n"], d["n"] - na_r, 1)).astype(dt)
            mult_c = np.where(na_c < few_c, d["m"], np.where(na_c < d["m"], d["m"] - na_c, 1)).astype(dt)
            O.set_lambda_multipliers(mult_r, mult_c)
            W = np.ones(len(d["ratings"]), dt)
    # rows / columns without a present entry: zero in the dense reference (factors and bias), left alone by the sparse path
    A0, B0, bA, bB = d["A0"].copy(), d["B0"].copy(), d["bA"].copy(), d["bB"].copy()
    er = np.bincount(d["row"], minlength=d["m"]) == 0; ec = np.bincount(d["col"], minlength=d["n"]) == 0
    A0[er] = 0; bA[er] = 0; B0[ec] = 0; bB[ec] = 0
    r = O.fit_explicit_als(A0, B0, d["row"], d["col"], d["ratings"], d["k"], biasA=bA, biasB=bB,
                           lam=0.3, niter=3, nthreads=nthreads, weight=W, use_cg=use_cg, finalize_chol=fin, **o)
    assert r["ret"] == 0
    r["biasA"][er] = 0; r["biasB"][ec] = 0
    out = dict(A=r["A"], B=r["B"], glob_mean=r["glob_mean"])
    if opts.get("user_bias", True): out["biasA"] = r["biasA"]
    if opts.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def dense_hip(d, opts, dtype, as_sparse=False):
    """as_sparse: the same entries as a COO triplet (the solver then follows use_cg in both half-steps)."""
    from cmfrec_amd import CMF
    o = dict(opts)
    seed = o.pop("seed", None)
    weights = o.pop("weights", False)
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, use_float=dtype is np.float32, precompute_for_predictions=False,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1,
              **(dict(random_state=seed) if seed is not None else {}), **o)
    start = {} if seed is not None else dict(A0=d["A0"], B0=d["B0"], biasA0=d["bA"], biasB0=d["bB"])
    if as_sparse:
        mdl.fit((d["row"], d["col"], d["ratings"]), shape=(d["m"], d["n"]), W=d["W"] if weights else None, **start)
    else:
        mdl.fit(d["X"], W=d["Wfull"] if weights else None, **start)
    out = dict(A=mdl.A_, B=mdl.B_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- dense X together with side information (round 6; fixture g33): optimizeA_collective's dense-X branches
#      (collective.c:5115-5565 shared factorisation + corrections, :5566-5968 row by row) ------------------------------------------
def dense_side_problem(dtype, variant, seed=137):
    d = dense_problem(dtype, variant)
    rng = np.random.default_rng(seed)
    m, n, p, q = d["m"], d["n"], 5, 4
    d["U"] = rng.standard_normal((m, p)).astype(dtype); d["I"] = rng.standard_normal((n, q)).astype(dtype)
    def coo(rows, cols, cnt, empty):
        lin = rng.choice(rows * cols, size=cnt, replace=False)
        r = (lin // cols).astype(np.int32); c = (lin % cols).astype(np.int32)
        keep = ~np.isin(r, empty)
        return r[keep], c[keep]
    ur, uc = coo(m, p, 3 * m, (2, 4)); ir, ic = coo(n, q, 2 * n, (7, 11))
    d["U_coo"] = (ur, uc, rng.standard_normal(len(ur)).astype(dtype), m, p)
    d["I_coo"] = (ir, ic, rng.standard_normal(len(ir)).astype(dtype), n, q)
    d["p"], d["q"] = p, q
    return d


# (name, pattern of X, sides -- small letters dense, capitals sparse --, options)
DENSE_SIDE_CASES = [
    ("full, dense UI, chol", "full", "ui", dict(use_cg=False, k_user=1, k_item=1)),
    ("full, dense UI, cg asked for, scaled", "full", "ui", dict(use_cg=True, finalize_chol=False, scale_lam=True, scale_lam_sideinfo=True)),
    ("holes, dense UI, cg", "holes", "ui", dict(use_cg=True, finalize_chol=True, k_user=2)),
    ("holes, dense UI, chol, scale_lam", "holes", "ui", dict(use_cg=False, scale_lam=True)),
    ("holes, sparse UI, chol", "holes", "UI", dict(use_cg=False, k_item=1)),
    # (a nearly complete X -- "near": the shared factorisation + row-by-row corrections of collective.c:5115-5565 -- with DENSE side
    #  information on either side: the reference's own build writes out of bounds.  With one sparse and one dense side it dies with
    #  SIGSEGV at once, either solver, both precisions; with dense side information only it returns, but a process that repeats the
    #  call a few times dies in the allocator or the garbage collector (ten repetitions of "near" with "u", "i" or "ui", either solver:
    #  every one of them; all the cases below: clean).  Nothing to be pinned against, so "near" appears with sparse side information only.)
    ("near dense, sparse UI, cg", "near", "UI", dict(use_cg=True, finalize_chol=False)),
    ("full, sparse U + dense I, cg", "full", "Ui", dict(use_cg=True, finalize_chol=False, k_user=1)),
    ("holes, sparse UI, pcg, no biases", "holes", "UI", dict(use_cg=True, precondition_cg=True, finalize_chol=False, user_bias=False, item_bias=False)),
    ("rows and columns on both sides, dense U, chol, scale_lam", "split2", "u", dict(use_cg=False, scale_lam=True)),
]


def _dense_sides(d, which):
    return (d["U"] if "u" in which else None, d["I"] if "i" in which else None,
            d["U_coo"] if "U" in which else None, d["I_coo"] if "I" in which else None)


def dense_side_reference(R, d, which, opts, nthreads=2):
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    ku, ki = o.get("k_user", 0), o.get("k_item", 0)
    Ud, Id, Us, Is = _dense_sides(d, which)
    r = R.fit_collective_explicit_als(A0, B0, None, None, None, d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(), lam=0.3, niter=3,
                                      w_user=2.0, w_item=0.5, nthreads=nthreads, Xfull=d["X"],
                                      use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False),
                                      U=Ud, II=Id, U_coo=Us, I_coo=Is,
                                      Cm=np.zeros((d["p"], ku + d["k"]), d["A0"].dtype) if Us is not None else None,
                                      Dm=np.zeros((d["q"], ki + d["k"]), d["A0"].dtype) if Is is not None else None, **o)
    assert r["ret"] == 0
    out = dict(A=r["A"], B=r["B"], C=r["C"], D=r["D"], glob_mean=r["glob_mean"])
    if o.get("user_bias", True): out["biasA"] = r["biasA"]
    if o.get("item_bias", True): out["biasB"] = r["biasB"]
    return out


def dense_side_hip(d, which, opts, dtype):
    import scipy.sparse as sp
    from cmfrec_amd import CMF
    o = dict(opts)
    if "U" not in which.upper(): o["k_user"] = 0
    if "I" not in which.upper(): o["k_item"] = 0
    A0, B0 = _impf_start(d, o)
    mk = lambda c: sp.coo_matrix((c[2], (c[0], c[1])), shape=(c[3], c[4]))
    Ud, Id, Us, Is = _dense_sides(d, which)
    U = mk(Us) if Us is not None else Ud; II = mk(Is) if Is is not None else Id
    mdl = CMF(k=d["k"], lambda_=0.3, niter=3, w_user=2.0, w_item=0.5, use_float=dtype is np.float32, precompute_for_predictions=False,
              use_cg=o.pop("use_cg", False), finalize_chol=o.pop("finalize_chol", False), nthreads=1, **o)
    mdl.fit(d["X"], U=U, I=II, A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
    out = dict(A=mdl.A_, B=mdl.B_, C=mdl.C_, D=mdl.D_, glob_mean=mdl.glob_mean_)
    if mdl.user_bias: out["biasA"] = mdl.user_bias_
    if mdl.item_bias: out["biasB"] = mdl.item_bias_
    return out


# ---- the reference's stand-alone prediction matrices and per-user ranking under their own names (round 6; fixtures g29 / g30):
# ---- precompute_collective_explicit / _implicit (src/collective.c:10209-10566), topN_old_collective_explicit / _implicit
# ---- (:11546-11613 over topN, src/common.c:5127-5380).  One marshalling for either shared object -- the product's
# ---- (cmfrec_amd._lib.load) or the compiled reference's (oracle.bindings.Reference(dtype).lib): same C signature.
import ctypes as _C


def _p(a):
    return None if a is None else a.ctypes.data_as(_C.c_void_p)


def _real(dt):
    return _C.c_double if dt is np.float64 else _C.c_float


def precompute_problem(dtype):
    rng = np.random.default_rng(2906)
    n, n_max, k, p = 140, 150, 6, 9
    return dict(n=n, n_max=n_max, k=k, p=p, rng_seed=2906,
                glob_mean=dtype(0.37), biasB=(rng.standard_normal(n_max) * 0.3).astype(dtype),
                U_colmeans=(rng.standard_normal(p) * 0.5).astype(dtype),
                lam6=np.array([0.11, 0.23, 0.31, 0.43, 0.53, 0.61], dtype))


# (name, options): k_user / k_item / k_main and the switches of the signature; "C": user side information present; "Bi": implicit features
PRECOMPUTE_EXPLICIT_CASES = [
    ("plain, user bias", dict()),
    ("no user bias", dict(user_bias=False)),
    ("side information, k_user / k_item / k_main, w_user, lam_unique", dict(C=True, k_user=2, k_item=1, k_main=2, w_user=0.7, lam_unique=True)),
    ("side information, scale_lam + scale_lam_sideinfo", dict(C=True, scale_lam=True, scale_lam_sideinfo=True, k_user=1)),
    ("scale_lam, scale_bias_const", dict(scale_lam=True, scale_bias_const=True, scaling_biasA=17.5)),
    ("w_main, side information", dict(C=True, w_main=1.6, w_user=0.8, k_main=1)),
    ("NA_as_zero_X: BtXbias, rows beyond n", dict(NA_as_zero_X=True, with_biasB=True)),
    ("NA_as_zero_X, no item bias, include_all_X", dict(NA_as_zero_X=True, include_all_X=True, user_bias=False)),
    ("NA_as_zero_U: CtUbias", dict(C=True, NA_as_zero_U=True, w_user=1.3, k_user=2)),
    ("implicit features", dict(Bi=True, w_implicit=0.6, k_main=1)),
    ("implicit features + side information", dict(Bi=True, C=True, w_implicit=0.5, k_user=1)),
    ("nonneg, side information", dict(C=True, nonneg=True)),
]
PRECOMPUTE_IMPLICIT_CASES = [
    ("plain", dict()),
    ("side information, k_user / k_item / k_main, w_user", dict(C=True, k_user=2, k_item=1, k_main=2, w_user=0.7)),
    ("w_main and its multiplier", dict(C=True, w_main=1.4, w_main_multiplier=2.5, w_user=0.9)),
    ("NA_as_zero_U, nonneg", dict(C=True, NA_as_zero_U=True, nonneg=True, k_user=1)),
]


def _precompute_inputs(d, o, dtype, implicit):
    rng = np.random.default_rng(d["rng_seed"] + 1)
    k, ku, ki, km = d["k"], o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
    rows = d["n"] if implicit else d["n_max"]
    B = (rng.standard_normal((rows, ki + k + km)) * 0.4).astype(dtype)
    Cm = (rng.standard_normal((d["p"], ku + k)) * 0.4).astype(dtype) if o.get("C") else None
    Bi = (rng.standard_normal((d["n_max"], k + km)) * 0.4).astype(dtype) if o.get("Bi") else None
    return B, Cm, Bi


def precompute_explicit_call(lib, d, opts, dtype):
    o = dict(opts); R = _real(dtype)
    B, Cm, Bi = _precompute_inputs(d, o, dtype, False)
    k, ku, ki, km = d["k"], o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
    ub = o.get("user_bias", True)
    kk = k + km + int(ub); kc = ku + k; kq = ku + kk; p = d["p"] if Cm is not None else 0
    n, n_max = d["n"], d["n_max"]
    n_used = n_max if o.get("include_all_X") else n
    out = dict(B_plus_bias=np.zeros((n_max, ki + k + km + 1), dtype), BtB=np.zeros((kk, kk), dtype),
               TransBtBinvBt=np.zeros((n_used, kk), dtype), BtXbias=np.zeros(kk, dtype), BeTBeChol=np.zeros((kq, kq), dtype),
               BiTBi=np.zeros((k + km, k + km), dtype), TransCtCinvCt=np.zeros((max(p, 1), max(kc, 1)), dtype),
               CtCw=np.zeros((max(kc, 1), max(kc, 1)), dtype), CtUbias=np.zeros(max(kc, 1), dtype))
    biasB = d["biasB"].copy() if o.get("with_biasB") else None
    lam6 = d["lam6"].copy() if o.get("lam_unique") else None
    fn = lib.precompute_collective_explicit
    fn.restype = _C.c_int
    ret = fn(_p(B), _C.c_int(n), _C.c_int(n_max), _C.c_bool(bool(o.get("include_all_X"))),
             _p(Cm), _C.c_int(p), _p(Bi), _C.c_bool(Bi is not None),
             _p(biasB), R(float(d["glob_mean"])), _C.c_bool(bool(o.get("NA_as_zero_X"))),
             _p(d["U_colmeans"].copy()), _C.c_bool(bool(o.get("NA_as_zero_U"))),
             _C.c_int(k), _C.c_int(ku), _C.c_int(ki), _C.c_int(km),
             _C.c_bool(ub), _C.c_bool(bool(o.get("nonneg"))),
             R(0.35), _p(lam6), _C.c_bool(bool(o.get("scale_lam"))), _C.c_bool(bool(o.get("scale_lam_sideinfo"))),
             _C.c_bool(bool(o.get("scale_bias_const"))), R(o.get("scaling_biasA", 0.0)),
             R(o.get("w_main", 1.0)), R(o.get("w_user", 1.0)), R(o.get("w_implicit", 1.0)),
             _p(out["B_plus_bias"]) if ub else None, _p(out["BtB"]), _p(out["TransBtBinvBt"]), _p(out["BtXbias"]), _p(out["BeTBeChol"]),
             _p(out["BiTBi"]) if Bi is not None else None, _p(out["TransCtCinvCt"]) if p else None, _p(out["CtCw"]) if p else None,
             _p(out["CtUbias"]) if p else None)
    assert ret == 0, ret
    # only what the call defines: upper triangles of the symmetric outputs; matrices the options leave untouched are dropped
    res = dict(BtB=np.triu(out["BtB"]))
    if ub: res["B_plus_bias"] = out["B_plus_bias"]
    if not o.get("nonneg") and Bi is None: res["TransBtBinvBt"] = out["TransBtBinvBt"]
    if o.get("NA_as_zero_X"): res["BtXbias"] = out["BtXbias"]
    if Bi is not None: res["BiTBi"] = np.triu(out["BiTBi"])
    if p:
        res["CtCw"] = np.triu(out["CtCw"])
        if not o.get("nonneg") and Bi is None: res["TransCtCinvCt"] = out["TransCtCinvCt"]
        if o.get("NA_as_zero_U"): res["CtUbias"] = out["CtUbias"]
    if (p or Bi is not None) and not o.get("nonneg"): res["BeTBeChol"] = np.triu(out["BeTBeChol"])
    return res


def precompute_implicit_call(lib, d, opts, dtype):
    o = dict(opts); R = _real(dtype)
    B, Cm, _ = _precompute_inputs(d, o, dtype, True)
    k, ku, ki, km = d["k"], o.get("k_user", 0), o.get("k_item", 0), o.get("k_main", 0)
    kk = k + km; kc = ku + k; kq = ku + kk; p = d["p"] if Cm is not None else 0
    out = dict(BtB=np.zeros((kk, kk), dtype), BeTBe=np.zeros((kq, kq), dtype), BeTBeChol=np.zeros((kq, kq), dtype), CtUbias=np.zeros(max(kc, 1), dtype))
    fn = lib.precompute_collective_implicit
    fn.restype = _C.c_int
    ret = fn(_p(B), _C.c_int(d["n"]), _p(Cm), _C.c_int(p), _p(d["U_colmeans"].copy()), _C.c_bool(bool(o.get("NA_as_zero_U"))),
             _C.c_int(k), _C.c_int(ku), _C.c_int(ki), _C.c_int(km),
             R(0.8), R(o.get("w_main", 1.0)), R(o.get("w_user", 1.0)), R(o.get("w_main_multiplier", 1.0)),
             _C.c_bool(bool(o.get("nonneg"))), _C.c_bool(False),
             _p(out["BtB"]), _p(out["BeTBe"]), _p(out["BeTBeChol"]), _p(out["CtUbias"]))
    assert ret == 0, ret
    res = dict(BtB=np.triu(out["BtB"]))
    if p:
        res["BeTBe"] = np.triu(out["BeTBe"])
        if not o.get("nonneg"): res["BeTBeChol"] = np.triu(out["BeTBeChol"])
        if o.get("NA_as_zero_U"): res["CtUbias"] = out["CtUbias"]
    return res


def topn_problem(dtype):
    rng = np.random.default_rng(3007)
    n, n_max, m, k, ku, ki, km = 500, 520, 7, 8, 2, 1, 1
    return dict(n=n, n_max=n_max, m=m, k=k, k_user=ku, k_item=ki, k_main=km,
                A=rng.standard_normal((m, ku + k + km)).astype(dtype), B=rng.standard_normal((n_max, ki + k + km)).astype(dtype),
                biasA=(rng.standard_normal(m) * 0.2).astype(dtype), biasB=(rng.standard_normal(n_max) * 0.2).astype(dtype),
                glob_mean=dtype(3.1),
                excl_few=np.sort(rng.choice(n, 12, replace=False)).astype(np.int32),
                excl_many=rng.permutation(n)[:130].astype(np.int32),          # more than n / 20, unsorted
                incl=rng.permutation(n)[:60].astype(np.int32))


# (name, options)
TOPN_CASES = [
    ("all items, row of A", dict(n_top=10)),
    ("own vector and bias", dict(n_top=7, a_vec=True)),
    ("few exclusions", dict(n_top=15, exclude="excl_few")),
    ("many exclusions, unsorted", dict(n_top=15, exclude="excl_many")),
    ("include list", dict(n_top=9, include="incl")),
    ("include list, all of it", dict(n_top=60, include="incl")),
    ("include_all_X (n_max items)", dict(n_top=12, include_all_X=True)),
    ("more than 128", dict(n_top=200)),
    ("implicit model", dict(n_top=10, implicit=True, exclude="excl_few")),
]


def topn_call(lib, d, opts, dtype, expect=0):
    o = dict(opts); R = _real(dtype)
    n_top = o["n_top"]
    incl = d[o["include"]].copy() if o.get("include") else None
    excl = d[o["exclude"]].copy() if o.get("exclude") else None
    ids = np.full(n_top, -7, np.int32); sc = np.zeros(n_top, dtype)
    row_index = 3
    A, B = d["A"].copy(), d["B"].copy()
    a_vec = (A[5] * dtype(0.5)).copy() if o.get("a_vec") else None
    common = (_C.c_int(d["k"]), _C.c_int(d["k_user"]), _C.c_int(d["k_item"]), _C.c_int(d["k_main"]),
              _p(incl), _C.c_int(0 if incl is None else len(incl)), _p(excl), _C.c_int(0 if excl is None else len(excl)),
              _p(ids), _p(sc), _C.c_int(n_top), _C.c_int(d["n"]))
    if o.get("implicit"):
        fn = lib.topN_old_collective_implicit
        fn.restype = _C.c_int
        ret = fn(_p(a_vec), _p(A), _C.c_int(row_index), _p(B), *common, _C.c_int(1))
    else:
        fn = lib.topN_old_collective_explicit
        fn.restype = _C.c_int
        ret = fn(_p(a_vec), R(0.25), _p(A), _p(d["biasA"].copy()), _C.c_int(row_index), _p(B), _p(d["biasB"].copy()), R(float(d["glob_mean"])),
                 *common, _C.c_int(d["n_max"]), _C.c_bool(bool(o.get("include_all_X"))), _C.c_int(1))
    assert ret == expect, (ret, expect)
    return dict(ids=ids, scores=sc)
