"""CPU: the oracle (oracle/cmf_oracle.c) against the golden vectors captured from the real
reference (tests/golden/make_golden.py).  This is what pins the oracle on machines where
/root/reference does not exist.  Tolerances: per-call 1e-10 fp64 / 2e-4 fp32;
whole fits 1e-6 fp64 / 1e-2 fp32 relative Frobenius."""
import numpy as np
import pytest

import golden_cases as gc

DT = [np.float64, np.float32]
TOL = {np.float64: 1e-10, np.float32: 2e-4}
TOL_FIT = {np.float64: 1e-6, np.float32: 1e-2}


@pytest.mark.parametrize("dtype", DT)
def test_prep(oracles, dtype):
    O = oracles[dtype]
    g = gc.load("g6_prep", dtype)
    m, n = int(g["m"]), int(g["n"])
    csr, csc = O.coo_to_csr_and_csc(g["row"], g["col"], g["val"], m, n)
    for got, name in zip(csr + csc, ("csr_p", "csr_i", "csr_v", "csc_p", "csc_i", "csc_v")):
        assert np.array_equal(got, g[name]), name          # stable ordering is exact
    v = g["val"].copy()
    gm = O.calc_mean_and_center(v, nthreads=1)
    assert gm == g["glob_mean"] and np.array_equal(v, g["val_centered"])
    csr_c, csc_c = O.coo_to_csr_and_csc(g["row"], g["col"], v, m, n)
    bA, bB = O.initialize_biases_twosided(m, n, csr_c, csc_c, float(g["lam_bias"]), float(g["lam_bias"]), True)
    assert np.array_equal(bA, g["biasA"]) and np.array_equal(bB, g["biasB"])   # bit-exact


@pytest.mark.parametrize("dtype", DT)
def test_operators(oracles, dtype):
    O = oracles[dtype]
    worst = 0.0
    for name, got, exp in gc.implicit_cases(gc.load("g1_implicit", dtype), O, O.optimizeA_implicit):
        e = gc.maxrel(got, exp); worst = max(worst, e)
        assert e < TOL[dtype], (name, e)
    for name, got, exp in gc.explicit_cases(gc.load("g2_explicit", dtype), O, O.optimizeA_explicit):
        e = gc.maxrel(got, exp)
        assert e < TOL[dtype], (name, e)
    for name, got, exp in gc.collective_cases(gc.load("g3_collective", dtype), O, O.optimizeA_collective_chol):
        e = gc.maxrel(got, exp)
        assert e < TOL[dtype], (name, e)
    g = gc.load("g4_dense_full", dtype)
    C = np.zeros_like(g["C"])
    O.optimizeA_dense_full(C, g["A_bias"], g["U"], float(g["lam"]), k=int(g["kc"]), do_B=True, scale_lam=True)
    assert gc.maxrel(C, g["C"]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_gram_matches_reference_syrk(oracles, dtype):
    O = oracles[dtype]
    g = gc.load("g1_implicit", dtype)
    csr, _ = gc.csr_from(g, O)
    for k in (8, 50, 64):
        A = g["A0_k%d" % k].copy()
        BtB = O.optimizeA_implicit(A, g["B_k%d" % k], csr, float(g["lam"]), return_BtB=True)
        assert gc.maxrel(np.triu(BtB), g["BtB_k%d" % k]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_fits(oracles, dtype):
    O = oracles[dtype]
    t = TOL_FIT[dtype]
    g = gc.load("g5_fit_implicit", dtype)
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    for mode in ("cg", "chol", "cgfin"):
        A, B = g["A0"].copy(), np.zeros((n, k), dtype)
        O.fit_implicit_als(A, B, g["row"], g["col"], g["val"], lam=float(g["lam"]), alpha=float(g["alpha"]),
                           niter=int(g["niter"]), use_cg=mode != "chol", finalize_chol=mode == "cgfin")
        assert gc.frob(A, g["A_" + mode]) < t and gc.frob(B, g["B_" + mode]) < t, mode
    g = gc.load("g5_fit_explicit", dtype)
    for mode in ("cg", "chol", "cgfin"):
        A, B = g["A0"].copy(), np.zeros((n, k), dtype)
        r = O.fit_explicit_als(A, B, g["row"], g["col"], g["val"], k, biasA=g["biasA0"].copy(), biasB=g["biasB0"].copy(),
                               lam=float(g["lam"]), scale_lam=True, niter=int(g["niter"]), use_cg=mode != "chol",
                               finalize_chol=mode == "cgfin")
        assert r["ret"] == 0 and r["glob_mean"] == g["glob_mean"]
        assert gc.frob(A, g["A_" + mode]) < t and gc.frob(B, g["B_" + mode]) < t, mode
        assert gc.frob(r["biasA"], g["biasA_" + mode]) < t and gc.frob(r["biasB"], g["biasB_" + mode]) < t
    g = gc.load("g5_fit_sideinfo", dtype)
    ku, ki, km = [int(x) for x in g["cfg"]]
    A, B = g["A0"].copy(), g["B0"].copy()
    r = O.fit_explicit_als(A, B, g["row"], g["col"], g["val"], k, lam=0.05, scale_lam=True, scale_lam_sideinfo=True,
                           niter=3, use_cg=False, U=g["U"], II=g["II"], k_user=ku, k_item=ki, k_main=km, w_user=0.5,
                           w_item=2.0)
    assert r["ret"] == 0
    for got, key in ((A, "A"), (B, "B"), (r["C"], "C"), (r["D"], "D"), (r["biasA"], "biasA"), (r["biasB"], "biasB")):
        assert gc.frob(got, g[key]) < t, key
    assert gc.maxrel(r["U_colmeans"], g["U_colmeans"]) < 1e-6
    g = gc.load("g8_fit_implicit_sideinfo", dtype)
    ku, ki, km = [int(x) for x in g["cfg"]]
    A, B = g["A0"].copy(), g["B0"].copy()
    r = O.fit_implicit_als_sideinfo(A, B, g["row"], g["col"], g["val"], k, lam=3.0, alpha=2.0, niter=3, use_cg=False,
                                    U=g["U"], II=g["II"], k_user=ku, k_item=ki, k_main=km, w_main=0.5, w_user=4.0,
                                    w_item=0.8)
    assert r["ret"] == 0
    for got, key in ((A, "A"), (B, "B"), (r["C"], "C"), (r["D"], "D")):
        assert gc.frob(got, g[key]) < t, key
    assert gc.maxrel(r["U_colmeans"], g["U_colmeans"]) < 1e-6 and gc.maxrel(r["I_colmeans"], g["I_colmeans"]) < 1e-6


@pytest.mark.parametrize("dtype", DT)
def test_result_metrics(oracles, dtype):
    """RMSE (explicit) and P@10 (implicit) of 15-iteration fits equal the reference's (SURVEY.md 8d:
    RMSE to 1e-6 fp64 / 1e-4 fp32, P@10 to 1e-4)."""
    O = oracles[dtype]
    g = gc.load("g10_metrics", dtype)
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    A, B = g["A0"].copy(), np.zeros((n, k), dtype)
    r = O.fit_explicit_als(A, B, g["e_row"], g["e_col"], g["e_val"], k, lam=0.05, scale_lam=True, niter=15, use_cg=True,
                           finalize_chol=False)
    got = gc.rmse(A, B, r["biasA"], r["biasB"], r["glob_mean"], g["e_trow"], g["e_tcol"], g["e_tval"])
    assert abs(got - float(g["rmse"])) < (1e-6 if dtype is np.float64 else 1e-4)
    A, B = g["A0"].copy(), np.zeros((n, k), dtype)
    O.fit_implicit_als(A, B, g["i_row"], g["i_col"], g["i_val"], lam=5.0, niter=15, use_cg=True)
    got = gc.precision_at_k(A, B, g["i_row"], g["i_col"], g["i_trow"], g["i_tcol"], 10)
    assert abs(got - float(g["p_at_10"])) < (1e-4 if dtype is np.float64 else 2e-3)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_new_rows(oracles, dtype):
    """G11: factors of new rows (factors_collective_{explicit,implicit}_multiple): warm rows with and without side
    information, side-information-only rows, empty rows, bias, the lambda scalings and their two quirks."""
    for label, err in gc.new_rows_vs_golden(oracles[dtype], dtype, dict(nthreads=2)):
        assert err < TOL[dtype], (label, err)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sparse_sideinfo(oracles, dtype):
    """G12: fits with sparse side information, the reference's outputs."""
    g = gc.load("g12_sparse_sideinfo", dtype)
    d = gc.sparse_sideinfo_problem(dtype)
    for ci, (name, implicit, which, sl, sls) in enumerate(gc.SPARSE_SIDE_CASES):
        got = gc.sparse_sideinfo_oracle(oracles[dtype], d, implicit, which, sl, sls)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert gc.compare_fits(got, exp) < TOL[dtype], name
    for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.SPARSE_SIDE_CG_CASES):
        got = gc.sparse_sideinfo_oracle(oracles[dtype], d, implicit, which, sl, sls, solver=solver)
        exp = {key[len("g%d_" % ci):]: g[key] for key in g.files if key.startswith("g%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name      # CG fits: the fit tolerance


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nonneg(oracles, dtype):
    """G13: non-negative fits, the reference's outputs."""
    g = gc.load("g13_nonneg", dtype)
    d = gc.nonneg_problem(dtype)
    for ci, (name, implicit, side, opts) in enumerate(gc.NONNEG_CASES):
        got = gc.nonneg_oracle(oracles[dtype], d, implicit, side, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_implicit_features(oracles, dtype):
    """G14: add_implicit_features fits (Ai, Bi and the extra term of the A / B updates), the reference's outputs."""
    g = gc.load("g14_implicit_feats", dtype)
    d = gc.nonneg_problem(dtype)
    for ci, (name, side, opts) in enumerate(gc.IMPLICIT_FEATS_CASES):
        got = gc.implicit_feats_oracle(oracles[dtype], d, side, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and "Ai" in exp and gc.compare_fits(got, exp) < TOL[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lam_unique(oracles, dtype):
    """G15: per-matrix penalties lam_unique / l1_lam_unique in both models, the reference's outputs."""
    g = gc.load("g15_lam_unique", dtype)
    d = gc.nonneg_problem(dtype)
    for ci, (name, implicit, side, opts) in enumerate(gc.LAM_UNIQUE_CASES):
        got = gc.lam_unique_oracle(oracles[dtype], d, implicit, side, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nan_side_info(oracles, dtype):
    """G16: dense U / I with NaN in the reference == the sparse route on the centred present entries."""
    g = gc.load("g16_nan_side", dtype)
    d = gc.nan_side_problem(dtype)
    for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.NAN_SIDE_CASES):
        got = gc.nan_side_oracle(oracles[dtype], d, implicit, which, sl, sls, solver=solver)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_observation_weights(oracles, dtype):
    """G17: fit_collective_explicit_als with weight != NULL (column-sorted entries, see golden_cases.weights_problem) -- the
    cases the oracle restates (no side information, given start values)."""
    g = gc.load("g17_weights", dtype)
    d = gc.weights_problem(dtype)
    seen = 0
    for ci, (name, side, opts) in enumerate(gc.WEIGHT_CASES):
        got = gc.weights_oracle(oracles[dtype], d, side, opts)
        if got is None:
            continue
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name
        seen += 1
    assert seen >= 5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_global_mean_with_eight_threads(oracles, dtype):
    """G23: the centred explicit fit called with nthreads = 8 -- calc_mean_and_center then takes sum / count instead of the
    running mean (common.c:3496-3513) and, with weights, the unweighted sum over the sum of the weights (:3561-3571)."""
    g = gc.load("g23_nthreads8_mean", dtype)
    d = gc.weights_problem(dtype)
    seen = 0
    for ci, (name, weighted, opts) in enumerate(gc.NTHREADS8_CASES):
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        got = gc.nthreads8_oracle(oracles[dtype], d, weighted, opts)
        if got is None:
            continue
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name
        assert abs(float(got["glob_mean"]) - float(exp["glob_mean"])) <= (1e-13 if dtype is np.float64 else 1e-6), name
        seen += 1
    assert seen == 4
    # the weighted branch is NOT the weighted mean: the fixture's number is sum(x) / sum(w)
    w = d["W"].astype(np.float64); x = d["ratings"].astype(np.float64)
    gm8 = float(g["c3_glob_mean"])
    assert abs(gm8 - x.sum() / w.sum()) < (1e-12 if dtype is np.float64 else 1e-5)
    assert abs(gm8 - (x * w).sum() / w.sum()) > 1e-2


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X(oracles, dtype):
    """G18: fit_collective_explicit_als with NA_as_zero_X on sparse X -- the cases with given start values."""
    g = gc.load("g18_na_as_zero", dtype)
    d = gc.naz_problem(dtype)
    seen = 0
    for ci, (name, opts) in enumerate(gc.NAZ_CASES):
        got = gc.naz_oracle(oracles[dtype], d, opts)
        if got is None:
            continue
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name
        seen += 1
    assert seen >= 6


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X_weighted(oracles, dtype):
    """G24: fit_collective_explicit_als with NA_as_zero_X AND observation weights (optimizeA Case 4's NA_as_zero + weight branches,
    common.c:3209-3302; the mean divided by the weights' share of all cells, :3590-3594; wsumA / wsumB counting the absent
    entries, collective.c:8014-8022)."""
    g = gc.load("g24_na_as_zero_weighted", dtype)
    d = gc.naz_weighted_problem(dtype)
    for ci, (name, opts) in enumerate(gc.NAZ_WEIGHTED_CASES):
        got = gc.naz_weighted_oracle(oracles[dtype], d, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X_sparse_sideinfo(oracles, dtype):
    """G25: NA_as_zero_X together with SPARSE side information -- row by row on the shared B^T B plus the rank-1 terms of the row's
    attributes (collective_closed_form_block's general branch with prefer_BtB, collective.c:1534-1846)."""
    g = gc.load("g25_na_as_zero_sparse_side", dtype)
    d = gc.naz_sparse_side_problem(dtype)
    for ci, (name, which, opts) in enumerate(gc.NAZ_SPARSE_SIDE_CASES):
        got = gc.naz_sparse_side_oracle(oracles[dtype], d, which, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X_implicit_features(oracles, dtype):
    """G26: NA_as_zero_X together with implicit features (no side information): every row of a half-step shares
    B^T B + w_i Bi^T Bi + lam mult I; right-hand sides X B + w_i sum_{observed} Bi_j + the bias / mean constant."""
    g = gc.load("g26_na_as_zero_implicit_features", dtype)
    d = gc.naz_problem(dtype)
    for ci, (name, opts) in enumerate(gc.NAZ_IMPF_CASES):
        got = gc.naz_impf_oracle(oracles[dtype], d, opts)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X_sideinfo(oracles, dtype):
    """G20: NA_as_zero_X together with dense side information -- one factorised block matrix per half-step
    (collective.c:5607-5617, :5700-5716), right-hand sides X B + w U C + the bias / mean constant."""
    g = gc.load("g20_na_as_zero_sideinfo", dtype)
    d = gc.naz_problem(dtype)
    seen = 0
    for ci, (name, sides, opts) in enumerate(gc.NAZ_SIDE_CASES):
        got = gc.naz_side_oracle(oracles[dtype], d, sides, opts)
        if got is None:
            continue
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name
        seen += 1
    assert seen >= 8


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_X_weighted_sideinfo(oracles, dtype):
    """G28: NA_as_zero_X with observation weights AND dense side information -- the rows with entries leave the factorised block
    matrix for collective_closed_form_block's general branch (collective.c:1367-1372 -> :1534-1846): (w_j - 1) b_j b_j^T on top of
    the shared matrix, lambda x (sum of the row's weights + its absent entries (+ p))."""
    g = gc.load("g28_na_as_zero_weighted_sideinfo", dtype)
    d = gc.naz_weighted_problem(dtype)
    for ci, (name, sides, opts) in enumerate(gc.NAZ_WEIGHTED_SIDE_CASES):
        got = gc.naz_side_oracle(oracles[dtype], d, sides, opts, weights=True)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_NA_as_zero_UI(oracles, dtype):
    """G21: NA_as_zero_U / NA_as_zero_I (the reference's sparse branches, collective.c:1277-1457, :5790-5836, C / D by optimizeA
    Case 3 with the column means as a rank-one correction) against the restatement: the dense route on the zero-filled
    matrices -- both models, closed form and CG, U on fewer rows than X."""
    g = gc.load("g21_na_as_zero_UI", dtype)
    d = gc.sparse_sideinfo_problem(dtype)
    for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.NAZ_UI_CASES):
        got = gc.naz_ui_oracle(oracles[dtype], d, implicit, which, sl, sls, solver)
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        got = {key: v for key, v in got.items() if key in exp}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_X(oracles, dtype):
    """G19: fit_collective_explicit_als on a dense X with NaN (the reference's optimizeA Cases 1-2) against the restatement run
    on the present entries as a sparse X: closed form in the half-steps whose rows are all / nearly all complete (whatever
    use_cg says), the solver asked for otherwise; rows / columns without a present entry are zero."""
    g = gc.load("g19_dense_X", dtype)
    seen = 0
    for ci, (name, variant, opts) in enumerate(gc.DENSE_CASES):
        got = gc.dense_oracle(oracles[dtype], gc.dense_problem(dtype, variant), variant, opts)
        if got is None:
            continue
        exp = {key[len("c%d_" % ci):]: g[key] for key in g.files if key.startswith("c%d_" % ci)}
        assert exp and gc.compare_fits(got, exp) < TOL_FIT[dtype], name
        seen += 1
    assert seen >= 8
