"""GPU: the HIP path (through the C ABI) against the golden vectors captured from the real
reference.  Same tolerances as the oracle's own golden test."""
import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu
DT = [np.float64, np.float32]
TOL = {np.float64: 1e-10, np.float32: 2e-4}
TOL_FIT = {np.float64: 1e-6, np.float32: 1e-2}


@pytest.mark.parametrize("dtype", DT)
def test_operators(oracles, dtype):
    from cmfrec_amd import ops
    O = oracles[dtype]   # only used for the (exact) COO -> CSR conversion of the stored triplets
    for name, got, exp in gc.implicit_cases(gc.load("g1_implicit", dtype), O, ops.optimizeA_implicit):
        assert gc.maxrel(got, exp) < TOL[dtype], name
    for name, got, exp in gc.explicit_cases(gc.load("g2_explicit", dtype), O, ops.optimizeA_explicit):
        assert gc.maxrel(got, exp) < TOL[dtype], name
    for name, got, exp in gc.collective_cases(gc.load("g3_collective", dtype), O, ops.optimizeA_collective):
        assert gc.maxrel(got, exp) < TOL[dtype], name
    g = gc.load("g4_dense_full", dtype)
    C = np.zeros_like(g["C"])
    ops.optimizeA_dense_full(C, g["A_bias"], g["U"], float(g["lam"]), k=int(g["kc"]), do_B=True, scale_lam=True)
    assert gc.maxrel(C, g["C"]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_fits(dtype):
    from cmfrec_amd import CMF, CMF_implicit
    t = TOL_FIT[dtype]
    uf = dtype is np.float32
    g = gc.load("g5_fit_implicit", dtype)
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    X = (g["row"], g["col"], g["val"])
    for mode in ("cg", "chol", "cgfin"):
        mdl = CMF_implicit(k=k, lambda_=float(g["lam"]), alpha=float(g["alpha"]), niter=int(g["niter"]),
                           use_cg=mode != "chol", finalize_chol=mode == "cgfin", use_float=uf).fit(
            X, shape=(m, n), A0=g["A0"], B0=np.zeros((n, k), dtype))
        assert gc.frob(mdl.A_, g["A_" + mode]) < t and gc.frob(mdl.B_, g["B_" + mode]) < t, mode
    g = gc.load("g5_fit_explicit", dtype)
    X = (g["row"], g["col"], g["val"])
    for mode in ("cg", "chol", "cgfin"):
        mdl = CMF(k=k, lambda_=float(g["lam"]), scale_lam=True, niter=int(g["niter"]), use_cg=mode != "chol",
                  finalize_chol=mode == "cgfin", use_float=uf, nthreads=1).fit(
            X, shape=(m, n), A0=g["A0"], B0=np.zeros((n, k), dtype), biasA0=g["biasA0"], biasB0=g["biasB0"])
        assert gc.frob(mdl.A_, g["A_" + mode]) < t and gc.frob(mdl.B_, g["B_" + mode]) < t, mode
        assert gc.frob(mdl.user_bias_, g["biasA_" + mode]) < t and gc.frob(mdl.item_bias_, g["biasB_" + mode]) < t
        assert abs(mdl.glob_mean_ - float(g["glob_mean"])) <= 1e-6 * abs(float(g["glob_mean"]))
    g = gc.load("g5_fit_sideinfo", dtype)
    ku, ki, km = [int(x) for x in g["cfg"]]
    mdl = CMF(k=k, lambda_=0.05, scale_lam=True, scale_lam_sideinfo=True, niter=3, use_cg=False, k_user=ku, k_item=ki,
              k_main=km, w_user=0.5, w_item=2.0, use_float=uf, nthreads=1).fit(
        (g["row"], g["col"], g["val"]), shape=(m, n), U=g["U"], I=g["II"], A0=g["A0"], B0=g["B0"])
    for got, key in ((mdl.A_, "A"), (mdl.B_, "B"), (mdl.C_, "C"), (mdl.D_, "D"), (mdl.user_bias_, "biasA"),
                     (mdl.item_bias_, "biasB")):
        assert gc.frob(got, g[key]) < t, key
    g = gc.load("g8_fit_implicit_sideinfo", dtype)
    ku, ki, km = [int(x) for x in g["cfg"]]
    mdl = CMF_implicit(k=k, lambda_=3.0, alpha=2.0, niter=3, use_cg=False, k_user=ku, k_item=ki, k_main=km, w_main=0.5,
                       w_user=4.0, w_item=0.8, use_float=uf).fit(
        (g["row"], g["col"], g["val"]), shape=(m, n), U=g["U"], I=g["II"], A0=g["A0"], B0=g["B0"])
    for got, key in ((mdl.A_, "A"), (mdl.B_, "B"), (mdl.C_, "C"), (mdl.D_, "D")):
        assert gc.frob(got, g[key]) < t, key
    assert gc.maxrel(mdl._U_colmeans, g["U_colmeans"]) < 1e-6
