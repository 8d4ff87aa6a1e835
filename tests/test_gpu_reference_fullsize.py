"""The reference itself in the parity loop at BASELINE.json's full sizes (north_star: "results match the reference CPU solver on the
same inputs ... RMSE / P@10 equal").  The compiled reference (oracle/_ref: cmfrec's own src/*.c, built by oracle/Makefile, shipped
with the repo snapshot) runs on the GPU box's host cores at the thread count bench.py found fastest there (8) -- about a second per
ALS iteration -- beside the HIP path on the same inputs:
  * configuration 2 (358,858 x 160,112, 17.3 M entries, k = 50 fp64): one full iteration operator by operator, then a whole 15-
    iteration fit through the 62-argument entry point on a 95 % split with P@10 of the held-out entries (the reference's
    benchmark_implicit_cmfrec.ipynb, cell 3) from both sets of factors;
  * configuration 1 (69,878 x 10,677, 10 M ratings, k = 50 fp64, biases + centring + scale_lam): a whole 15-iteration fit through
    the 82-argument entry point from the reference's own seeded start, RMSE of the held-out ratings from both.
Tolerances: SURVEY.md 8d -- factors 1e-6 (fp64 whole fit), RMSE 1e-6, P@10 1e-4."""
import multiprocessing
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

K = 50
NTHREADS = max(1, min(8, multiprocessing.cpu_count()))


def _reference():
    from oracle.bindings import Reference, ref_available
    if not ref_available(np.float64):
        pytest.skip("oracle/_ref (the compiled reference) did not travel with this snapshot")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    return Reference(np.float64)


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def c2():
    import bench
    row, col, val = bench.synth_block(bench.M_USERS, bench.N_ITEMS, bench.NNZ, seed=2)
    rng = np.random.default_rng(100)
    A0 = rng.random((bench.M_USERS, K)) * 2.0 ** -7
    return bench.M_USERS, bench.N_ITEMS, row, col, val.astype(np.float64), A0


def test_c2_one_iteration_operator_by_operator(c2):
    """optimizeA_implicit (common.c:3305-3421) B-step then A-step on the whole of configuration 2, reference vs device session."""
    from cmfrec_amd.session import AlsSession
    from oracle.bindings import Oracle
    R = _reference()
    m, n, row, col, val, A0 = c2
    csr, csc = Oracle(np.float64).coo_to_csr_and_csc(row, col, val, m, n)
    A = A0.copy(); B = np.zeros((n, K))
    R.optimizeA_implicit(B, A, csc, 5.0, nthreads=NTHREADS, use_cg=True, max_cg_steps=3)
    R.optimizeA_implicit(A, B, csr, 5.0, nthreads=NTHREADS, use_cg=True, max_cg_steps=3)
    s = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=5.0, use_cg=True, max_cg_steps=3)
    s.set_X_coo(row, col, val)
    s.set_factors(A=A0, B=np.zeros((n, K)))
    s.update("B"); s.update("A")
    f = s.get_factors()
    assert _rel(f["B"], B) < 1e-10 and _rel(f["A"], A) < 1e-10, (_rel(f["A"], A), _rel(f["B"], B))
    # row by row: no row further than 1e-6 from the reference's (a row whose CG left through another exit would show here)
    for got, ref in ((f["A"], A), (f["B"], B)):
        d = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-12)
        assert d.max() < 1e-6, d.max()


def test_c2_whole_fit_and_precision_at_10(c2):
    """fit_collective_implicit_als through the 62-argument ABI, 15 iterations of ALS-CG on a 95 % split of configuration 2 from the
    same start; the held-out 5 % give P@10 as the reference's benchmark computes it (a sample of 2,000 users with held-out items:
    ranking all 160,112 items for every user on the host is the slow part)."""
    from cmfrec_amd import CMF_implicit
    R = _reference()
    m, n, row, col, val, A0 = c2
    rng = np.random.default_rng(11)
    test = rng.random(len(row)) < 0.05
    tr, te = ~test, test
    Ar = A0.copy(); Br = np.zeros((n, K))
    r = R.fit_collective_implicit_als(Ar, Br, row[tr], col[tr], val[tr], K, lam=5.0, alpha=1.0, niter=15, nthreads=NTHREADS,
                                      use_cg=True, max_cg_steps=3, finalize_chol=False, m=m, n=n)
    assert r == 0                       # (plain model: the binding returns the code, the factors are updated in place)
    r = dict(A=Ar, B=Br)
    mdl = CMF_implicit(k=K, lambda_=5.0, niter=15, use_float=False, use_cg=True, finalize_chol=False, precompute_for_predictions=False,
                       nthreads=NTHREADS).fit((row[tr], col[tr], val[tr]), shape=(m, n), A0=A0, B0=np.zeros((n, K)))
    eA, eB = _rel(mdl.A_, r["A"]), _rel(mdl.B_, r["B"])
    assert eA < 1e-6 and eB < 1e-6, (eA, eB)
    users = np.unique(row[te])
    users = rng.choice(users, 2000, replace=False)
    keep_te = np.isin(row[te], users); keep_tr = np.isin(row[tr], users)
    args = (row[tr][keep_tr], col[tr][keep_tr], row[te][keep_te], col[te][keep_te])
    p_ref = gc.precision_at_k(r["A"], r["B"], *args)
    p_hip = gc.precision_at_k(mdl.A_, mdl.B_, *args)
    print("C2 15 iterations: rel. Frobenius A %.2e B %.2e; P@10 reference %.6f, HIP %.6f" % (eA, eB, p_ref, p_hip))
    chance = 10.0 * keep_te.sum() / len(users) / n            # ten random items against a user's held-out ones
    assert p_ref > 20 * chance, "the fit must rank held-out items well above chance"
    assert abs(p_ref - p_hip) <= 1e-4


def test_c1_whole_fit_and_rmse():
    """fit_collective_explicit_als through the 82-argument ABI on configuration 1's full shape: the reference's own seeded start
    (random_parallel + two-sided bias start values), centring, both biases, scale_lam, 15 iterations of ALS-CG; RMSE of a held-out
    5 % (benchmark_explicit_cmfrec.ipynb) from both sets of factors."""
    import bench
    from cmfrec_amd import CMF
    R = _reference()
    m, n, nnz = 69_878, 10_677, 10_000_054
    row, col, _ = bench.synth_block(m, n, nnz, seed=1)
    rng = np.random.default_rng(1)
    val = 0.5 * rng.integers(1, 11, nnz).astype(np.float64)
    test = rng.random(nnz) < 0.05
    tr, te = ~test, test
    Ar = np.zeros((m, K)); Br = np.zeros((n, K))
    r = R.fit_collective_explicit_als(Ar, Br, row[tr], col[tr], val[tr], K, lam=0.05, scale_lam=True, niter=15, nthreads=NTHREADS,
                                      use_cg=True, max_cg_steps=3, finalize_chol=False, reset_values=True, seed=1, m=m, n=n)
    assert r["ret"] == 0
    mdl = CMF(k=K, lambda_=0.05, scale_lam=True, niter=15, use_cg=True, finalize_chol=False, use_float=False,
              precompute_for_predictions=False, random_state=1, nthreads=NTHREADS).fit((row[tr], col[tr], val[tr]), shape=(m, n))
    assert abs(float(mdl.glob_mean_) - float(r["glob_mean"])) < 1e-12
    errs = dict(A=_rel(mdl.A_, r["A"]), B=_rel(mdl.B_, r["B"]), biasA=_rel(mdl.user_bias_, r["biasA"]), biasB=_rel(mdl.item_bias_, r["biasB"]))
    assert max(errs.values()) < 1e-6, errs
    rm_ref = gc.rmse(r["A"], r["B"], r["biasA"], r["biasB"], r["glob_mean"], row[te], col[te], val[te])
    rm_hip = gc.rmse(mdl.A_, mdl.B_, mdl.user_bias_, mdl.item_bias_, mdl.glob_mean_, row[te], col[te], val[te])
    print("C1 15 iterations: %s; RMSE reference %.8f, HIP %.8f" % (errs, rm_ref, rm_hip))
    assert abs(rm_ref - rm_hip) <= 1e-6


def test_c3_one_iteration_vs_reference():
    """BASELINE configuration 3 at its full shape -- 69,878 x 10,677, 10 M ratings, k = 128 fp64, closed form (Cholesky), a 64-column
    dense item side information, both biases, centring, scale_lam -- one ALS iteration (the reference's order C / D, B, A:
    collective.c:8334-8898) through the 82-argument entry point from the reference's own seeded start, against the compiled
    reference on the same inputs: A, B, D and both biases to 1e-10 (Frobenius), every row of A and B to 1e-6."""
    import bench
    from cmfrec_amd import CMF
    R = _reference()
    m, n, nnz, k, q = 69_878, 10_677, 10_000_054, 128, 64
    row, col, _ = bench.synth_block(m, n, nnz, seed=1)
    rng = np.random.default_rng(1)
    val = 0.5 * rng.integers(1, 11, nnz).astype(np.float64)
    II = rng.standard_normal((n, q))
    Ar = np.zeros((m, k)); Br = np.zeros((n, k))
    r = R.fit_collective_explicit_als(Ar, Br, row, col, val, k, II=II.copy(), lam=0.05, scale_lam=True, niter=1, nthreads=NTHREADS,
                                      use_cg=False, finalize_chol=False, reset_values=True, seed=1, m=m, n=n)
    assert r["ret"] == 0
    mdl = CMF(k=k, lambda_=0.05, scale_lam=True, niter=1, use_cg=False, finalize_chol=False, use_float=False,
              precompute_for_predictions=False, random_state=1, nthreads=NTHREADS).fit((row, col, val), I=II.copy(), shape=(m, n))
    assert abs(float(mdl.glob_mean_) - float(r["glob_mean"])) < 1e-12
    errs = dict(A=_rel(mdl.A_, r["A"]), B=_rel(mdl.B_, r["B"]), D=_rel(mdl.D_, r["D"]), biasA=_rel(mdl.user_bias_, r["biasA"]),
                biasB=_rel(mdl.item_bias_, r["biasB"]))
    print("C3 one iteration vs the reference:", errs)
    assert max(errs.values()) < 1e-10, errs
    for got, ref in ((mdl.A_, r["A"]), (mdl.B_, r["B"])):
        d = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-12)
        assert d.max() < 1e-6, d.max()


def test_c4_shard_one_iteration_vs_reference_float():
    """One rank's share of BASELINE configuration 4 at N = 8 (1.25 M users x the 1 M-item replica, 62.5 M entries, k = 64) in SINGLE
    precision: optimizeA_implicit (common.c:3305-3421) B-step then A-step, the compiled single-precision reference
    (oracle/_ref/libcmfrec_ref_float.so) against the device session on the same inputs.
    At this size the reference's own single-precision arithmetic is the larger error: it adds the rank-1 terms of an item with
    tens of thousands of entries one after the other in float (and its B^T B of a million rows comes out of a float syrk), the
    device sums tiles and slices pairwise.  So the yardstick is the reference's DOUBLE-precision run of the same iteration: the
    device must be as close to it as the tolerance SURVEY 8d states for single precision (1e-4 Frobenius, 1e-3 per row) or as
    close as the single-precision reference itself is (per row: twice its 99.9 % quantile, ten times its worst row), whichever is larger -- and it must not be further from the single-precision
    reference than the two distances to the double-precision run add up to."""
    import bench
    from cmfrec_amd.session import AlsSession
    from oracle.bindings import Oracle, Reference, ref_available
    if not (ref_available(np.float32) and ref_available(np.float64)):
        pytest.skip("oracle/_ref (the compiled reference, both precisions) did not travel with this snapshot")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    m, n, nnz, k = 1_250_000, 1_000_000, 62_500_000, 64
    row, col, val = bench.synth_block(m, n, nnz, seed=4)
    val = val.astype(np.float32)
    rng = np.random.default_rng(7)
    A0 = rng.random((m, k), dtype=np.float32) * np.float32(2.0 ** -7)
    out = {}
    for dt in (np.float32, np.float64):
        R = Reference(dt)
        csr, csc = Oracle(dt).coo_to_csr_and_csc(row, col, val.astype(dt), m, n)
        A = A0.astype(dt); B = np.zeros((n, k), dt)
        R.optimizeA_implicit(B, A, csc, 5.0, nthreads=NTHREADS, use_cg=True, max_cg_steps=3)
        R.optimizeA_implicit(A, B, csr, 5.0, nthreads=NTHREADS, use_cg=True, max_cg_steps=3)
        out[dt] = (A, B)
        del csr, csc
    s = AlsSession(m, n, k, implicit=True, dtype=np.float32, lam=5.0, use_cg=True, max_cg_steps=3)
    s.set_X_coo(row, col, val)
    s.set_factors(A=A0, B=np.zeros((n, k), np.float32))
    s.update("B"); s.update("A")
    f = s.get_factors()
    (A32, B32), (A64, B64) = out[np.float32], out[np.float64]
    rows = lambda x, y: np.abs(x.astype(np.float64) - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-6)
    for name, got, r32, r64 in (("B", f["B"], B32, B64), ("A", f["A"], A32, A64)):
        e_hip, e_ref, e_pair = _rel(got, r64), _rel(r32, r64), _rel(got, r32.astype(np.float64))
        d_hip, d_ref = rows(got, r64), rows(r32, r64)
        # per row: all but one row in a thousand, and the worst row (an ill-conditioned row of a few entries: its error is some
        # percent in EITHER single-precision run and moves with the order of the sums, i.e. with the box's BLAS threads)
        q_hip, q_ref, w_hip, w_ref = float(np.quantile(d_hip, 0.999)), float(np.quantile(d_ref, 0.999)), float(d_hip.max()), float(d_ref.max())
        print("c4 shard, one iteration, %s: device vs double reference %.2e (rows: 99.9 %% below %.2e, worst %.2e); single reference vs double "
              "reference %.2e (rows: %.2e, worst %.2e); device vs single reference %.2e" % (name, e_hip, q_hip, w_hip, e_ref, q_ref, w_ref, e_pair))
        assert e_hip <= max(1e-4, e_ref), (name, e_hip, e_ref)
        assert q_hip <= max(1e-3, 2.0 * q_ref), (name, q_hip, q_ref)
        assert w_hip <= max(1e-3, 10.0 * w_ref), (name, w_hip, w_ref)
        assert e_pair <= e_hip + e_ref + 1e-7, (name, e_pair)
