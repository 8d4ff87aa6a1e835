"""GPU parity at the kernel instantiations of BASELINE.json's configs ("configs_untested"): the row
kernels are templated on the width of the system, so a test at k=14 does not exercise what C3 / C4 / C5 launch.

  C3: explicit Cholesky, k = 128 + bias (k_t = 129), dense item side information q = 64, double precision
  C4: implicit CG, k = 64, single precision
  C5: explicit Cholesky, k = 256 + bias (k_t = 257), p = q = 512 dense side information on both sides, single precision

Each case runs the operator through the C ABI on ~50-200 rows -- among them a heavy row (beyond the 1024-entry split
between the wave-per-row and the workgroup-per-row Cholesky kernels), rows with a handful of entries (C5's users: 20
entries against 257 unknowns) and empty rows -- and compares with the oracle on the same inputs.
Reference: optimizeA_collective /root/reference/src/collective.c:5566-5968, collective_closed_form_block :1534-1846,
optimizeA_implicit /root/reference/src/common.c:3305-3421.
Tolerances: SURVEY.md 8d -- double 1e-10; single 2e-4 measured worst case over these cases stays below 1e-4 for the
CG operator and below 2e-4 for the 257-wide Cholesky systems (condition number ~1e3 at lambda 0.05 x nnz)."""
import numpy as np
import pytest

from conftest import make_coo, rel_err

pytestmark = pytest.mark.gpu


def _collective_case(O, dtype, m, n, k, p, nnz, seed, heavy, w, ragged=True):
    """A-side collective system: users x items with user side information U [m, p] and a bias column."""
    row, col, val = make_coo(m, n, nnz, seed, counts=False, dtype=dtype, heavy_row=heavy, empty_rows=(4, m - 2))
    if ragged:       # a few users with exactly 1, 2, 3 entries
        keep = np.ones(len(row), bool)
        for r, cnt in ((6, 1), (7, 2), (8, 3)):
            idx = np.flatnonzero(row == r)
            keep[idx[cnt:]] = False
        row, col, val = row[keep], col[keep], val[keep]
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(seed + 1)
    Bm = (rng.standard_normal((n, k + 1)) * 0.3).astype(dtype)
    Bm[:, k] = 1                                      # the opposing bias column is fixed to 1 (collective.c:8538-8543)
    Cm = (rng.standard_normal((p, k)) * (0.3 / np.sqrt(p / 12.0))).astype(dtype)
    U = rng.standard_normal((m, p)).astype(dtype)
    A0 = rng.standard_normal((m, k + 1)).astype(dtype)
    bias = (rng.standard_normal(n) * 0.2).astype(dtype)
    return csr, Bm, Cm, U, A0, bias


@pytest.mark.parametrize("chol_wg", ["4", "2", "0"])
@pytest.mark.parametrize("scale_lam", [True, False])
def test_c3_width_collective_double(oracles, scale_lam, chol_wg, monkeypatch):
    """k = 128 + bias, q = 64, fp64: the rank-k producer + the factorisation of the eight-block rows with their border column --
    by a workgroup of four wavefronts per row (the default), of two (CMFREC_HIP_CHOL_WG=2), by one wavefront per row (=0: the
    kernel of rounds 2-5) -- and, for the heavy row, its slices' partials."""
    from cmfrec_amd import ops
    monkeypatch.setenv("CMFREC_HIP_CHOL_WG", chol_wg)
    dtype = np.float64
    O = oracles[dtype]
    m, n, k, p = 56, 2400, 128, 64
    csr, Bm, Cm, U, A0, bias = _collective_case(O, dtype, m, n, k, p, 9000, 31, (3, 1500), 0.5)
    assert np.diff(csr[0].astype(np.int64)).max() > 1024
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(w_user=0.5, lam_last=0.2, k=k, scale_lam=scale_lam)
    ops.optimizeA_collective(Ah, Bm, Cm, csr, U, 0.05, bias_sub=bias, **kw)
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O.optimizeA_collective_chol(Ao, Bm, Cm, csr_sub, U, 0.05, nthreads=4, **kw)
    assert rel_err(Ah, Ao) < 1e-10
    # rows without entries are still solved from their side information
    assert np.abs(Ah[4]).max() > 0


@pytest.mark.parametrize("chol_wg", ["4", "2", "0"])
@pytest.mark.parametrize("k", [128, 127, 120])
def test_c3_width_explicit_double(oracles, k, chol_wg, monkeypatch):
    """The user side of C3 has no side information: plain explicit Cholesky at k_t = 129 (bias as the border column), 128 (no
    border: the bias inside the tiles) and 121 (padding inside the last block), on each of the three factorisation kernels."""
    from cmfrec_amd import ops
    monkeypatch.setenv("CMFREC_HIP_CHOL_WG", chol_wg)
    dtype = np.float64
    O = oracles[dtype]
    m, n = 64, 2000
    row, col, val = make_coo(m, n, 9000, 5, counts=False, dtype=dtype, heavy_row=(9, 1300), empty_rows=(2,))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(6)
    A0 = (rng.standard_normal((m, k + 1)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k + 1)) * 0.2).astype(dtype)
    B[:, k] = 1
    bias = (rng.standard_normal(n) * 0.3).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(k=k + 1, lam_last=0.3, scale_lam=True, use_cg=False)
    ops.optimizeA_explicit(Ah, B, csr, 0.05, bias_sub=bias, **kw)
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O.optimizeA_explicit(Ao, B, csr_sub, 0.05, nthreads=4, **kw)
    assert rel_err(Ah, Ao) < 1e-10
    assert np.array_equal(Ah[2], A0[2])


@pytest.mark.parametrize("k", [128, 127])
def test_c3_width_shared_gather_is_the_per_wavefront_gather(oracles, k, monkeypatch):
    """The rank-k update of the eight-block rows with ONE gather shared by the row's two wavefronts through LDS
    (chol_parts_coop_kernels.hpp, the default since round 6) against each wavefront gathering for itself (CMFREC_HIP_PARTS_COOP=0,
    round 5): the sums are taken in the same order, so the factors are bit for bit the same -- rows of one to 1300 entries (steps that
    are not a multiple of the registers' depth, a slice boundary, entries that are not a multiple of four), with and without the
    border column; and both agree with the oracle."""
    from cmfrec_amd import ops
    dtype = np.float64
    O = oracles[dtype]
    m, n = 72, 2000
    row, col, val = make_coo(m, n, 9000, 15, counts=False, dtype=dtype, heavy_row=(9, 1300), empty_rows=(2,))
    keep = np.ones(len(row), bool)
    for r, cnt in ((6, 1), (7, 2), (8, 3), (10, 5), (11, 17), (12, 99), (13, 100), (14, 101)):
        idx = np.flatnonzero(row == r)
        keep[idx[cnt:]] = False
    row, col, val = row[keep], col[keep], val[keep]
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(16)
    A0 = (rng.standard_normal((m, k + 1)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k + 1)) * 0.2).astype(dtype)
    B[:, k] = 1
    bias = (rng.standard_normal(n) * 0.3).astype(dtype)
    kw = dict(k=k + 1, lam_last=0.3, scale_lam=True, use_cg=False)
    out = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("CMFREC_HIP_PARTS_COOP", coop)
        monkeypatch.setenv("CMFREC_HIP_LOWRANK", "0")          # every row through the producer / factorisation pair
        Ah = A0.copy()
        ops.optimizeA_explicit(Ah, B, csr, 0.05, bias_sub=bias, **kw)
        out[coop] = Ah
    assert np.array_equal(out["1"], out["0"])
    Ao = A0.copy()
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O.optimizeA_explicit(Ao, B, csr_sub, 0.05, nthreads=4, **kw)
    assert rel_err(out["1"], Ao) < 1e-10


def test_c3_width_shared_gather_implicit_model(oracles, monkeypatch):
    """The same kernel with the implicit model's weights (k = 128, closed form: every entry's matrix weight is its confidence -- the
    build that keeps the per-step weights and selects) against the per-wavefront gather, bit for bit, and against the oracle."""
    from cmfrec_amd import ops
    dtype = np.float64
    O = oracles[dtype]
    m, n, k = 60, 1800, 128
    row, col, val = make_coo(m, n, 8000, 25, counts=True, dtype=dtype, heavy_row=(5, 1250), empty_rows=(3,))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(26)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    out = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("CMFREC_HIP_PARTS_COOP", coop)
        monkeypatch.setenv("CMFREC_HIP_LOWRANK", "0")
        Ah = A0.copy()
        ops.optimizeA_implicit(Ah, B, csr, 4.0, use_cg=False)
        out[coop] = Ah
    assert np.array_equal(out["1"], out["0"])
    Ao = A0.copy()
    O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4, use_cg=False)
    assert rel_err(out["1"], Ao) < 1e-9


@pytest.mark.parametrize("side", ["users", "items"])
@pytest.mark.parametrize("gramk", ["default", "off", "batch7"])
def test_c5_width_collective_single(oracles, side, gramk, monkeypatch):
    """k = 256 + bias, 512-dimensional side information, fp32.  users: ~20 entries per row (nnz << k_t);
    items: hundreds to thousands of entries per row.  gramk: the rank-k update by the four-wavefront producer kernel with
    the partial matrices through HBM (default; split rows arrive as several partials), by the row kernel's own LDS-staged
    loop (off), and the producer in batches of 7 work items."""
    from cmfrec_amd import ops
    if gramk == "off":
        monkeypatch.setenv("CMFREC_HIP_GRAMK", "0")
    elif gramk == "batch7":
        monkeypatch.setenv("CMFREC_HIP_GRAMK_BATCH", "7")
    dtype = np.float32
    O = oracles[dtype]
    k, p = 256, 512
    if side == "users":
        m, n, nnz, heavy = 96, 3000, 96 * 20, (3, 1200)
    else:
        m, n, nnz, heavy = 40, 4000, 40 * 400, (5, 2500)
    csr, Bm, Cm, U, A0, bias = _collective_case(O, dtype, m, n, k, p, nnz, 77, heavy, 0.5)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(w_user=0.5, lam_last=0.1, k=k, scale_lam=True)
    ops.optimizeA_collective(Ah, Bm, Cm, csr, U, 0.05, bias_sub=bias, **kw)
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O.optimizeA_collective_chol(Ao, Bm, Cm, csr_sub, U, 0.05, nthreads=4, **kw)
    # the oracle itself is single precision here: compare both against a double-precision solve of the same systems
    O64 = oracles[np.float64]
    A64 = A0.astype(np.float64)
    O64.optimizeA_collective_chol(A64, Bm.astype(np.float64), Cm.astype(np.float64),
                                  (csr_sub[0], csr_sub[1], csr_sub[2].astype(np.float64)), U.astype(np.float64), 0.05,
                                  nthreads=4, **kw)
    e_hip, e_orc = rel_err(Ah, A64), rel_err(Ao, A64)
    assert e_hip < 2e-4, (e_hip, e_orc)
    assert e_hip < 4 * e_orc + 2e-5, (e_hip, e_orc)   # no worse than the reference's own single-precision arithmetic
    assert rel_err(Ah, Ao) < 2e-4


@pytest.mark.parametrize("mode", ["cg", "chol"])
def test_c4_width_implicit_single(oracles, mode):
    """k = 64 fp32 implicit: CG (C4's solver) and Cholesky (finalize_chol)."""
    from cmfrec_amd import ops
    dtype = np.float32
    O = oracles[dtype]
    m, n, k = 300, 5000, 64
    row, col, val = make_coo(m, n, 16000, 13, dtype=dtype, heavy_row=(3, 3000), empty_rows=(5, 17))
    # rows of 513..1024 entries: in single precision the 8-wave team keeps two tiles per wave (one gather for all passes);
    # 1024 fills both tiles of every wave, 577 leaves the second tile of the first wave with one entry
    rng0 = np.random.default_rng(99)
    keep = ~np.isin(row, (20, 21, 22))
    er, ec = [], []
    for r, cnt in ((20, 1024), (21, 577), (22, 700)):
        er.append(np.full(cnt, r, np.int32)); ec.append(rng0.choice(n, cnt, replace=False).astype(np.int32))
    ev = np.ceil(rng0.lognormal(1, 1, sum(len(e) for e in er))).astype(dtype)
    row = np.concatenate([row[keep]] + er); col = np.concatenate([col[keep]] + ec); val = np.concatenate([val[keep], ev])
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    lens = np.diff(csr[0].astype(np.int64))
    assert lens[20] == 1024 and lens[21] == 577
    rng = np.random.default_rng(k)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(use_cg=mode == "cg", max_cg_steps=3)
    ops.optimizeA_implicit(Ah, B, csr, 4.0, **kw)
    O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4, **kw)
    assert rel_err(Ah, Ao) < 1e-4


@pytest.mark.parametrize("dtype,k,p,ku", [(np.float32, 256, 512, 0), (np.float32, 100, 40, 3), (np.float64, 128, 64, 0),
                                          (np.float64, 72, 24, 2)])
@pytest.mark.parametrize("scale_lam", [True, False])
def test_lowrank_rows(oracles, dtype, k, p, ku, scale_lam, monkeypatch):
    """Rows with few entries against many unknowns take the low-rank path (lowrank_kernels.hpp: the shared matrix
    w C^T C diagonalised once, an s x s system per row); CMFREC_HIP_LOWRANK=1 forces it on this small problem.  Row r
    has r entries (0 .. 159): every block count of the s x s kernel, the hand-over to the full factorisation, empty
    rows.  Same system as the reference's (collective.c:1534-1846), different arithmetic: tolerance-based."""
    from cmfrec_amd import ops
    monkeypatch.setenv("CMFREC_HIP_LOWRANK", "1")
    O = oracles[dtype]
    m, n = 160, 3000
    rng = np.random.default_rng(k + p)
    rows, cols = [], []
    for r in range(m):
        rows.append(np.full(r, r, np.int32)); cols.append(rng.choice(n, r, replace=False).astype(np.int32))
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    # one k_main factor stands in for the bias column: the last unknown has its own lambda and lies outside the block that
    # the side information couples (what fit() launches with a user bias)
    kA = ku + k + 1
    Bm = (rng.standard_normal((n, k + 1)) * 0.3).astype(dtype); Bm[:, k] = 1
    Cm = (rng.standard_normal((p, ku + k)) * (0.3 / np.sqrt(p / 12.0))).astype(dtype)
    U = rng.standard_normal((m, p)).astype(dtype)
    A0 = rng.standard_normal((m, kA)).astype(dtype)
    bias = (rng.standard_normal(n) * 0.2).astype(dtype)
    Ah, A64 = A0.copy(), A0.astype(np.float64)
    kw = dict(w_user=0.5, lam_last=0.1, k=k, k_user=ku, k_main=1, scale_lam=scale_lam)
    ops.optimizeA_collective(Ah, Bm, Cm, csr, U, 0.05, bias_sub=bias, **kw)
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(np.float64))
    oracles[np.float64].optimizeA_collective_chol(A64, Bm.astype(np.float64), Cm.astype(np.float64), csr_sub, U.astype(np.float64),
                                                  0.05, nthreads=4, **kw)
    tol = 1e-9 if dtype is np.float64 else 2e-4
    assert rel_err(Ah, A64) < tol


@pytest.mark.parametrize("dtype,k", [(np.float64, 129), (np.float64, 100), (np.float32, 257), (np.float32, 129), (np.float64, 48)])
@pytest.mark.parametrize("scale_lam", [True, False])
def test_plain_lowrank_rows(oracles, dtype, k, scale_lam, monkeypatch):
    """Plain closed-form rows (no side information) with few entries against many unknowns take the low-rank kernel without any
    rotation (round 4: lam_i I + a rank-s update; config 3's users); CMFREC_HIP_LOWRANK=1 forces it on this small problem.  Row r has
    r entries (0 .. 159): every block count of the s x s kernel, the hand-over to the full factorisation, a row without entries
    (left as it is), the fused bias subtraction, the last unknown's own lambda.  Same system as the reference's
    (factors_closed_form, common.c:978-1070), different arithmetic: tolerance-based; and against the full factorisation of every
    row (CMFREC_HIP_LOWRANK=0)."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n = 160, 3000
    rng = np.random.default_rng(k)
    rows, cols = [], []
    for r in range(m):
        rows.append(np.full(r, r, np.int32)); cols.append(rng.choice(n, r, replace=False).astype(np.int32))
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    B = (rng.standard_normal((n, k)) * 0.3).astype(dtype); B[:, k - 1] = 1
    bias = (rng.standard_normal(n) * 0.2).astype(dtype)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    kw = dict(lam_last=0.2, scale_lam=scale_lam, use_cg=False)
    Ah, Af, Ao = A0.copy(), A0.copy(), A0.copy()
    monkeypatch.setenv("CMFREC_HIP_LOWRANK", "1")
    ops.optimizeA_explicit(Ah, B, csr, 0.4, bias_sub=bias, **kw)
    monkeypatch.setenv("CMFREC_HIP_LOWRANK", "0")
    ops.optimizeA_explicit(Af, B, csr, 0.4, bias_sub=bias, **kw)
    csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O64 = oracles[np.float64]
    Ao = A0.astype(np.float64)
    O64.optimizeA_explicit(Ao, B.astype(np.float64), (csr[0], csr[1], csr_b[2].astype(np.float64)), 0.4, nthreads=4, **kw)
    tol = 1e-9 if dtype is np.float64 else 2e-4
    assert rel_err(Ah, Ao) < tol and rel_err(Af, Ao) < tol
    assert not np.array_equal(Ah, Af)                       # the two paths are different arithmetic ...
    assert np.array_equal(Ah[0], A0[0])                     # ... and a row without entries is left alone by both
    lens = np.diff(csr[0].astype(np.int64))
    big = lens > (128 if (dtype is np.float32 and k >= 256) else (96 if dtype is np.float64 else 64) if k >= 128 else 32)
    assert np.array_equal(Ah[big], Af[big])                 # rows beyond the low-rank limit take the same factorisation
