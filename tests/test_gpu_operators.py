"""GPU parity: every operator of the HIP path (through the C ABI, host-buffer level) against the
oracle on the same seeded inputs.  Tolerances: SURVEY.md 8d -- single operator call, fp64 1e-10
relative (max-abs / max-abs), fp32 1e-4 (worst case measured over this file: 2.7e-5, profiles/r02_ao_relerr_maxima.txt)."""
import numpy as np
import pytest

from conftest import make_coo, rel_err, row_rel_err

pytestmark = pytest.mark.gpu
TOL = {np.float64: 1e-10, np.float32: 1e-4}      # measured worst cases: profiles/r02_ao_relerr_maxima.txt
ROW_TOL = {np.float64: 1e-9, np.float32: 1e-3}   # per-row criterion (row max-abs error / row max-abs), SURVEY.md 8d


def check_rows(got, exp, dtype):
    e, r = row_rel_err(got, exp)
    assert e < ROW_TOL[dtype], "row %d: per-row relative error %.3e" % (r, e)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k", [8, 50, 64])
@pytest.mark.parametrize("mode", ["cg", "chol", "pcg"])
def test_optimizeA_implicit(oracles, dtype, k, mode):
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n = 700, 450
    row, col, val = make_coo(m, n, 20000, 11 + k, dtype=dtype, heavy_row=(3, 400), empty_rows=(5, 17))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(k)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
    Gh = ops.optimizeA_implicit(Ah, B, csr, 4.0, return_BtB=True, **kw)
    Go = O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4, return_BtB=True, **kw)
    assert rel_err(Gh, Go) < TOL[dtype]
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah, Ao, dtype)
    if mode != "chol":   # empty rows are left untouched by the CG path (common.c:3354)
        assert np.array_equal(Ah[5], A0[5]) and np.array_equal(Ah[17], A0[17])
    else:              # and zeroed by the Cholesky path (common.c:3334)
        assert not Ah[5].any()


@pytest.mark.parametrize("k", [49, 50, 51, 52, 53])
def test_gramian_last_block_widths(oracles, k):
    """Double precision, 48 < k <= 52: the Gramian kernel takes the last column block's live columns through the vector ALU
    (gram_mfma_partial_kernel<double, REM>, REM = 1 .. 4), k = 53 is back on four matrix-core blocks; several row blocks,
    both triangles of the result, and the half-step that uses it."""
    from cmfrec_amd import ops
    dtype = np.float64
    O = oracles[dtype]
    m, n = 300, 5000
    row, col, val = make_coo(m, n, 30000, 70 + k, dtype=dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(k)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    Gh = ops.optimizeA_implicit(Ah, B, csr, 4.0, return_BtB=True, use_cg=True, max_cg_steps=3)
    Go = O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4, return_BtB=True, use_cg=True, max_cg_steps=3)
    assert np.array_equal(Gh, Gh.T)
    assert rel_err(Gh, Go) < 1e-13
    dg = np.sqrt(np.diag(Go))
    assert np.max(np.abs(Gh - Go) / np.outer(dg, dg)) < 1e-13    # every entry on the scale of its two columns, the last block's too
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah, Ao, dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,pad", [(51, 1), (16, 0), (33, 2)])
@pytest.mark.parametrize("mode", ["cg", "chol", "pcg"])
def test_optimizeA_explicit(oracles, dtype, k, pad, mode):
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n = 600, 380
    row, col, val = make_coo(m, n, 15000, 5 + k, counts=False, dtype=dtype, heavy_row=(7, 300), empty_rows=(2,))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(k)
    A0 = (rng.standard_normal((m, k + pad)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k + 1)) * 0.2).astype(dtype)
    bias = (rng.standard_normal(n) * 0.3).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(k=k, lam_last=0.3, scale_lam=True, use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
    ops.optimizeA_explicit(Ah, B, csr, 0.05, bias_sub=bias, **kw)
    csr_sub = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))   # what the reference's host sweep does
    O.optimizeA_explicit(Ao, B, csr_sub, 0.05, nthreads=4, **kw)
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah[:, :k], Ao[:, :k], dtype)
    assert np.array_equal(Ah[2], A0[2])            # empty rows untouched in Case 4 (common.c:3270)
    if pad:
        assert np.array_equal(Ah[:, k:], A0[:, k:])   # padding columns never written


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_optimizeA_dense_full(oracles, dtype):
    from cmfrec_amd import ops
    O = oracles[dtype]
    rng = np.random.default_rng(2)
    m_u, p, kc = 500, 24, 20
    U = rng.standard_normal((m_u, p)).astype(dtype)
    Ab = (rng.standard_normal((m_u, kc + 2)) * 0.3).astype(dtype)
    Ch, Co = np.zeros((p, kc), dtype), np.zeros((p, kc), dtype)
    ops.optimizeA_dense_full(Ch, Ab, U, 0.7, k=kc, do_B=True, scale_lam=True)
    O.optimizeA_dense_full(Co, Ab, U, 0.7, k=kc, do_B=True, scale_lam=True)
    assert rel_err(Ch, Co) < TOL[dtype]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ku,ki,km,sls,m_u", [(0, 0, 0, False, None), (2, 3, 1, True, None), (0, 0, 1, False, 350)])
def test_optimizeA_collective(oracles, dtype, ku, ki, km, sls, m_u):
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n, p, k = 420, 300, 12, 14
    row, col, val = make_coo(m, n, 9000, 3, counts=False, dtype=dtype, empty_rows=(4, 400))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(9)
    kA, kB = ku + k + km, ki + k + km
    Bm = (rng.standard_normal((n, kB + 1)) * 0.3).astype(dtype)
    Cm = (rng.standard_normal((p, ku + k)) * 0.3).astype(dtype)
    U = rng.standard_normal((m if m_u is None else m_u, p)).astype(dtype)
    A0 = rng.standard_normal((m, kA + 1)).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    kw = dict(w_user=0.5, lam_last=0.2, k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=True, scale_lam_sideinfo=sls)
    ops.optimizeA_collective(Ah, Bm, Cm, csr, U, 0.05, **kw)
    O.optimizeA_collective_chol(Ao, Bm, Cm, csr, U, 0.05, nthreads=4, **kw)
    assert rel_err(Ah[:, :kA], Ao[:, :kA]) < TOL[dtype]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit", [True, False])
@pytest.mark.parametrize("vh", ["stream", "gram", "gram-slice"])
@pytest.mark.parametrize("k", [50, 7, 33])
def test_very_heavy_rows_split_path(oracles, dtype, implicit, vh, k, monkeypatch):
    """Rows above 1024 nnz take the split-row path (one launch pair per CG pass) or, with
    CMFREC_HIP_VH=gram, the single-gather Gramian path (gram_cg_kernels.hpp: one wavefront per slice, or the
    LDS-staged workgroup kernel with CMFREC_HIP_GRAM_KERNEL=slice); 257..1024 the 8-wave
    team with re-streamed tiles; all must agree with the sequential reference sums."""
    from cmfrec_amd import ops
    monkeypatch.setenv("CMFREC_HIP_VH", vh.split("-")[0])
    if vh == "gram-slice":
        monkeypatch.setenv("CMFREC_HIP_GRAM_KERNEL", "slice")
    O = oracles[dtype]
    m, n = 60, 5000            # k = 7: one 16-column block, mostly padding; 33: two blocks and one live column of the third
    row, col, val = make_coo(m, n, 12000, 41, counts=implicit, dtype=dtype, heavy_row=(3, 4500), empty_rows=(8,))
    # a second very heavy row and a 257..1024 one
    rng = np.random.default_rng(4)
    extra_r = np.concatenate([np.full(2500, 10, np.int32), np.full(900, 11, np.int32)])
    keep = (row != 10) & (row != 11)
    extra_c = np.concatenate([rng.choice(n, 2500, replace=False), rng.choice(n, 900, replace=False)]).astype(np.int32)
    extra_v = (np.ceil(rng.lognormal(1, 1, 3400)) if implicit else 0.5 * rng.integers(1, 11, 3400)).astype(dtype)
    row = np.concatenate([row[keep], extra_r]); col = np.concatenate([col[keep], extra_c]); val = np.concatenate([val[keep], extra_v])
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    assert np.diff(csr[0].astype(np.int64)).max() > 2048
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    if implicit:
        ops.optimizeA_implicit(Ah, B, csr, 4.0)
        O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4)
    else:
        bias = (rng.standard_normal(n) * 0.2).astype(dtype)          # the fused "X - bias" of the fit's half-steps
        ops.optimizeA_explicit(Ah, B, csr, 0.05, lam_last=0.3, scale_lam=True, bias_sub=bias)
        csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
        O.optimizeA_explicit(Ao, B, csr_b, 0.05, lam_last=0.3, scale_lam=True, nthreads=4)
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah, Ao, dtype)
    assert np.array_equal(Ah[8], A0[8])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit", [True, False])
@pytest.mark.parametrize("k", [50, 8, 64])
def test_two_rows_per_wave(oracles, dtype, implicit, k):
    """Rows of at most 32 entries (cg_rows_tiny_kernel, one row per wavefront; the name dates from round 5's two-rows-per-wavefront
    kernel, removed in round 6): every length 0 .. 32 several times -- rows on the 16-slot tile, on the 32-slot tile and at the
    boundary between them -- an odd number of such rows, rows that take the first exit (warm start = zero in the implicit model with unit
    counts does not; a row whose start already solves its system does), next to rows of 33 .. 60 entries on the one-row kernels."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n = 140, 600
    rng = np.random.default_rng(k + 3)
    rows, cols = [], []
    for r in range(m):
        cnt = r % 33 if r < 133 else 33 + 4 * (r - 133)
        rows.append(np.full(cnt, r, np.int32)); cols.append(rng.choice(n, cnt, replace=False).astype(np.int32))
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (np.ceil(rng.lognormal(1, 1, len(row))) if implicit else 0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    if implicit:
        ops.optimizeA_implicit(Ah, B, csr, 4.0)
        O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4)
    else:
        bias = (rng.standard_normal(n) * 0.2).astype(dtype)
        ops.optimizeA_explicit(Ah, B, csr, 0.05, lam_last=0.3, scale_lam=True, bias_sub=bias)
        csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
        O.optimizeA_explicit(Ao, B, csr_b, 0.05, lam_last=0.3, scale_lam=True, nthreads=4)
    assert rel_err(Ah, Ao) < TOL[dtype]
    for r in (0, 33, 66):                                   # rows without entries are untouched
        assert np.array_equal(Ah[r], A0[r])
    # a second call starts from the first one's result: after enough calls rows converge and take the early exits
    Ah2, Ao2 = Ah.copy(), Ao.copy()
    for _ in range(6):
        if implicit:
            ops.optimizeA_implicit(Ah2, B, csr, 4.0); O.optimizeA_implicit(Ao2, B, csr, 4.0, nthreads=4)
        else:
            ops.optimizeA_explicit(Ah2, B, csr, 0.05, lam_last=0.3, scale_lam=True, bias_sub=bias)
            O.optimizeA_explicit(Ao2, B, csr_b, 0.05, lam_last=0.3, scale_lam=True, nthreads=4)
    assert rel_err(Ah2, Ao2) < 10 * TOL[dtype]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("implicit", [True, False])
@pytest.mark.parametrize("k", [50, 64, 9])
def test_every_slot_count(oracles, dtype, implicit, k):
    """Rows of EVERY length 1 .. 150 (all slot counts 1 .. 8 of a 64-entry tile and 1 .. 4 of a 32-entry one, full and
    partly filled last slots, one-, two- and four-wave teams; round 5: every tile size 8 NT, NT = 5 .. 8 entries per lane group,
    of the one- and two-wave teams), the first and last length of every tile size of the four- and eight-wave teams (129 .. 512)
    and a few up to 1024 (the re-streamed second tile in double precision), checked row by row."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    lens = list(range(0, 151)) + [160, 161, 192, 193, 224, 225, 250, 256, 257, 300, 320, 321, 384, 385, 448, 449, 511, 512, 513, 640, 777, 1000, 1024]
    m, n = len(lens), 2200
    rng = np.random.default_rng(100 + k)
    rows = [np.full(c, r, np.int32) for r, c in enumerate(lens)]
    cols = [rng.choice(n, c, replace=False).astype(np.int32) for c in lens]
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (np.ceil(rng.lognormal(1, 1, len(row))) if implicit else 0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    if implicit:
        ops.optimizeA_implicit(Ah, B, csr, 4.0)
        O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4)
    else:
        bias = (rng.standard_normal(n) * 0.2).astype(dtype)
        ops.optimizeA_explicit(Ah, B, csr, 0.05, lam_last=0.3, scale_lam=True, bias_sub=bias)
        csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
        O.optimizeA_explicit(Ao, B, csr_b, 0.05, lam_last=0.3, scale_lam=True, nthreads=4)
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah, Ao, dtype)
    assert np.array_equal(Ah[0], A0[0])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lane_primitives_selftest(dtype):
    """DPP / permlane-swap shuffles, broadcasts and the transposed butterflies, checked lane by lane."""
    from cmfrec_amd import _lib
    assert _lib.load(dtype).cmfrec_hip_selftest_lanes() == 0


def test_coo_device_rejects_indices_outside_the_shard():
    """set_X_coo_device takes caller tensors: a key outside [0, rows) or an opposing index outside [0, n_other) must be refused
    before the device build counts with atomics on it (the host entry points validate theirs)."""
    import torch
    from cmfrec_amd.session import AlsSession
    dev = torch.device("cuda", 0)
    s = AlsSession(50, 40, 8, implicit=True, dtype=np.float32)
    key = torch.tensor([0, 3, 49], dtype=torch.int32, device=dev)
    oth = torch.tensor([1, 39, 7], dtype=torch.int32, device=dev)
    val = torch.ones(3, dtype=torch.float32, device=dev)
    s.set_X_coo_device("r", key, oth, val)                              # in range: accepted
    for bad_key, bad_oth in ((50, 1), (-1, 1), (3, 40), (3, -2)):
        k2 = key.clone(); o2 = oth.clone()
        k2[1] = bad_key; o2[1] = bad_oth
        with pytest.raises(ValueError, match="outside the shard"):
            s.set_X_coo_device("r", k2, o2, val)
    with pytest.raises(ValueError, match="outside the shard"):          # 'c': key is the column, other the row
        s.set_X_coo_device("c", torch.tensor([40], dtype=torch.int32, device=dev), torch.tensor([0], dtype=torch.int32, device=dev), val[:1])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m,n,nnz", [(300, 200, 6000), (5000, 70000, 400000), (7, 3, 0)])
def test_coo_device_matches_host(dtype, m, n, nnz):
    """COO -> CSR / CSC on the device (coo_device.hpp) against the reference's counting sort
    (helpers.c:1375-1491: stable in COO order): index work bit-exact, values one IEEE multiply."""
    from cmfrec_amd.session import AlsSession
    rng = np.random.default_rng(5)
    row = rng.integers(0, m, nnz).astype(np.int32)         # duplicates and empty rows on purpose
    col = rng.integers(0, n, nnz).astype(np.int32)
    if nnz:
        row[: nnz // 10] = 3                                # one heavy row (exercises the > 1024 bin when large)
    val = rng.lognormal(size=nnz).astype(dtype)
    alpha = dtype(2.5)
    s = AlsSession(m, n, 8, implicit=True, dtype=dtype)
    s.set_X_coo(row, col, val, alpha=float(alpha))
    for which, key, other, rows in (("r", row, col, m), ("c", col, row, n)):
        p, i, v, order = s.get_X(which)
        perm = np.argsort(key, kind="stable")
        cnt = np.bincount(key, minlength=rows)
        ptr = np.concatenate([[0], np.cumsum(cnt)])
        vh_min = s.vh_min("A" if which == "r" else "B")     # 1025; double precision: 513 where the split rows take the Gramian path
        assert vh_min in (513, 1025)
        for r in np.nonzero(cnt >= vh_min)[0]:              # very heavy rows: entries by opposing index (stable), the
            seg = perm[ptr[r]:ptr[r + 1]]                   # XCD-aware split-row schedule (coo_device.hpp)
            perm[ptr[r]:ptr[r + 1]] = seg[np.argsort(other[seg], kind="stable")]
        assert np.array_equal(p, ptr.astype(np.uint64))
        assert np.array_equal(i, other[perm])
        assert np.array_equal(v, val[perm] * alpha)
        assert np.array_equal(order, np.argsort(-cnt, kind="stable").astype(np.int32))
    # a half-step on the device-built shards equals one on the host-built ones
    if nnz:
        A0 = rng.normal(size=(m, 8)).astype(dtype) * 0.1
        B0 = rng.normal(size=(n, 8)).astype(dtype) * 0.1
        s.set_factors(A=A0, B=B0); s.update("A"); fa = s.get_factors()["A"]
        s2 = AlsSession(m, n, 8, implicit=True, dtype=dtype)
        permr = np.argsort(row, kind="stable"); permc = np.argsort(col, kind="stable")
        pr = np.concatenate([[0], np.cumsum(np.bincount(row, minlength=m))]).astype(np.uint64)
        pc = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=n))]).astype(np.uint64)
        s2.set_X((pr, col[permr], val[permr] * alpha), (pc, row[permc], val[permc] * alpha))
        s2.set_factors(A=A0, B=B0); s2.update("A"); fb = s2.get_factors()["A"]
        assert np.array_equal(fa, fb)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("scale_lam", [False, True])
@pytest.mark.parametrize("long_rows", [False, True])
def test_bias_init_device_matches_oracle(oracles, dtype, scale_lam, long_rows):
    """Centring + two-sided bias start values computed on the device (coo_device.hpp) against the
    oracle's restatement of calc_mean_and_center / initialize_biases_twosided
    (common.c:3423-3648, 4410-4909): sequential per-row running means in the same order.  Centring
    and the CSR are bit-exact; the biases agree to 1 ulp-level (the compiled reference fuses
    `cnt + lam*cnt` into an FMA in one of the two sweeps, the device kernel in neither)."""
    from cmfrec_amd.session import AlsSession
    O = oracles[dtype]
    m, n, k = 400, 250, 4
    row, col, val = make_coo(m, n, 9000, seed=11, counts=False, dtype=dtype, empty_rows=(5, 17))
    if long_rows:      # rows / columns beyond 1024 entries take the wave-parallel sum (same mean up to rounding)
        m, n = 3000, 2500
        row, col, val = make_coo(m, n, 60000, seed=12, counts=False, dtype=dtype, heavy_row=(3, 1800), empty_rows=(5, 17))
        rng = np.random.default_rng(3)
        extra = np.setdiff1d(rng.choice(m, 1500, replace=False), row[col == 7]).astype(np.int32)
        extra = extra[(extra != 5) & (extra != 17)]
        row = np.concatenate([row, extra]); col = np.concatenate([col, np.full(len(extra), 7, np.int32)])
        val = np.concatenate([val, (0.5 * rng.integers(1, 11, len(extra))).astype(dtype)])
        assert np.bincount(row).max() > 1024 and np.bincount(col).max() > 1024
    Xc = val.copy()
    gm = O.calc_mean_and_center(Xc, nthreads=1)            # centres Xc in place, returns the mean
    assert gm != 0
    csr, csc = O.coo_to_csr_and_csc(row, col, Xc, m, n)
    lam = 0.7
    bA, bB = O.initialize_biases_twosided(m, n, csr, csc, lam, lam, scale_lam)
    s = AlsSession(m, n, k, implicit=False, dtype=dtype, lam=lam, user_bias=True, item_bias=True, scale_lam=scale_lam)
    s.set_X_coo(row, col, val, subtract=float(gm))
    p, i, v, _ = s.get_X("r")
    ptr = csr[0].astype(np.int64)
    vh_min = s.vh_min("A")
    for r in range(m):                                      # split rows (>= vh_min entries) are re-ordered by column (coo_device.hpp)
        a, b = ptr[r], ptr[r + 1]
        o = np.argsort(csr[1][a:b], kind="stable") if b - a >= vh_min else np.arange(b - a)
        assert np.array_equal(i[a:b], csr[1][a:b][o]) and np.array_equal(v[a:b], csr[2][a:b][o])
    s.set_factors(A=np.zeros((m, k), dtype), B=np.zeros((n, k), dtype))
    s.init_biases(lam, lam)
    f = s.get_factors()
    tol = 1e-13 if dtype is np.float64 else 2e-6
    assert np.abs(f["biasA"] - bA).max() <= tol * max(1.0, np.abs(bA).max())
    assert np.abs(f["biasB"] - bB).max() <= tol * max(1.0, np.abs(bB).max())
    assert np.array_equal(f["biasA"][[5, 17]], np.zeros(2, dtype))          # rows without entries


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("nu,n,k,n_top", [(37, 1000, 50, 10), (5, 300, 17, 100), (64, 5000, 64, 10)])
def test_topN_batch(dtype, nu, n, k, n_top):
    """Batched top-N (topn_kernels.hpp) against a plain ranking: score = A_u . B_i + biasB[i], excluded items
    skipped, descending, ties by lower id (reference scoring rule: topN, common.c:5127-5380)."""
    from cmfrec_amd import ops
    rng = np.random.default_rng(nu + n)
    A = rng.standard_normal((nu, k)).astype(dtype); B = rng.standard_normal((n, k)).astype(dtype)
    B[7] = B[3]; B[11] = B[3]                                   # exact ties
    bias = rng.standard_normal(n).astype(dtype); bias[7] = bias[3]; bias[11] = bias[3]
    lens = rng.integers(0, min(60, n - n_top), nu); lens[0] = 0
    ep = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ei = np.concatenate([rng.choice(n, l, replace=False) for l in lens]).astype(np.int32) if lens.sum() else np.zeros(0, np.int32)
    ids, sc = ops.topN_batch(A, B, n_top=n_top, biasB=bias, exclude=(ep, ei))
    S = (A.astype(np.float64) @ B.astype(np.float64).T) + bias.astype(np.float64)
    for u in range(nu):
        s = S[u].copy(); s[ei[int(ep[u]):int(ep[u + 1])]] = -np.inf
        want = np.lexsort((np.arange(n), -s))[:n_top]
        if dtype is np.float64:
            assert np.array_equal(ids[u], want), u
        else:                                                    # fp32 scores may swap near-ties: compare as scores
            assert np.allclose(np.sort(s[ids[u]])[::-1], s[want], rtol=1e-4, atol=1e-4), u
            assert len(set(ids[u].tolist())) == n_top and not set(ids[u].tolist()) & set(ei[int(ep[u]):int(ep[u + 1])].tolist())
        assert np.allclose(sc[u], s[ids[u]], rtol=1e-12 if dtype is np.float64 else 1e-4, atol=1e-12 if dtype is np.float64 else 1e-4)
    # no bias, no exclusion, and P@10 of a golden fit computed through the device ranking
    ids2, _ = ops.topN_batch(A, B, n_top=min(n_top, 10))
    S2 = A.astype(np.float64) @ B.astype(np.float64).T
    if dtype is np.float64:
        assert np.array_equal(ids2[1], np.lexsort((np.arange(n), -S2[1]))[:min(n_top, 10)])


@pytest.mark.parametrize("dtype,k", [(np.float64, 161), (np.float64, 250), (np.float32, 200), (np.float32, 257)])
@pytest.mark.parametrize("implicit", [False, True])
def test_cholesky_large_k(oracles, dtype, k, implicit):
    """Closed-form row update beyond k_t = 144: the 12/16/17-block instantiations of chol_rows_kernel (k = 256 + bias
    in single precision is BASELINE config 5's system size)."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    m, n = 48, 700
    row, col, val = make_coo(m, n, 9000, 3 + k, counts=implicit, dtype=dtype, heavy_row=(3, 600), empty_rows=(5,))
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(k)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    if implicit:
        ops.optimizeA_implicit(Ah, B, csr, 4.0, use_cg=False)
        O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4, use_cg=False)
    else:
        ops.optimizeA_explicit(Ah, B, csr, 0.05, k=k, lam_last=0.3, scale_lam=True, use_cg=False)
        O.optimizeA_explicit(Ao, B, csr, 0.05, k=k, lam_last=0.3, scale_lam=True, use_cg=False, nthreads=4)
    assert rel_err(Ah, Ao) < (1e-9 if dtype is np.float64 else 1e-3)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k,ku,ki,km", [(8, 0, 0, 0), (50, 2, 1, 1), (120, 3, 0, 2)])
def test_collective_sparse_sideinfo(oracles, dtype, k, ku, ki, km):
    """Two gather sources in one launch of the Cholesky row kernel: rows of B through X and rows of C through the sparse
    side information; explicit (with the lambda scalings and the fused bias subtraction) and implicit."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    tol = 1e-10 if dtype is np.float64 else 3e-4
    rng = np.random.default_rng(k)
    m, n, p, m_u = 700, 500, 40, 640
    B = (rng.standard_normal((n, ki + k + km)) * 0.3).astype(dtype); Cm = (rng.standard_normal((p, ku + k)) * 0.3).astype(dtype)
    ur, uc, _ = make_coo(m_u, p, 5000, 21, counts=False, dtype=dtype, heavy_row=(8, 35), empty_rows=(3, 5, 600))
    uv = rng.standard_normal(len(ur)).astype(dtype)
    ucsr, _ = O.coo_to_csr_and_csc(ur, uc, uv, m_u, p)
    bias = (rng.standard_normal(n) * 0.1).astype(dtype)
    for implicit in (False, True):
        row, col, val = make_coo(m, n, 14000, 22, counts=implicit, dtype=dtype, heavy_row=(9, 300), empty_rows=(3, 7, 650))
        csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
        A0 = rng.standard_normal((m, ku + k + km)).astype(dtype)
        for sl, sls in (((False, False),) if implicit else ((False, False), (True, False), (True, True))):
            a1, a2 = A0.copy(), A0.copy()
            kw = dict(w_user=2.5, lam_last=None if implicit else 1.3, k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=sl,
                      scale_lam_sideinfo=sls, implicit=implicit)
            ops.optimizeA_collective_sparse(a1, B, Cm, csr, ucsr, 0.7, bias_sub=None if implicit else bias, **kw)
            csr_b = csr if implicit else (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
            O.optimizeA_collective_sparse(a2, B, Cm, csr_b, ucsr, 0.7, nthreads=4, **kw)
            assert rel_err(a1, a2) < tol, (implicit, sl, sls)
            assert not a1[3].any() and not a1[650].any()          # neither observations nor attributes


def _driver_wsum(csr_p, w_csr, dtype):
    """wsumA of the driver: the row's weights summed in double, 1 for a row without entries (collective.c:7988-7998)."""
    p = csr_p.astype(np.int64)
    # (np.cumsum adds entry by entry like the driver's loop; np.sum adds pairwise)
    return np.array([np.cumsum(w_csr[p[r]:p[r + 1]].astype(np.float64))[-1] if p[r + 1] > p[r] else 1.0 for r in range(len(p) - 1)]).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mode", ["cg", "pcg", "chol"])
@pytest.mark.parametrize("k", [50, 9, 64, 100])
def test_observation_weights_every_row_length(oracles, dtype, mode, k):
    """Observation weights in every row kernel of the explicit model: rows of every length 0 .. 150 (two-rows-per-wavefront,
    32- and 64-entry tiles, all team sizes), 250 .. 1024 (four- and eight-wave teams, re-streamed tiles), split rows (2500 and
    4500 entries), k = 100 on the one-wavefront-per-row kernel; the fused bias subtraction, lambda scaled by the driver's sums
    of weights (scale_lam) with its own value on the last unknown.  Row by row against the oracle."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    lens = list(range(0, 151)) + [250, 257, 300, 511, 512, 513, 640, 777, 1000, 1024, 2500, 4500]
    m, n = len(lens), 5000
    rng = np.random.default_rng(300 + k)
    rows = [np.full(c, r, np.int32) for r, c in enumerate(lens)]
    cols = [rng.choice(n, c, replace=False).astype(np.int32) for c in lens]
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    w = (0.2 + 2.5 * rng.random(len(row)) ** 2).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    w_csr = O.coo_to_csr_and_csc(row, col, w, m, n)[0][2]
    wsum = _driver_wsum(csr[0], w_csr, dtype)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    bias = (rng.standard_normal(n) * 0.2).astype(dtype)
    kw = dict(lam_last=0.3, scale_lam=True, use_cg=mode != "chol", precondition_cg=mode == "pcg")
    Ah, Ao, Au = A0.copy(), A0.copy(), A0.copy()
    ops.optimizeA_explicit(Ah, B, csr, 0.05, bias_sub=bias, weight=w_csr, wsum=wsum, **kw)
    csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
    O.optimizeA_explicit(Ao, B, csr_b, 0.05, nthreads=4, weight=w_csr, wsum=wsum, **kw)
    # (the preconditioned variant runs its three steps without exits: in single precision the 4500-entry rows of the k = 100
    #  case end 1.2e-4 from the oracle's sequential sums)
    assert rel_err(Ah, Ao) < TOL[dtype] * (3 if (mode == "pcg" and dtype is np.float32) else 1)
    check_rows(Ah, Ao, dtype)
    assert np.array_equal(Ah[0], A0[0])
    # without `wsum` the rows' own sums are taken: the same numbers here
    Ah2 = A0.copy()
    ops.optimizeA_explicit(Ah2, B, csr, 0.05, bias_sub=bias, weight=w_csr, **kw)
    assert np.array_equal(Ah2, Ah)
    # unit weights reproduce the unweighted kernels bit for bit (a multiplication by one), other weights do not
    ops.optimizeA_explicit(Au, B, csr, 0.05, bias_sub=bias, **kw)
    A1 = A0.copy()
    ops.optimizeA_explicit(A1, B, csr, 0.05, bias_sub=bias, weight=np.ones_like(w_csr), **kw)
    if mode == "chol" or (mode == "cg" and k > 64):
        # (weighted Cholesky launches stay on the workgroup-per-row kernel, weighted CG beyond 64 unknowns on the lane <-> unknown
        #  kernel while the unweighted one goes through the row's Gramian: other summation orders)
        assert rel_err(A1, Au) < TOL[dtype]
    else:
        assert np.array_equal(A1, Au)
    assert rel_err(Ah, Au) > 1e-3


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("vh", ["stream", "gram", "gram-slice"])
def test_observation_weights_split_rows(oracles, dtype, vh, monkeypatch):
    """The split rows' weights on each of their three paths: streamed per CG pass (entries re-sorted by index, the weights
    with them), Gramian per slice from one wavefront, Gramian per slice from the LDS-staged workgroup kernel."""
    from cmfrec_amd import ops
    monkeypatch.setenv("CMFREC_HIP_VH", vh.split("-")[0])
    if vh == "gram-slice":
        monkeypatch.setenv("CMFREC_HIP_GRAM_KERNEL", "slice")
    O = oracles[dtype]
    m, n, k = 40, 6000, 33
    row, col, val = make_coo(m, n, 9000, 43, counts=False, dtype=dtype, heavy_row=(3, 5200), empty_rows=(8,))
    rng = np.random.default_rng(12)
    extra_c = rng.choice(n, 2100, replace=False).astype(np.int32)
    keep = row != 10
    row = np.concatenate([row[keep], np.full(2100, 10, np.int32)]); col = np.concatenate([col[keep], extra_c])
    val = np.concatenate([val[keep], (0.5 * rng.integers(1, 11, 2100)).astype(dtype)])
    w = (0.2 + 2.5 * rng.random(len(row)) ** 2).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    w_csr = O.coo_to_csr_and_csc(row, col, w, m, n)[0][2]
    wsum = _driver_wsum(csr[0], w_csr, dtype)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    ops.optimizeA_explicit(Ah, B, csr, 0.05, lam_last=0.3, scale_lam=True, weight=w_csr, wsum=wsum)
    O.optimizeA_explicit(Ao, B, csr, 0.05, lam_last=0.3, scale_lam=True, nthreads=4, weight=w_csr, wsum=wsum)
    assert rel_err(Ah, Ao) < TOL[dtype]
    check_rows(Ah, Ao, dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("k", [65, 80, 97, 100, 128, 129])
def test_explicit_cg_beyond_64_unknowns(oracles, dtype, k):
    """Explicit-model CG with 64 < k <= 129 unknowns (round 4): the Cholesky path's producer builds the row's Gramian on the matrix
    cores, gram_cg_wide_kernel runs the reference's CG steps on it (gram_cg_wide_kernels.hpp) -- every tile grid of the producer
    (4 / 6 / 8 blocks, with and without the border column), rows of every length 0 .. 150, four- and eight-wave lengths, split rows
    (2500 and 4500 entries: partials summed in slice order), the fused bias subtraction, scale_lam with its own lambda on the last
    unknown.  The rows beyond 256 entries take this path, the shorter ones the lane <-> unknown kernel.  Row by row against the
    oracle."""
    from cmfrec_amd import ops
    O = oracles[dtype]
    lens = list(range(0, 151)) + [250, 257, 512, 513, 777, 1024, 2500, 4500]
    m, n = len(lens), 5000
    rng = np.random.default_rng(900 + k)
    rows = [np.full(c, r, np.int32) for r, c in enumerate(lens)]
    cols = [rng.choice(n, c, replace=False).astype(np.int32) for c in lens]
    row, col = np.concatenate(rows), np.concatenate(cols)
    perm = rng.permutation(len(row)); row, col = row[perm], col[perm]
    val = (0.5 * rng.integers(1, 11, len(row))).astype(dtype)
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    bias = (rng.standard_normal(n) * 0.2).astype(dtype)
    for scale_lam in (True, False):
        kw = dict(lam_last=0.3, scale_lam=scale_lam, use_cg=True)
        Ah, Ao = A0.copy(), A0.copy()
        ops.optimizeA_explicit(Ah, B, csr, 0.05, bias_sub=bias, **kw)
        csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
        O.optimizeA_explicit(Ao, B, csr_b, 0.05, nthreads=4, **kw)
        assert rel_err(Ah, Ao) < TOL[dtype]
        check_rows(Ah, Ao, dtype)
        assert np.array_equal(Ah[0], A0[0])                  # a row without entries stays as it is


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_session_naz_weighted_multipliers_follow_uploads(dtype):
    """Level-2 session API, NA_as_zero_X with observation weights under scale_lam: the lambda multipliers of that model (sum of
    the row's weights + its absent entries) live beside the plain sums of the weights and are rebuilt whenever X is uploaded
    again or the flag changes (they used to overwrite the plain sums once).  A second upload of X + a second
    set_NA_as_zero_X(on) gives the bits of a fresh session; switching the flag off gives the bits of a session that never had it;
    a different X after the flag gives what a fresh session gives for that X."""
    from cmfrec_amd.session import AlsSession
    rng = np.random.default_rng(11)
    m, n, k, nnz = 300, 200, 12, 6000
    lin = rng.choice(m * n, size=nnz, replace=False)
    row, col = (lin // n).astype(np.int32), (lin % n).astype(np.int32)
    val = rng.standard_normal(nnz).astype(dtype)
    w = rng.uniform(0.5, 3.0, nnz).astype(dtype)
    A0 = (rng.standard_normal((m, k)) * 0.1).astype(dtype)
    B0 = (rng.standard_normal((n, k)) * 0.1).astype(dtype)

    def run(steps):
        s = AlsSession(m, n, k, implicit=False, dtype=dtype, lam=0.05, use_cg=False, scale_lam=True)
        for st in steps:
            if st == "X": s.set_X_coo(row, col, val, weight=w)
            elif st == "X2": s.set_X_coo(row[: nnz // 2], col[: nnz // 2], val[: nnz // 2], weight=w[: nnz // 2])
            elif st == "on": s.set_NA_as_zero_X(True)
            elif st == "off": s.set_NA_as_zero_X(False)
        s.set_factors(A=A0, B=B0)
        s.update("A", use_cholesky=True)
        s.update("B", use_cholesky=True)
        out = s.get_factors()
        s.close()
        return out["A"].copy(), out["B"].copy()

    fresh = run(["X", "on"])
    again = run(["X", "on", "X", "on"])
    assert np.array_equal(fresh[0], again[0]) and np.array_equal(fresh[1], again[1])
    flag_first = run(["X2", "on", "X"])                   # the upload after the flag rebuilds the multipliers for the new X
    assert np.array_equal(fresh[0], flag_first[0]) and np.array_equal(fresh[1], flag_first[1])
    plain = run(["X"])
    back = run(["X", "on", "off"])
    assert np.array_equal(plain[0], back[0]) and np.array_equal(plain[1], back[1])
    assert np.abs(plain[0] - fresh[0]).max() > 1e-3       # (the flag changes the model)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [2, 3, 17, 64, 65, 129, 256, 257, 289, 320])
@pytest.mark.parametrize("kind", ["gram", "rank_deficient", "diagonal", "clustered"])
def test_sym_eig(dtype, n, kind):
    """The library's own symmetric eigen-decomposition (eig_kernels.hpp: Householder tridiagonalisation + implicit QL, the
    rotations applied row-parallel; what the low-rank row path runs on w C^T C once per half-step) against numpy.linalg.eigh:
    eigenvalues, A Q = Q diag(lam), Q^T Q = I -- on Gramians of full and of deficient rank (p < k: a null space), on a matrix
    that is already diagonal (every reflector is the identity) and on clustered spectra; widths around the 64-row workgroups of
    the QL kernel and its 32-row form (n > 288).  The one-workgroup Jacobi kernel (CMFREC_HIP_EIG=jacobi) gives the same."""
    import ctypes as C
    from cmfrec_amd import _lib
    lib = _lib.load(dtype)
    rng = np.random.default_rng(n * 7 + len(kind))
    if kind == "gram":
        Cm = rng.standard_normal((2 * n + 3, n)); A = Cm.T @ Cm * 0.3
    elif kind == "rank_deficient":
        Cm = rng.standard_normal((max(1, n // 3), n)); A = Cm.T @ Cm
    elif kind == "diagonal":
        A = np.diag(rng.uniform(0.0, 5.0, n))
    else:
        Qr, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam0 = np.concatenate([np.full(n // 2, 2.0), np.full(n - n // 2, 2.0 + 1e-9)]) if n > 3 else np.arange(1.0, n + 1)
        A = (Qr * lam0) @ Qr.T
    A = (0.5 * (A + A.T)).astype(dtype)
    R = _lib.real(dtype)
    tol = 2e-12 if dtype is np.float64 else 2e-5        # (float: the results are rounded to single precision on the way out)
    ref = np.linalg.eigvalsh(A.astype(np.float64))
    scale = max(np.abs(ref).max(), 1e-30)
    for method in (0, 1):
        Q = np.empty((n, n), dtype); lam = np.empty(n, dtype); ms = C.c_double(0)
        rc = lib.cmfrec_hip_sym_eig(C.c_int(n), _lib.ptr(A), _lib.ptr(Q), _lib.ptr(lam), C.c_int(method), C.c_int(1), C.byref(ms))
        assert rc == 0, (method, lib.cmfrec_hip_last_error())
        Q64, l64, A64 = Q.astype(np.float64), lam.astype(np.float64), A.astype(np.float64)
        assert np.isfinite(Q64).all() and np.isfinite(l64).all()
        assert np.abs(np.sort(l64) - np.maximum(ref, 0.0)).max() <= tol * scale * 20, (method, kind, n)
        assert np.abs(A64 @ Q64 - Q64 * l64).max() <= tol * scale * 50, (method, kind, n)
        assert np.abs(Q64.T @ Q64 - np.eye(n)).max() <= tol * 50, (method, kind, n)
