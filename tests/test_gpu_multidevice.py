"""GPU: several device shards behind the unchanged C signature: CMFREC_HIP_DEVICES lists HIP device
ordinals; fit_collective_implicit_als then cuts users / items into row blocks, one session per entry, and moves the updated
rows between the replicas with peer copies ordered by events (cmfrec_amd/csrc/fit.hip, fit_implicit_multi).  An ordinal may
repeat, which shards ONE device -- the only way to run the path on a single-GPU box, and it exercises everything but the
xGMI hop: block boundaries (equal users, nnz-balanced items), shard construction from the COO, the exchange and its event
ordering, the epilogue on the first replica.  Rows are independent given the opposing matrix, so the result must be the
single-device fit bit for bit -- for rows that take the same kernel path on the shard as on the whole matrix (at most 1024
entries); split rows choose their path and slice length from the shard's statistics and agree to rounding."""
import numpy as np
import pytest

from conftest import make_coo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", ["0", "0,0", "0,0,0"])
@pytest.mark.parametrize("use_cg", [True, False])
@pytest.mark.parametrize("use_float", [False, True])
def test_devices_env_matches_single_device(devices, use_cg, use_float, monkeypatch):
    from cmfrec_amd import CMF_implicit
    m, n = 900, 700
    row, col, val = make_coo(m, n, 30000, 3, heavy_row=(5, 600), empty_rows=(7, 11))
    kw = dict(k=20, lambda_=3.0, niter=3, use_cg=use_cg, use_float=use_float, finalize_chol=use_cg, random_state=7)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF_implicit(**kw).fit((row, col, val), shape=(m, n))
    monkeypatch.setenv("CMFREC_HIP_DEVICES", devices)
    shard = CMF_implicit(**kw).fit((row, col, val), shape=(m, n))
    assert np.isfinite(shard.A_).all() and np.abs(shard.A_).max() > 0
    assert np.array_equal(shard.A_, base.A_) and np.array_equal(shard.B_, base.B_)
    assert np.array_equal(shard._BtB, base._BtB)


@pytest.mark.parametrize("use_float", [False, True])
def test_devices_env_split_rows_agree_to_rounding(use_float, monkeypatch):
    """A user and an item above the split-row boundary (> 1024 entries): the shards may pick the other split-row path or slice
    length than the single-device run (per-shard statistics), so the sums may be ordered differently -- equal to rounding."""
    from cmfrec_amd import CMF_implicit
    m, n = 2600, 2400
    row, col, val = make_coo(m, n, 90000, 5, heavy_row=(5, 2000), empty_rows=(7,))
    hr = np.arange(0, m, 2, dtype=row.dtype)[:1200]                      # a heavy item too: column 3 rated by 1200 users
    keep = ~((col == 3) & np.isin(row, hr))
    row = np.concatenate([row[keep], hr]); col = np.concatenate([col[keep], np.full(hr.size, 3, col.dtype)])
    val = np.concatenate([val[keep], np.ones(hr.size, val.dtype)])
    kw = dict(k=20, lambda_=3.0, niter=3, use_cg=True, use_float=use_float, finalize_chol=False, random_state=7)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF_implicit(**kw).fit((row, col, val), shape=(m, n))
    monkeypatch.setenv("CMFREC_HIP_DEVICES", "0,0,0")
    shard = CMF_implicit(**kw).fit((row, col, val), shape=(m, n))
    tol = 2e-4 if use_float else 1e-11
    for a, b in ((shard.A_, base.A_), (shard.B_, base.B_)):
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= tol * np.abs(b).max()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("opts", [dict(use_cg=True, finalize_chol=True), dict(use_cg=False, scale_lam=True),
                                  dict(use_cg=True, finalize_chol=False, user_bias=False, center=False),
                                  dict(use_cg=False, item_bias=False, lambda_=[0.5, 0.4, 3.0, 2.0, 1.0, 1.0]),
                                  dict(use_cg=False, nonneg=True, user_bias=False, item_bias=False, center=False),
                                  dict(use_cg=False, l1_lambda=0.05)],
                         ids=["cg+fin", "chol-scale_lam", "cg-item_bias-only", "chol-lam6-user_bias-only", "nonneg", "l1"])
@pytest.mark.parametrize("use_float", [False, True])
def test_devices_env_explicit_model(devices, opts, use_float, monkeypatch):
    """fit_collective_explicit_als behind CMFREC_HIP_DEVICES (plain model: biases, centring, CG / Cholesky, scale_lam, per-matrix
    penalties, non-negative / L1): row-block shards, the bias columns riding in the exchanged rows, bias start values from the
    temporary whole-X session, the precomputed matrices from the first replica.  No split rows here (at most 600 entries), so
    every row takes the same kernel as on one device: bit for bit, including the seeded start."""
    from cmfrec_amd import CMF
    m, n = 900, 700
    row, col, val = make_coo(m, n, 30000, 3, counts=False, heavy_row=(5, 600), empty_rows=(7, 11))
    o = dict(opts)
    kw = dict(k=12, lambda_=o.pop("lambda_", 2.0), niter=3, use_float=use_float, random_state=7, nthreads=1, **o)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF(**kw).fit((row, col, val), shape=(m, n))
    monkeypatch.setenv("CMFREC_HIP_DEVICES", devices)
    shard = CMF(**kw).fit((row, col, val), shape=(m, n))
    assert np.isfinite(shard.A_).all() and np.abs(shard.A_).max() > 0
    for name in ("A_", "B_", "user_bias_", "item_bias_", "_BtB", "_TransBtBinvBt", "_B_plus_bias"):
        assert np.array_equal(getattr(shard, name), getattr(base, name)), name
    assert shard.glob_mean_ == base.glob_mean_


def test_devices_env_explicit_falls_back(monkeypatch):
    """Configurations the sharded driver does not take (observation weights, sparse side information, ...) run on the first
    listed device: exactly the single-device result."""
    from cmfrec_amd import CMF
    m, n = 300, 200
    row, col, val = make_coo(m, n, 5000, 4, counts=False)
    W = np.random.default_rng(1).random(len(val)) + 0.5
    kw = dict(k=6, niter=2, use_float=False, random_state=3, nthreads=1)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF(**kw).fit((row, col, val), W=W, shape=(m, n))
    monkeypatch.setenv("CMFREC_HIP_DEVICES", "0,0")
    two = CMF(**kw).fit((row, col, val), W=W, shape=(m, n))
    assert np.array_equal(two.A_, base.A_) and np.array_equal(two.B_, base.B_)


def test_devices_env_ignores_bad_ordinals(monkeypatch):
    """Ordinals outside the visible devices are dropped; an empty list falls back to the current device."""
    from cmfrec_amd import CMF_implicit
    m, n = 300, 200
    row, col, val = make_coo(m, n, 5000, 4)
    monkeypatch.setenv("CMFREC_HIP_DEVICES", "99,-1")
    mdl = CMF_implicit(k=8, niter=1, use_float=False).fit((row, col, val), shape=(m, n))
    assert np.isfinite(mdl.A_).all()


@pytest.mark.parametrize("use_cg", [True, False])
def test_devices_env_dense_X_stays_on_one_device(use_cg, monkeypatch):
    """A dense X (NaN = missing) follows the reference's per-half-step choice of solver (optimizeA Cases 1-2, common.c:2787-3116) and
    has its empty rows zeroed after the loop; the row-block loop knows neither, so such a fit keeps to the first listed device
    and must return exactly what the single-device call returns (it used to run max_cg_steps CG steps instead)."""
    from cmfrec_amd import CMF
    rng = np.random.default_rng(11)
    m, n = 120, 90
    X = rng.integers(1, 11, size=(m, n)).astype(np.float64) * 0.5
    X[rng.random((m, n)) < 0.7] = np.nan
    X[5, :] = np.nan                                                     # an empty row
    kw = dict(k=6, lambda_=1.5, niter=3, use_cg=use_cg, finalize_chol=False, use_float=False, random_state=3, nthreads=1)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF(**kw).fit(X)
    monkeypatch.setenv("CMFREC_HIP_DEVICES", "0,0")
    shard = CMF(**kw).fit(X)
    for name in ("A_", "B_", "user_bias_", "item_bias_"):
        assert np.array_equal(getattr(shard, name), getattr(base, name)), name
    assert np.all(shard.A_[5] == 0)


def _side_problem(seed=5, m=700, n=500, nnz=24000, p=9, q=7, counts=False):
    rng = np.random.default_rng(seed)
    row, col, val = make_coo(m, n, nnz, seed, counts=counts, heavy_row=(5, 400), empty_rows=(7, 11))
    U = rng.standard_normal((m, p)) + 0.5
    I = rng.standard_normal((n, q)) - 1.0
    return m, n, row, col, val, U, I


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("opts", [dict(use_cg=False), dict(use_cg=True, finalize_chol=True), dict(use_cg=False, scale_lam=True, scale_lam_sideinfo=True),
                                  dict(use_cg=False, k_user=2, k_item=1, k_main=1, w_user=0.5, w_item=2.0)],
                         ids=["chol", "cg+fin", "chol-scale_lam_sideinfo", "chol-k_user-k_item-k_main"])
@pytest.mark.parametrize("sides", ["U", "I", "UI"])
@pytest.mark.parametrize("use_float", [False, True])
def test_devices_env_collective_model(devices, opts, sides, use_float, monkeypatch):
    """The explicit model WITH dense side information behind CMFREC_HIP_DEVICES (BASELINE configs 3 and 5 are of
    this kind): U / I cut into the shards' row blocks, the C / D update as partial sums per shard + their total on every shard +
    the same small solve everywhere (multi_sideinfo_step, fit.hip), bias columns riding in the exchanged rows.  Against the
    single-device fit: the partial sums are added in another order, nothing else differs."""
    from cmfrec_amd import CMF
    m, n, row, col, val, U, I = _side_problem()
    kw = dict(k=10, lambda_=1.5, niter=3, use_float=use_float, random_state=7, nthreads=1, **opts)
    side = dict(U=U if "U" in sides else None, I=I if "I" in sides else None)
    if "U" not in sides: kw.pop("k_user", None)
    if "I" not in sides: kw.pop("k_item", None)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF(**kw).fit((row, col, val), shape=(m, n), **side)
    monkeypatch.setenv("CMFREC_HIP_DEVICES", devices)
    shard = CMF(**kw).fit((row, col, val), shape=(m, n), **side)
    tol = 2e-3 if use_float else 1e-10
    names = ["A_", "B_", "user_bias_", "item_bias_"] + (["C_"] if "U" in sides else []) + (["D_"] if "I" in sides else [])
    for name in names:
        a, b = getattr(shard, name), getattr(base, name)
        assert np.isfinite(a).all() and np.abs(b).max() > 0, name
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), name


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("use_cg", [False, True])
@pytest.mark.parametrize("use_float", [False, True])
def test_devices_env_implicit_with_side_information(devices, use_cg, use_float, monkeypatch):
    from cmfrec_amd import CMF_implicit
    m, n, row, col, val, U, I = _side_problem(seed=6, counts=True)
    kw = dict(k=10, lambda_=2.0, niter=3, use_cg=use_cg, finalize_chol=False, use_float=use_float, random_state=7, k_user=1, k_item=2)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = CMF_implicit(**kw).fit((row, col, val), shape=(m, n), U=U, I=I)
    monkeypatch.setenv("CMFREC_HIP_DEVICES", devices)
    shard = CMF_implicit(**kw).fit((row, col, val), shape=(m, n), U=U, I=I)
    tol = 2e-3 if use_float else 1e-10
    for name in ("A_", "B_", "C_", "D_"):
        a, b = getattr(shard, name), getattr(base, name)
        assert np.isfinite(a).all() and np.abs(b).max() > 0, name
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), name


@pytest.mark.parametrize("model", ["explicit", "implicit"])
def test_single_shard_through_rccl(model, monkeypatch):
    """What a one-GPU box can run of the RCCL transport of the C host (fit.hip, MultiDev): CMFREC_HIP_SHARDED=1 sends a fit on ONE
    listed device through the sharded driver, CMFREC_HIP_EXCHANGE=rccl makes the library bind librccl, build a communicator
    (ncclCommInitAll over the one device) and all-reduce the C / D partial sums through it (a one-rank ncclAllReduce is the
    identity).  The N > 1 exchange (ncclSend / ncclRecv group) needs N devices and is covered by construction only."""
    from cmfrec_amd import CMF, CMF_implicit
    m, n, row, col, val, U, I = _side_problem(seed=8, counts=(model == "implicit"))
    if model == "explicit":
        mk = lambda: CMF(k=8, lambda_=1.5, niter=2, use_cg=False, use_float=False, random_state=3, nthreads=1)
    else:
        mk = lambda: CMF_implicit(k=8, lambda_=1.5, niter=2, use_cg=True, use_float=False, random_state=3)
    monkeypatch.delenv("CMFREC_HIP_DEVICES", raising=False)
    base = mk().fit((row, col, val), shape=(m, n), U=U, I=I)
    monkeypatch.setenv("CMFREC_HIP_DEVICES", "0")
    monkeypatch.setenv("CMFREC_HIP_SHARDED", "1")
    monkeypatch.setenv("CMFREC_HIP_EXCHANGE", "rccl")
    one = mk().fit((row, col, val), shape=(m, n), U=U, I=I)
    for name in ("A_", "B_", "C_", "D_"):
        a, b = getattr(one, name), getattr(base, name)
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), name
