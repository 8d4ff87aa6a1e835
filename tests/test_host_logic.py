"""CPU: host-side logic of the product that does not need a GPU (input munging of the estimators,
unsupported-option errors, bench.py's workload generator and byte accounting)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_estimator_defaults_follow_reference():
    from cmfrec_amd import CMF, CMF_implicit
    a = CMF()
    assert (a.k, a.lambda_, a.use_cg, a.user_bias, a.item_bias, a.center, a.max_cg_steps, a.finalize_chol,
            a.use_float, a.niter, a.random_state) == (40, 10.0, True, True, True, True, 3, True, True, 10, 1)
    b = CMF_implicit()
    assert (b.k, b.lambda_, b.alpha, b.w_user, b.w_item, b.finalize_chol, b.use_float, b.niter) == \
        (50, 1.0, 1.0, 10.0, 10.0, False, True, 10)
    assert a.nthreads >= 1 and b.dtype_ is np.float32


def test_unsupported_options_raise():
    from cmfrec_amd import CMF, CMF_implicit
    with pytest.raises(NotImplementedError):
        CMF(method="lbfgs")
    assert CMF(NA_as_zero=True).NA_as_zero          # (its prediction matrices exist for the model without side information: checked in fit())
    assert CMF(scale_bias_const=True, scale_lam=True).scale_bias_const
    six = CMF_implicit(l1_lambda=np.array([0.1, 0.2, 0.3, 0.4, 0.5, 0.6]), lambda_=[1, 2, 3, 4, 5, 6])
    assert six.l1_lambda == 0.0 and six._l16[5] == 0.6 and six.lambda_ == 0.0 and six._lam6[2] == 3.0   # scalar 0 + array, like the reference
    with pytest.raises(ValueError):
        CMF(lambda_=[1.0, 2.0])
    assert CMF_implicit(l1_lambda=0.1).l1_lambda == 0.1
    with pytest.raises(NotImplementedError):
        CMF(add_implicit_features=True, nonneg=True)         # the reference crashes on this combination: nothing to pin
    assert CMF(add_implicit_features=True).add_implicit_features and CMF(add_implicit_features=True, use_cg=False).w_implicit == 0.5
    assert CMF(nonneg=True, nonneg_C=True).nonneg_C and CMF_implicit(nonneg=True, max_cd_steps=50).max_cd_steps == 50


def test_coo_input_handling():
    import scipy.sparse as sp
    from cmfrec_amd.models import _coo_triplet
    X = sp.random(30, 20, density=0.2, format="csr", random_state=1)
    row, col, val, m, n = _coo_triplet(X)
    assert (m, n) == (30, 20) and row.dtype == np.int32 and col.dtype == np.int32 and len(row) == X.nnz
    r2, c2, v2, m2, n2 = _coo_triplet((np.array([0, 5]), np.array([1, 2]), np.ones(2)))
    assert (m2, n2) == (6, 3)


def test_bench_generator_and_bytes():
    import bench
    row, col, val = bench.synth_block(500, 300, 6000, seed=2)
    assert len(row) == len(col) == len(val) == 6000
    assert row.min() >= 0 and row.max() < 500 and col.min() >= 0 and col.max() < 300
    assert len(np.unique(row.astype(np.int64) * 300 + col)) == 6000          # no duplicates
    assert (val >= 1).all() and (val == np.ceil(val)).all()
    r2, c2, v2 = bench.synth_block(500, 300, 6000, seed=2)
    assert np.array_equal(row, r2) and np.array_equal(val, v2)               # seeded
    p, i, v = bench.to_csr(row, col, val, 500)
    assert p[-1] == 6000 and (np.diff(p.astype(np.int64)) >= 0).all()
    # 6.92 GB gather + 0.21 GB CSR per half-step at LastFM size
    b = bench.algorithmic_bytes(bench.NNZ, bench.M_USERS, bench.K)
    assert abs(bench.NNZ * bench.K * 8 / 1e9 - 6.92) < 0.01 and 7.3e9 < b < 7.6e9


def test_dealt_item_order():
    """Item blocks equal in rows and balanced in entries (cmfrec_amd/distributed.py, dealt_item_order): a bijection into
    [0, parts * blk), every block within one row of the others, entry counts within a fraction of a per cent of each other on a
    heavily skewed popularity (contiguous nnz-balanced blocks would be far from equal in rows there), deterministic."""
    from cmfrec_amd.distributed import balanced_boundaries, dealt_item_order
    rng = np.random.default_rng(3)
    for n, parts in ((1000003, 8), (5001, 2), (7, 4), (64, 8)):
        counts = (rng.zipf(1.2, size=n) % 100000).astype(np.int64)
        ids, blk = dealt_item_order(counts, parts)
        assert blk == -(-n // parts) and ids.min() >= 0 and ids.max() < parts * blk and len(np.unique(ids)) == n
        rows = np.bincount(ids // blk, minlength=parts)
        assert rows.max() - rows.min() <= 1
        nnz = np.bincount(ids // blk, weights=counts, minlength=parts)
        if n > 1000:
            assert nnz.max() - nnz.min() <= 0.005 * nnz.sum()
            cb = np.diff(balanced_boundaries(counts, parts))
            assert cb.max() > 1.01 * cb.min() or parts == 1          # what the renumbering is for
        ids2, _ = dealt_item_order(counts, parts)
        assert np.array_equal(ids, ids2)
        # the most popular items lead their blocks
        top = np.argsort(-counts, kind="stable")[:parts]
        assert sorted(ids[top] % blk) == [0] * parts
