import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def make_coo(m, n, nnz, seed, counts=True, dtype=np.float64, heavy_row=None, empty_rows=()):
    """Seeded random COO without duplicates; optional heavy row / empty rows (reference tests use
    tiny random problems too, test_math/test_implicit.py:7-27)."""
    rng = np.random.default_rng(seed)
    lin = rng.choice(m * n, size=nnz, replace=False)
    row = (lin // n).astype(np.int32)
    col = (lin % n).astype(np.int32)
    if heavy_row is not None:
        r, cnt = heavy_row
        cnt = min(cnt, n)
        keep = row != r
        row, col = row[keep], col[keep]
        hc = rng.choice(n, size=cnt, replace=False).astype(np.int32)
        row = np.concatenate([row, np.full(cnt, r, np.int32)])
        col = np.concatenate([col, hc])
        perm = rng.permutation(len(row))
        row, col = row[perm], col[perm]
    for r in empty_rows:
        keep = row != r
        row, col = row[keep], col[keep]
    nn = len(row)
    val = np.ceil(rng.lognormal(1, 1, nn)) if counts else 0.5 * rng.integers(1, 11, nn)
    return row, col, val.astype(dtype)


def rel_err(a, b):
    e = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
              / max(float(np.abs(np.asarray(b, np.float64)).max()), 1e-300))
    log = os.environ.get("CMFREC_TEST_RELERR_LOG")       # tools/gpu/r02_ao.sh: the worst case behind each tolerance
    if log:
        with open(log, "a") as f:
            f.write("%s %s %.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], np.asarray(a).dtype, e))
    return e


def row_rel_err(a, b, floor=1e-6):
    """Per-row criterion: for every row, max-abs error over the row's max-abs value; rows whose entries are all
    below `floor` x the matrix maximum are measured against that floor instead (a row of zeros has no scale of its own).
    Returns (worst ratio, index of the worst row).  The global rel_err above lets a row whose entries are 1e-3 of the matrix
    maximum be 10 % off at a 1e-4 tolerance; this one does not."""
    a = np.asarray(a, np.float64).reshape(len(a), -1)
    b = np.asarray(b, np.float64).reshape(len(b), -1)
    scale = np.maximum(np.abs(b).max(axis=1), floor * max(float(np.abs(b).max()), 1e-300))
    e = np.abs(a - b).max(axis=1) / scale
    worst = int(np.argmax(e))
    log = os.environ.get("CMFREC_TEST_RELERR_LOG")
    if log:
        with open(log, "a") as f:
            f.write("%s %s row %.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], np.asarray(a).dtype, e[worst]))
    return float(e[worst]), worst


@pytest.fixture(scope="session")
def oracles():
    from oracle.bindings import Oracle
    return {np.float64: Oracle(np.float64), np.float32: Oracle(np.float32)}
