"""GPU: the kernel-selecting environment switches that are read once per process ("Run-time switches") -- the generic CG kernel, the
workgroup-per-row Cholesky kernel, two rows per wavefront for the shortest rows, the split-row boundary, the library GEMMs.
Each runs the same three small fits in a child process with the switch set; the factors must agree with the default paths'
to rounding (the switches select another kernel for the same row systems, never another model)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from conftest import make_coo
from cmfrec_amd import CMF, CMF_implicit
out = {}
K, NNZ = int(sys.argv[2]), int(sys.argv[3])
for uf, tag in ((False, "f64"), (True, "f32")):
    dt = np.float32 if uf else np.float64
    m, n = 900, 700
    row, col, val = make_coo(m, n, NNZ, 3, heavy_row=(5, 600), empty_rows=(7, 11), dtype=dt)
    a = CMF_implicit(k=K, lambda_=3.0, niter=3, use_cg=True, finalize_chol=False, use_float=uf, random_state=7).fit((row, col, val), shape=(m, n))
    out["icg_A_" + tag] = a.A_; out["icg_B_" + tag] = a.B_
    rng = np.random.default_rng(2)
    II = rng.standard_normal((n, 6)).astype(dt)
    row, col, val = make_coo(m, n, NNZ, 4, counts=False, heavy_row=(5, 600), dtype=dt)
    b = CMF(k=K, lambda_=0.5, niter=2, use_cg=False, use_float=uf, random_state=9, nthreads=1, precompute_for_predictions=False).fit(
        (row, col, val), I=II, shape=(m, n))
    out["ech_A_" + tag] = b.A_; out["ech_B_" + tag] = b.B_; out["ech_D_" + tag] = b.D_
    c = CMF(k=K, lambda_=0.5, niter=2, use_cg=True, finalize_chol=False, use_float=uf, random_state=9, nthreads=1,
            precompute_for_predictions=False).fit((row, col, val), shape=(m, n))
    out["ecg_A_" + tag] = c.A_; out["ecg_B_" + tag] = c.B_
np.savez(sys.argv[1], **out)
"""


def _run(tmp_path, name, env, k=20, nnz=30000):
    path = str(tmp_path / (name + ".npz"))
    e = dict(os.environ)
    for name_ in ("CMFREC_HIP_CG_KERNEL", "CMFREC_HIP_CHOL", "CMFREC_HIP_VH_MIN", "CMFREC_HIP_NT_SPLIT"):
        e.pop(name_, None)
    e.update(env)
    code = CHILD % dict(root=ROOT, tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code, path, str(k), str(nnz)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


@pytest.fixture(scope="module")
def default_fits(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("switches"), "default", {})


@pytest.mark.parametrize("env", [{"CMFREC_HIP_CG_KERNEL": "generic"}, {"CMFREC_HIP_CHOL": "rows"},
                                 {"CMFREC_HIP_VH_MIN": "400"}],
                         ids=lambda e: "-".join("%s=%s" % kv for kv in e.items()))
def test_process_wide_switch_agrees_with_default(default_fits, tmp_path, env):
    got = _run(tmp_path, "alt", env)
    assert set(got.files) == set(default_fits.files)
    for key in got.files:
        a, b = got[key], default_fits[key]
        tol = 2e-4 if key.endswith("f32") else 1e-10
        assert np.isfinite(a).all(), key
        assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30), (key, float(np.abs(a - b).max()), float(np.abs(b).max()))


def test_launches_by_tile_size_are_bit_identical(tmp_path):
    """Double precision, 24 < k <= 56: a length bin of the CG row kernels runs as two launches by tile size (the rows of at most
    48 W entries on the build that fits three wavefronts per SIMD; cg_kernels.hpp NTSEL, device.hpp launch_cg_bin_by_tile).  A row's
    arithmetic depends on its own length only, so CMFREC_HIP_NT_SPLIT=0 -- one launch per bin -- gives the same bits, in both CG
    models (implicit with the Gramian, explicit without), rows in every length bin."""
    a = _run(tmp_path, "split", {}, k=50, nnz=90000)
    b = _run(tmp_path, "one", {"CMFREC_HIP_NT_SPLIT": "0"}, k=50, nnz=90000)
    assert set(a.files) == set(b.files)
    for key in a.files:
        if key.startswith("ech"):
            continue                      # (closed form: not these kernels)
        assert np.isfinite(a[key]).all(), key
        assert np.array_equal(a[key], b[key]), key

