"""Repeats the heavy-row parity case of tests/test_gpu_operators.py and prints the relative error per configuration
(looking for run-to-run variation).  python tools/microbench/heavy_rows_relerr.py gram|stream REPS"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
os.environ["CMFREC_HIP_VH"] = sys.argv[1]
from conftest import make_coo, rel_err
from oracle import bindings as ob
from cmfrec_amd import ops
def run(dtype, implicit, k, O):
    m, n = 60, 5000
    row, col, val = make_coo(m, n, 12000, 41, counts=implicit, dtype=dtype, heavy_row=(3, 4500), empty_rows=(8,))
    rng = np.random.default_rng(4)
    extra_r = np.concatenate([np.full(2500, 10, np.int32), np.full(900, 11, np.int32)])
    keep = (row != 10) & (row != 11)
    extra_c = np.concatenate([rng.choice(n, 2500, replace=False), rng.choice(n, 900, replace=False)]).astype(np.int32)
    extra_v = (np.ceil(rng.lognormal(1, 1, 3400)) if implicit else 0.5 * rng.integers(1, 11, 3400)).astype(dtype)
    row = np.concatenate([row[keep], extra_r]); col = np.concatenate([col[keep], extra_c]); val = np.concatenate([val[keep], extra_v])
    csr, _ = O.coo_to_csr_and_csc(row, col, val, m, n)
    A0 = (rng.standard_normal((m, k)) * 0.05).astype(dtype)
    B = (rng.standard_normal((n, k)) * 0.2).astype(dtype)
    Ah, Ao = A0.copy(), A0.copy()
    if implicit:
        ops.optimizeA_implicit(Ah, B, csr, 4.0); O.optimizeA_implicit(Ao, B, csr, 4.0, nthreads=4)
    else:
        bias = (rng.standard_normal(n) * 0.2).astype(dtype)
        ops.optimizeA_explicit(Ah, B, csr, 0.05, lam_last=0.3, scale_lam=True, bias_sub=bias)
        csr_b = (csr[0], csr[1], (csr[2] - bias[csr[1]]).astype(dtype))
        O.optimizeA_explicit(Ao, B, csr_b, 0.05, lam_last=0.3, scale_lam=True, nthreads=4)
    per_row = np.abs(Ah - Ao).max(axis=1) / max(np.abs(Ao).max(), 1e-30)
    return rel_err(Ah, Ao), int(per_row.argmax()), float(per_row.max())
O32 = ob.Oracle(np.float32)
for rep in range(int(sys.argv[2])):
    out = []
    for implicit in (True, False):
        for k in (50, 7, 33):
            e, r, pr = run(np.float32, implicit, k, O32)
            out.append("%s k=%d: %.2e (row %d)" % ("impl" if implicit else "expl", k, e, r))
    print(sys.argv[1], " | ".join(out), flush=True)
