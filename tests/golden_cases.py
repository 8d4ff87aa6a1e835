"""Shared drivers for the golden-fixture tests: every case runs an engine (oracle on CPU, HIP on
GPU) on the stored inputs and returns (name, got, expected) triples."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = {np.float64: "f64", np.float32: "f32"}


def load(name, dtype):
    return np.load(os.path.join(GOLD, "%s_%s.npz" % (name, TAGS[dtype])))


def csr_from(g, O):
    return O.coo_to_csr_and_csc(g["row"], g["col"], g["val"], int(g["m"]), int(g["n"]))


def implicit_cases(g, O, op_implicit, modes=("cg", "pcg", "chol")):
    csr, _ = csr_from(g, O)
    for k in (8, 50, 64):
        for mode in modes:
            A = g["A0_k%d" % k].copy()
            op_implicit(A, g["B_k%d" % k], csr, float(g["lam"]), use_cg=mode != "chol", precondition_cg=mode == "pcg",
                        max_cg_steps=3)
            yield "implicit %s k=%d" % (mode, k), A, g["A_%s_k%d" % (mode, k)]


def explicit_cases(g, O, op_explicit, modes=("cg", "pcg", "chol")):
    csr, _ = csr_from(g, O)
    for k in (51, 17):
        for mode in modes:
            A = g["A0_k%d" % k].copy()
            op_explicit(A, g["B_k%d" % k], csr, float(g["lam"]), lam_last=float(g["lam_last"]), k=k, scale_lam=True,
                        use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
            yield "explicit %s k=%d" % (mode, k), A, g["A_%s_k%d" % (mode, k)]


def collective_cases(g, O, op_collective):
    csr, _ = csr_from(g, O)
    for ci in (0, 1):
        p, k, ku, ki, km, sls = [int(x) for x in g["cfg_%d" % ci]]
        A = g["A0_%d" % ci].copy()
        op_collective(A, g["B_%d" % ci], g["C_%d" % ci], csr, g["U_%d" % ci], float(g["lam"]), w_user=float(g["w_user"]),
                      lam_last=float(g["lam_last"]), k=k, k_main=km, k_user=ku, k_item=ki, scale_lam=True,
                      scale_lam_sideinfo=bool(sls))
        kA = ku + k + km
        yield "collective cfg %d" % ci, A[:, :kA], g["A_%d" % ci][:, :kA]


def frob(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def maxrel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
