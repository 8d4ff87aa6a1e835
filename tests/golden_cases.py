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


# ---- result metrics of the reference's benchmarks (SURVEY.md 8c G7, 8d "results parity") ----
def rmse(A, B, biasA, biasB, glob_mean, urow, icol, y):
    """Root mean squared error of  glob_mean + biasA[u] + biasB[i] + A_u . B_i  on held-out pairs
    (reference predict_multiple, src/common.c:5098-5106; benchmark_explicit_cmfrec.ipynb)."""
    pred = np.einsum("ij,ij->i", np.asarray(A, np.float64)[urow], np.asarray(B, np.float64)[icol]) + float(glob_mean)
    if biasA is not None and len(biasA):
        pred = pred + np.asarray(biasA, np.float64)[urow]
    if biasB is not None and len(biasB):
        pred = pred + np.asarray(biasB, np.float64)[icol]
    return float(np.sqrt(np.mean((pred - np.asarray(y, np.float64)) ** 2)))


def precision_at_k(A, B, train_row, train_col, test_row, test_col, k=10):
    """P@k as the reference's implicit benchmark computes it (benchmark_implicit_cmfrec.ipynb, cell 3): for every
    user with held-out items, rank all items by A_u . B_i, drop the user's training items, take the top k and
    count the fraction that are held-out items of that user; mean over those users."""
    A = np.asarray(A, np.float64); B = np.asarray(B, np.float64)
    n = B.shape[0]
    seen = {}
    for u, i in zip(train_row, train_col):
        seen.setdefault(int(u), []).append(int(i))
    held = {}
    for u, i in zip(test_row, test_col):
        held.setdefault(int(u), set()).add(int(i))
    hits = []
    for u, items in sorted(held.items()):
        s = B @ A[u]
        s[seen.get(u, [])] = -np.inf
        top = np.argpartition(-s, min(k, n - 1))[:k]
        hits.append(len(items.intersection(top.tolist())) / float(k))
    return float(np.mean(hits))
