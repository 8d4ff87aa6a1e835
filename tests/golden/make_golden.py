#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref, david-cortes/cmfrec compiled
from /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Fixtures are data: seeded inputs + the reference's outputs.  No reference
source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_coo          # noqa: E402
import golden_cases as gc              # noqa: E402
from oracle.bindings import Reference  # noqa: E402


ONLY = sys.argv[1:]          # e.g. `make_golden.py g19`: write only the fixtures whose name starts with one of these


def save(name, **arrs):
    if ONLY and not any(name.startswith(o + "_") for o in ONLY):
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-40s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def main():
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        R = Reference(dt)
        # ---- G36: NA_as_zero for the main matrix with observation weights AND sparse side information ----
        # (the fixtures of round 6 are checked against a fresh run of the compiled reference by tests/test_oracle_vs_ref.py.  A note for
        #  whoever adds cases: the reference writes out of bounds on some combinations -- a nearly complete dense X with dense side
        #  information, a dense X with weights and side information -- and then a LATER block of this script dies ('SystemError: unknown
        #  opcode', a segmentation fault in the garbage collector); golden_cases.py names the combinations found so far)
        out = {}
        d = gc.weights_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.NAZ_WEIGHTED_SPARSE_SIDE_CASES):
            r = gc.naz_weighted_sparse_side_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g36_na_as_zero_weighted_sparse_side_" + tag, **out)

        # ---- G6: COO -> CSR/CSC ordering, global mean, bias initialisation ----
        m, n = 120, 90
        row, col, val = make_coo(m, n, 1500, 101, counts=False, dtype=dt, heavy_row=(3, 60), empty_rows=(7,))
        csr, csc = R.coo_to_csr_and_csc(row, col, val, m, n)
        gm, Xc = R.calc_mean_and_center(row, col, val, m, n, nthreads=1)
        csr_c, csc_c = R.coo_to_csr_and_csc(row, col, Xc, m, n)
        bA, bB = R.initialize_biases_twosided(m, n, csr_c, csc_c, 0.05, 0.05, True)
        save("g6_prep_" + tag, row=row, col=col, val=val, m=m, n=n, csr_p=csr[0], csr_i=csr[1], csr_v=csr[2],
             csc_p=csc[0], csc_i=csc[1], csc_v=csc[2], glob_mean=gm, val_centered=Xc, biasA=bA, biasB=bB,
             lam_bias=0.05)

        # ---- G1: optimizeA_implicit CG(3) / PCG(3) / Cholesky ----
        m, n = 300, 200
        row, col, val = make_coo(m, n, 6000, 102, dtype=dt, heavy_row=(11, 150), empty_rows=(5, 250))
        csr, _ = R.coo_to_csr_and_csc(row, col, val, m, n)
        out = dict(row=row, col=col, val=val, m=m, n=n, lam=4.0)
        for k in (8, 50, 64):
            rng = np.random.default_rng(k)
            A0 = (rng.standard_normal((m, k)) * 0.05).astype(dt); B = (rng.standard_normal((n, k)) * 0.2).astype(dt)
            out["A0_k%d" % k] = A0; out["B_k%d" % k] = B
            for mode in ("cg", "pcg", "chol"):
                A = A0.copy()
                BtB = R.optimizeA_implicit(A, B, csr, 4.0, nthreads=2, use_cg=mode != "chol",
                                           precondition_cg=mode == "pcg", max_cg_steps=3, return_BtB=True)
                out["A_%s_k%d" % (mode, k)] = A
                if mode == "cg":
                    out["BtB_k%d" % k] = np.triu(BtB)
        save("g1_implicit_" + tag, **out)

        # ---- G2: optimizeA Case 4 (explicit sparse), scale_lam, lam_last != lam, lda = k + pad ----
        row, col, val = make_coo(m, n, 6000, 103, counts=False, dtype=dt, heavy_row=(2, 120), empty_rows=(9,))
        csr, _ = R.coo_to_csr_and_csc(row, col, val, m, n)
        out = dict(row=row, col=col, val=val, m=m, n=n, lam=0.05, lam_last=0.3)
        for k in (51, 17):
            rng = np.random.default_rng(100 + k)
            A0 = (rng.standard_normal((m, k + 1)) * 0.05).astype(dt); B = (rng.standard_normal((n, k + 2)) * 0.2).astype(dt)
            out["A0_k%d" % k] = A0; out["B_k%d" % k] = B
            for mode in ("cg", "pcg", "chol"):
                A = A0.copy()
                R.optimizeA(A, B, csr=csr, lam=0.05, lam_last=0.3, k=k, scale_lam=True, nthreads=2,
                            use_cg=mode != "chol", precondition_cg=mode == "pcg", max_cg_steps=3)
                out["A_%s_k%d" % (mode, k)] = A
        save("g2_explicit_" + tag, **out)

        # ---- G3: optimizeA_collective general branch, dense U, Cholesky ----
        out = dict(row=row, col=col, val=val, m=m, n=n, lam=0.05, lam_last=0.2, w_user=0.5)
        rng = np.random.default_rng(7)
        for ci, (p, k, ku, ki, km, sls) in enumerate(((16, 12, 0, 0, 0, False), (32, 10, 2, 3, 1, True))):
            kA, kB = ku + k + km, ki + k + km
            Bm = (rng.standard_normal((n, kB + 1)) * 0.3).astype(dt); Cm = (rng.standard_normal((p, ku + k)) * 0.3).astype(dt)
            U = rng.standard_normal((m, p)).astype(dt)
            A = rng.standard_normal((m, kA + 1)).astype(dt)
            out.update({"B_%d" % ci: Bm, "C_%d" % ci: Cm, "U_%d" % ci: U, "A0_%d" % ci: A.copy(),
                        "cfg_%d" % ci: np.array([p, k, ku, ki, km, int(sls)])})
            R.optimizeA_collective(A, Bm, Cm, csr, U, 0.05, w_user=0.5, lam_last=0.2, k=k, k_main=km, k_user=ku,
                                   k_item=ki, scale_lam=True, scale_lam_sideinfo=sls, nthreads=2)
            out["A_%d" % ci] = A
        save("g3_collective_" + tag, **out)

        # ---- G4: optimizeA Case 1, do_B (the C / D update) ----
        rng = np.random.default_rng(8)
        m_u, p, kc = 250, 16, 20
        U = rng.standard_normal((m_u, p)).astype(dt); Ab = (rng.standard_normal((m_u, kc + 1)) * 0.3).astype(dt)
        Cm = np.zeros((p, kc), dt)
        R.optimizeA(Cm, Ab, Xfull=U, lam=0.7, k=kc, do_B=True, scale_lam=True, full_dense=True, use_cg=False)
        save("g4_dense_full_" + tag, U=U, A_bias=Ab, C=Cm, lam=0.7, kc=kc)

        # ---- G5: whole fits with injected start values ----
        m, n, k = 400, 250, 16
        rng = np.random.default_rng(9)
        row, col, val = make_coo(m, n, 8000, 104, dtype=dt, heavy_row=(0, 200))
        out = dict(row=row, col=col, val=val, m=m, n=n, k=k, lam=5.0, alpha=1.5, niter=5)
        A0 = (rng.standard_normal((m, k)) * 0.01).astype(dt)
        out["A0"] = A0
        for mode in ("cg", "chol", "cgfin"):
            A, B = A0.copy(), np.zeros((n, k), dt)
            R.fit_collective_implicit_als(A, B, row, col, val, k, lam=5.0, alpha=1.5, niter=5, nthreads=2,
                                          use_cg=mode != "chol", finalize_chol=mode == "cgfin")
            out["A_" + mode] = A; out["B_" + mode] = B
        save("g5_fit_implicit_" + tag, **out)

        row, col, val = make_coo(m, n, 8000, 105, counts=False, dtype=dt, heavy_row=(1, 180))
        out = dict(row=row, col=col, val=val, m=m, n=n, k=k, lam=0.05, niter=4)
        A0 = (rng.standard_normal((m, k)) * 0.01).astype(dt)
        bA0 = (rng.standard_normal(m) * 0.1).astype(dt); bB0 = (rng.standard_normal(n) * 0.1).astype(dt)
        out.update(A0=A0, biasA0=bA0, biasB0=bB0)
        for mode in ("cg", "chol", "cgfin"):
            A, B = A0.copy(), np.zeros((n, k), dt)
            r = R.fit_collective_explicit_als(A, B, row, col, val, k, biasA=bA0.copy(), biasB=bB0.copy(), lam=0.05,
                                              scale_lam=True, niter=4, nthreads=2, use_cg=mode != "chol",
                                              finalize_chol=mode == "cgfin")
            out.update({"A_" + mode: A, "B_" + mode: B, "biasA_" + mode: r["biasA"], "biasB_" + mode: r["biasB"],
                        "glob_mean": r["glob_mean"]})
        save("g5_fit_explicit_" + tag, **out)

        p, q = 12, 9
        U = (rng.standard_normal((m, p)) + 1).astype(dt); II = (rng.standard_normal((n, q)) - 2).astype(dt)
        ku, ki, km = 2, 3, 1
        A0 = (rng.standard_normal((m, ku + k + km)) * 0.01).astype(dt); B0 = (rng.standard_normal((n, ki + k + km)) * 0.01).astype(dt)
        A, B = A0.copy(), B0.copy()
        r = R.fit_collective_explicit_als(A, B, row, col, val, k, lam=0.05, scale_lam=True, scale_lam_sideinfo=True,
                                          niter=3, nthreads=2, use_cg=False, U=U, II=II, k_user=ku, k_item=ki, k_main=km,
                                          w_user=0.5, w_item=2.0)
        save("g5_fit_sideinfo_" + tag, row=row, col=col, val=val, m=m, n=n, k=k, U=U, II=II, A0=A0, B0=B0, A=A, B=B,
             C=r["C"], D=r["D"], biasA=r["biasA"], biasB=r["biasB"], glob_mean=r["glob_mean"],
             U_colmeans=r["U_colmeans"], I_colmeans=r["I_colmeans"], cfg=np.array([ku, ki, km]))

        # ---- implicit model with dense side information, Cholesky (collective.c:5971-6244) ----
        row, col, val = make_coo(m, n, 8000, 106, dtype=dt, heavy_row=(2, 150), empty_rows=(7, 390))
        m_u, n_i = m - 40, n                       # users beyond m_u carry no side information
        U = (rng.standard_normal((m_u, p)) + 1).astype(dt); II = (rng.standard_normal((n_i, q)) - 2).astype(dt)
        A0 = (rng.standard_normal((m, ku + k + km)) * 0.01).astype(dt); B0 = (rng.standard_normal((n, ki + k + km)) * 0.01).astype(dt)
        A, B = A0.copy(), B0.copy()
        r = R.fit_collective_implicit_als(A, B, row, col, val, k, lam=3.0, alpha=2.0, niter=3, nthreads=2, use_cg=False,
                                          U=U, II=II, k_user=ku, k_item=ki, k_main=km, w_main=0.5, w_user=4.0, w_item=0.8)
        assert r["ret"] == 0
        save("g8_fit_implicit_sideinfo_" + tag, row=row, col=col, val=val, m=m, n=n, k=k, U=U, II=II, A0=A0, B0=B0, A=A, B=B,
             C=r["C"], D=r["D"], U_colmeans=r["U_colmeans"], I_colmeans=r["I_colmeans"], cfg=np.array([ku, ki, km]))

        # ---- precompute_for_predictions epilogue (collective.c:8936-9249, 10056-10115) ----
        # implicit + side information, last step CG, w_user != 1: pins quirk Q9 (unweighted C^T C in BeTBe)
        row, col, val = make_coo(m, n, 8000, 107, dtype=dt)
        A0 = (rng.standard_normal((m, ku + k + km)) * 0.01).astype(dt); B0 = (rng.standard_normal((n, ki + k + km)) * 0.01).astype(dt)
        U = (rng.standard_normal((m, p)) + 1).astype(dt); II = (rng.standard_normal((n, q)) - 2).astype(dt)
        out = dict(row=row, col=col, val=val, m=m, n=n, k=k, U=U, II=II, A0=A0, B0=B0, cfg=np.array([ku, ki, km]))
        for mode in ("cg", "chol"):
            A, B = A0.copy(), B0.copy()
            r = R.fit_collective_implicit_als(A, B, row, col, val, k, lam=3.0, alpha=2.0, niter=2, nthreads=2,
                                              use_cg=mode == "cg", U=U, II=II, k_user=ku, k_item=ki, k_main=km,
                                              w_main=0.5, w_user=4.0, w_item=0.8, precompute=True)
            assert r["ret"] == 0
            out.update({"B_" + mode: B, "C_" + mode: r["C"], "BtB_" + mode: np.triu(r["pre"]["BtB"]),
                        "BeTBe_" + mode: np.triu(r["pre"]["BeTBe"]), "BeTBeChol_" + mode: np.triu(r["pre"]["BeTBeChol"])})
        save("g9_precompute_implicit_" + tag, **out)
        row, col, val = make_coo(m, n, 8000, 108, counts=False, dtype=dt)
        A, B = A0.copy(), B0.copy()
        r = R.fit_collective_explicit_als(A, B, row, col, val, k, lam=0.05, scale_lam=True, scale_lam_sideinfo=True, niter=2,
                                          nthreads=2, use_cg=False, U=U, II=II, k_user=ku, k_item=ki, k_main=km,
                                          w_user=0.5, w_item=2.0, precompute=True)
        assert r["ret"] == 0
        pr = r["pre"]
        save("g9_precompute_explicit_" + tag, row=row, col=col, val=val, m=m, n=n, k=k, U=U, II=II, A0=A0, B0=B0,
             cfg=np.array([ku, ki, km]), B=B, C=r["C"], B_plus_bias=pr["B_plus_bias"], BtB=np.triu(pr["BtB"]),
             TransBtBinvBt=pr["TransBtBinvBt"], BeTBeChol=np.triu(pr["BeTBeChol"]), CtCw=np.triu(pr["CtCw"]),
             TransCtCinvCt=pr["TransCtCinvCt"])

        # ---- G10: result metrics of the benchmarks -- RMSE (explicit) and P@10 (implicit) from the reference's own
        #      15-iteration fits on a train / held-out split (SURVEY.md 8c G7, 8d "results parity") ----
        mm, nn, kk = 600, 400, 16
        mrng = np.random.default_rng(77)
        row, col, val = make_coo(mm, nn, 30000, 109, counts=False, dtype=dt)
        Pt = (mrng.standard_normal((mm, 4)) @ mrng.standard_normal((4, nn)))                 # low-rank signal + noise
        val = (3.0 + Pt[row, col] + 0.3 * mrng.standard_normal(len(row))).astype(dt)
        te = mrng.random(len(row)) < 0.2
        A0 = (mrng.standard_normal((mm, kk)) * 0.01).astype(dt)
        A, B = A0.copy(), np.zeros((nn, kk), dt)
        r = R.fit_collective_explicit_als(A, B, row[~te], col[~te], val[~te], kk, lam=0.05, scale_lam=True, niter=15,
                                          nthreads=2, use_cg=True, finalize_chol=False)
        assert r["ret"] == 0
        out = dict(m=mm, n=nn, k=kk, A0=A0, e_row=row[~te], e_col=col[~te], e_val=val[~te], e_trow=row[te], e_tcol=col[te],
                   e_tval=val[te], rmse=gc.rmse(A, B, r["biasA"], r["biasB"], r["glob_mean"], row[te], col[te], val[te]))
        # implicit data with structure to recover: 8 user clusters, each interacting with its own block of 50 items
        # (95 %) plus a little noise elsewhere, so the held-out items of a user are predictable from the others
        row, col, val = make_coo(mm, nn, 180000, 110, counts=True, dtype=dt)
        own = (col // 50) == (row % 8)
        keep = own | (mrng.random(len(row)) < 0.02)
        row, col, val = row[keep], col[keep], val[keep]
        te = mrng.random(len(row)) < 0.2
        A, B = A0.copy(), np.zeros((nn, kk), dt)
        rc = R.fit_collective_implicit_als(A, B, row[~te], col[~te], val[~te], kk, lam=5.0, niter=15, nthreads=2, use_cg=True)
        assert rc == 0
        out.update(i_row=row[~te], i_col=col[~te], i_val=val[~te], i_trow=row[te], i_tcol=col[te],
                   p_at_10=gc.precision_at_k(A, B, row[~te], col[~te], row[te], col[te], 10))
        print("   reference metrics: RMSE %.6f   P@10 %.6f" % (out["rmse"], out["p_at_10"]))
        save("g10_metrics_" + tag, **out)

        # ---- G11: factors of new rows (factors_collective_{explicit,implicit}_multiple) ----
        out = {}
        for k in (6, 50):
            d = gc.new_rows_problem(dt, k)
            for name, kind, kw in gc.new_rows_cases(d):
                A, bA = gc.run_new_rows(R, kind, dict(kw, nthreads=2))
                key = "k%d_%s" % (k, name.split()[0])
                out["A_" + key] = A
                if bA is not None:
                    out["biasA_" + key] = bA
        save("g11_new_rows_" + tag, **out)

        # ---- G22: factors of new rows under an L1 penalty (solve_elasticnet behind factors_collective_*_multiple) ----
        out = {}
        for k in (6, 50):
            d = gc.new_rows_problem(dt, k)
            for name, kind, kw in gc.new_rows_l1_cases(d):
                A, bA = gc.run_new_rows(R, kind, dict(kw, nthreads=2))
                key = "k%d_%s" % (k, name.split()[0])
                out["A_" + key] = A
                if bA is not None:
                    out["biasA_" + key] = bA
        save("g22_new_rows_l1_" + tag, **out)

        # ---- G12: fits with sparse side information (Cholesky updates) ----
        out = {}
        d = gc.sparse_sideinfo_problem(dt)
        for ci, (name, implicit, which, sl, sls) in enumerate(gc.SPARSE_SIDE_CASES):
            r = gc.sparse_sideinfo_reference(R, d, implicit, which, sl, sls)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.SPARSE_SIDE_CG_CASES):
            r = gc.sparse_sideinfo_reference(R, d, implicit, which, sl, sls, solver=solver)
            for key, v in r.items():
                if v is not None:
                    out["g%d_%s" % (ci, key)] = v
        save("g12_sparse_sideinfo_" + tag, **out)

        # ---- G13: non-negative factors ----
        out = {}
        d = gc.nonneg_problem(dt)
        for ci, (name, implicit, side, opts) in enumerate(gc.NONNEG_CASES):
            r = gc.nonneg_reference(R, d, implicit, side, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g13_nonneg_" + tag, **out)

        # ---- G14: implicit features of the explicit model ----
        out = {}
        d = gc.nonneg_problem(dt)
        for ci, (name, side, opts) in enumerate(gc.IMPLICIT_FEATS_CASES):
            r = gc.implicit_feats_reference(R, d, side, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g14_implicit_feats_" + tag, **out)

        # ---- G15: per-matrix penalties (lam_unique / l1_lam_unique) ----
        out = {}
        d = gc.nonneg_problem(dt)
        for ci, (name, implicit, side, opts) in enumerate(gc.LAM_UNIQUE_CASES):
            r = gc.lam_unique_reference(R, d, implicit, side, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g15_lam_unique_" + tag, **out)

        # ---- G16: dense side information with missing values ----
        out = {}
        d = gc.nan_side_problem(dt)
        for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.NAN_SIDE_CASES):
            r = gc.nan_side_reference(R, d, implicit, which, sl, sls, solver=solver)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g16_nan_side_" + tag, **out)

        # ---- G17: observation weights of the explicit model ----
        out = {}
        d = gc.weights_problem(dt)
        for ci, (name, side, opts) in enumerate(gc.WEIGHT_CASES):
            r = gc.weights_reference(R, d, side, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g17_weights_" + tag, **out)

        # ---- G18: NA_as_zero for the main matrix ----
        out = {}
        d = gc.naz_problem(dt)
        for ci, (name, opts) in enumerate(gc.NAZ_CASES):
            r = gc.naz_reference(R, d, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g18_na_as_zero_" + tag, **out)

        # ---- G24: NA_as_zero for the main matrix WITH observation weights ----
        out = {}
        d = gc.naz_weighted_problem(dt)
        for ci, (name, opts) in enumerate(gc.NAZ_WEIGHTED_CASES):
            r = gc.naz_weighted_reference(R, d, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g24_na_as_zero_weighted_" + tag, **out)

        # ---- G25: NA_as_zero for the main matrix together with SPARSE side information ----
        out = {}
        d = gc.naz_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.NAZ_SPARSE_SIDE_CASES):
            r = gc.naz_sparse_side_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g25_na_as_zero_sparse_side_" + tag, **out)

        # ---- G26: NA_as_zero for the main matrix together with implicit features ----
        out = {}
        d = gc.naz_problem(dt)
        for ci, (name, opts) in enumerate(gc.NAZ_IMPF_CASES):
            r = gc.naz_impf_reference(R, d, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g26_na_as_zero_implicit_features_" + tag, **out)

        # ---- G27: NA_as_zero for the main matrix with the matrices for predictions ----
        out = {}
        d = gc.naz_problem(dt)
        for ci, (name, opts) in enumerate(gc.NAZ_PRE_CASES):
            r = gc.naz_pre_reference(R, d, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g27_na_as_zero_precompute_" + tag, **out)

        # ---- G28: NA_as_zero for the main matrix with observation weights AND dense side information ----
        out = {}
        d = gc.naz_weighted_problem(dt)
        for ci, (name, sides, opts) in enumerate(gc.NAZ_WEIGHTED_SIDE_CASES):
            r = gc.naz_side_reference(R, d, sides, opts, weights=True)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g28_na_as_zero_weighted_sideinfo_" + tag, **out)

        # ---- G35: ... under use_cg ----
        out = {}
        d = gc.naz_weighted_problem(dt)
        for ci, (name, sides, opts) in enumerate(gc.NAZ_WEIGHTED_SIDE_CG_CASES):
            r = gc.naz_side_reference(R, d, sides, opts, weights=True)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g35_na_as_zero_weighted_sideinfo_cg_" + tag, **out)

        # ---- G29: the stand-alone prediction matrices (precompute_collective_explicit / _implicit) ----
        out = {}
        d = gc.precompute_problem(dt)
        for ci, (name, opts) in enumerate(gc.PRECOMPUTE_EXPLICIT_CASES):
            for key, v in gc.precompute_explicit_call(R.lib, d, opts, dt).items():
                out["e%d_%s" % (ci, key)] = v
        for ci, (name, opts) in enumerate(gc.PRECOMPUTE_IMPLICIT_CASES):
            for key, v in gc.precompute_implicit_call(R.lib, d, opts, dt).items():
                out["i%d_%s" % (ci, key)] = v
        save("g29_precompute_standalone_" + tag, **out)

        # ---- G30: the per-user ranking under the reference's names (topN_old_collective_explicit / _implicit) ----
        out = {}
        d = gc.topn_problem(dt)
        for ci, (name, opts) in enumerate(gc.TOPN_CASES):
            for key, v in gc.topn_call(R.lib, d, opts, dt).items():
                out["c%d_%s" % (ci, key)] = v
        save("g30_topn_old_" + tag, **out)

        # ---- G31: observation weights together with SPARSE side information ----
        out = {}
        d = gc.weights_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.WEIGHT_SPARSE_SIDE_CASES):
            r = gc.weights_sparse_side_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g31_weights_sparse_side_" + tag, **out)

        # ---- G32: implicit features together with SPARSE side information ----
        out = {}
        d = gc.weights_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.IMPF_SPARSE_SIDE_CASES):
            r = gc.impf_sparse_side_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g32_implicit_features_sparse_side_" + tag, **out)

        # ---- G34: NA_as_zero for the main matrix together with sparse side information under use_cg ----
        out = {}
        d = gc.naz_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.NAZ_SPARSE_SIDE_CG_CASES):
            r = gc.naz_sparse_side_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g34_na_as_zero_sparse_side_cg_" + tag, **out)

        # ---- G37: NA_as_zero for the main matrix with implicit features AND side information ----
        out = {}
        for ci, (name, kind, which, opts) in enumerate(gc.NAZ_IMPF_SIDE_CASES):
            r = gc.naz_impf_side_reference(R, kind, which, opts, dt)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g37_na_as_zero_implicit_features_sideinfo_" + tag, **out)

        # ---- G38: observation weights together with implicit features ----
        out = {}
        d = gc.weights_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.WEIGHT_IMPF_CASES):
            r = gc.weights_impf_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g38_weights_implicit_features_" + tag, **out)

        # ---- G39: NA_as_zero for the main matrix with observation weights AND implicit features ----
        out = {}
        d = gc.weights_sparse_side_problem(dt)
        for ci, (name, which, opts) in enumerate(gc.NAZ_WEIGHTED_IMPF_CASES):
            r = gc.naz_weighted_impf_reference(R, d, which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g39_na_as_zero_weighted_implicit_features_" + tag, **out)

        # ---- G19: dense X with NaN for the missing entries (optimizeA Cases 1-2) ----
        out = {}
        for ci, (name, variant, opts) in enumerate(gc.DENSE_CASES):
            r = gc.dense_reference(R, gc.dense_problem(dt, variant), opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g19_dense_X_" + tag, **out)

        # ---- G33: dense X together with side information ----
        out = {}
        for ci, (name, variant, which, opts) in enumerate(gc.DENSE_SIDE_CASES):
            r = gc.dense_side_reference(R, gc.dense_side_problem(dt, variant), which, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g33_dense_X_sideinfo_" + tag, **out)

        # ---- G20: NA_as_zero_X together with dense side information (shared block matrix, collective.c:5607-5617) ----
        out = {}
        d = gc.naz_problem(dt)
        for ci, (name, sides, opts) in enumerate(gc.NAZ_SIDE_CASES):
            r = gc.naz_side_reference(R, d, sides, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g20_na_as_zero_sideinfo_" + tag, **out)

        # ---- G21: NA_as_zero_U / NA_as_zero_I (sparse side information whose absent entries are zeros) ----
        out = {}
        d = gc.sparse_sideinfo_problem(dt)
        for ci, (name, implicit, which, sl, sls, solver) in enumerate(gc.NAZ_UI_CASES):
            r = gc.naz_ui_reference(R, d, implicit, which, sl, sls, solver)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g21_na_as_zero_UI_" + tag, **out)

        # ---- G23: the global mean under nthreads >= 8 (sum / count; weighted: unweighted sum / sum of weights) ----
        out = {}
        d = gc.weights_problem(dt)
        for ci, (name, weighted, opts) in enumerate(gc.NTHREADS8_CASES):
            r = gc.nthreads8_reference(R, d, weighted, opts)
            for key, v in r.items():
                if v is not None:
                    out["c%d_%s" % (ci, key)] = v
        save("g23_nthreads8_mean_" + tag, **out)

        # ---- RNG streams of the reference (pins the start-value generator) ----
        out = {}
        for seed in (1, 123):
            for size in (1000, 2 ** 18 + 1000):
                for normal in (True, False):
                    a, _ = R.random_parallel(size, 0, seed, normal)
                    out["seed%d_size%d_%s" % (seed, size, "normal" if normal else "unif")] = a[:64]
        save("g7_rng_" + tag, **out)


if __name__ == "__main__":
    main()
