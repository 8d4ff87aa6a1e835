import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc
from oracle.bindings import Reference
dt = np.float64
R = Reference(dt)
d = gc.weights_problem(dt)
for opts in (dict(use_cg=False, center=False), dict(use_cg=False, user_bias=False, item_bias=False, center=True),
             dict(use_cg=False, user_bias=True, item_bias=False, center=False), dict(use_cg=False, user_bias=False, item_bias=True, center=False),
             dict(use_cg=True, finalize_chol=False, user_bias=True, item_bias=False, center=False)):
    for niter in (1,):
        for sides in ("U", "I"):
            dd = dict(d)
            o = dict(opts); o["niter"] = niter
            A0, B0 = gc._impf_start(d, o)
            U = d["U"] if "U" in sides else None; II = d["I"] if "I" in sides else None
            kw = dict(use_cg=o.pop("use_cg"), finalize_chol=o.pop("finalize_chol", False)); o.pop("niter")
            r = R.fit_collective_explicit_als(A0.copy(), B0.copy(), d["row"], d["col"], d["ratings"], d["k"], biasA=d["bA"].copy(), biasB=d["bB"].copy(),
                                              lam=0.3, niter=niter, U=U, II=II, w_user=2.0, w_item=0.5, nthreads=2, weight=d["W"], **kw, **o)
            from cmfrec_amd import CMF
            mdl = CMF(k=d["k"], lambda_=0.3, niter=niter, w_user=2.0, w_item=0.5, use_float=False, precompute_for_predictions=False, nthreads=1, **kw, **o)
            mdl.fit((d["row"], d["col"], d["ratings"]), U=U, I=II, shape=(d["m"], d["n"]), W=d["W"], A0=A0, B0=B0, biasA0=d["bA"], biasB0=d["bB"])
            e = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if b is not None and np.size(b) else -1
            print(opts, niter, sides, "A %.2e B %.2e C %.2e D %.2e" % (e(mdl.A_, r["A"]), e(mdl.B_, r["B"]), e(mdl.C_, r["C"]), e(mdl.D_, r["D"])),
                  "worst rows A", np.argsort(-np.abs(mdl.A_ - r["A"]).max(1))[:3], "B", np.argsort(-np.abs(mdl.B_ - r["B"]).max(1))[:3])
