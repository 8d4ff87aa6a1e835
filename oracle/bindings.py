"""ctypes bindings for the CPU checkers -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

* ``Oracle(dtype)``  -> oracle/libcmf_oracle_{double,float}.so   (our C restatement, cmf_oracle.c)
* ``Reference(dtype)`` -> oracle/_ref/libcmfrec_ref_{double,float}.so (the real cmfrec, compiled from
  /root/reference by oracle/Makefile; present only where it was built and shipped)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

c_real = {np.float64: C.c_double, np.float32: C.c_float}


def _np_dtype(dtype):
    dtype = np.dtype(dtype).type
    if dtype not in (np.float64, np.float32):
        raise ValueError("dtype must be float64 or float32")
    return dtype


def _suffix(dtype):
    return "double" if _np_dtype(dtype) is np.float64 else "float"


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _coo_args(coo, dtype):
    """(row ptr, col ptr, val ptr, nnz, keep-alive) of a COO side-information triplet, or NULLs."""
    if coo is None:
        return (None, None, None, C.c_size_t(0), None)
    r = np.ascontiguousarray(coo[0], np.int32); c = np.ascontiguousarray(coo[1], np.int32)
    v = np.ascontiguousarray(coo[2], dtype)
    return (_ptr(r), _ptr(c), _ptr(v), C.c_size_t(len(v)), (r, c, v))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def build_ref():
    """Builds oracle/_ref from /root/reference (only possible where that tree exists)."""
    if not os.path.isdir("/root/reference/src"):
        return False
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return True


def _preload_openblas():
    import scipy
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)),
                                  "scipy.libs", "libscipy_openblas*.so"))
    for lib in libs:
        C.CDLL(lib, mode=C.RTLD_GLOBAL)


class Oracle:
    def __init__(self, dtype=np.float64):
        self.dtype = _np_dtype(dtype)
        path = os.path.join(_HERE, "libcmf_oracle_%s.so" % _suffix(dtype))
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        assert self.lib.oracle_sizeof_real() == np.dtype(self.dtype).itemsize
        self.real = c_real[self.dtype]

    def _r(self, x):
        return self.real(float(x))

    def set_nonneg(self, nonneg=False, nonneg_C=False, nonneg_D=False, max_cd_steps=100):
        """Non-negativity option of the following fits (solve_nonneg instead of the Cholesky solve; no CG)."""
        self.lib.oracle_set_nonneg(C.c_bool(nonneg), C.c_bool(nonneg_C), C.c_bool(nonneg_D), C.c_int(max_cd_steps))

    def set_l1(self, l1_lam, max_cd_steps=100):
        """L1 penalty of the following fits (solve_elasticnet / shifted solve_nonneg; no CG)."""
        self.lib.oracle_set_l1(self._r(l1_lam), C.c_int(max_cd_steps))

    def set_lam_unique(self, lam_unique=None, l1_lam_unique=None):
        """Per-matrix penalties of the following fits (user bias, item bias, A, B, C, D); None switches them off.  The
        explicit fit expects them divided by w_main already, the implicit one divides entries 2..5 itself."""
        cv = lambda a: None if a is None else np.ascontiguousarray(a, self.dtype)
        self._lam6, self._l16 = cv(lam_unique), cv(l1_lam_unique)
        self.lib.oracle_set_lam_unique(_ptr(self._lam6), _ptr(self._l16))

    def set_scale_bias_const(self, on):
        """scale_bias_const of the following explicit fits."""
        self.lib.oracle_set_scale_bias_const(C.c_bool(on))

    def set_nonneg_now(self, on, max_cd_steps=100):
        """The same for operator-level calls outside a fit."""
        self.lib.oracle_set_nonneg_now(C.c_bool(on), C.c_int(max_cd_steps))

    def coo_to_csr_and_csc(self, row, col, val, m, n):
        nnz = len(val)
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        csr_p = np.zeros(m + 1, np.uint64); csr_i = np.zeros(nnz, np.int32); csr_v = np.zeros(nnz, self.dtype)
        csc_p = np.zeros(n + 1, np.uint64); csc_i = np.zeros(nnz, np.int32); csc_v = np.zeros(nnz, self.dtype)
        self.lib.oracle_coo_to_csr_and_csc(_ptr(row), _ptr(col), _ptr(val), C.c_int(m), C.c_int(n),
                                           C.c_size_t(nnz), _ptr(csr_p), _ptr(csr_i), _ptr(csr_v),
                                           _ptr(csc_p), _ptr(csc_i), _ptr(csc_v))
        return (csr_p, csr_i, csr_v), (csc_p, csc_i, csc_v)

    def optimizeA_implicit(self, A, B, csr, lam, k=None, nthreads=1, use_cg=True,
                           precondition_cg=False, max_cg_steps=3, return_BtB=False):
        """A [m, lda] in/out (modified in place), B [n, ldb]."""
        assert A.dtype == self.dtype and B.dtype == self.dtype and A.flags.c_contiguous and B.flags.c_contiguous
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        p, i, v = csr
        BtB = np.zeros((k, k), self.dtype) if return_BtB else None
        self.lib.oracle_optimizeA_implicit(_ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb),
                                           C.c_int(m), C.c_int(n), C.c_int(k), _ptr(p), _ptr(i), _ptr(v),
                                           self._r(lam), C.c_int(nthreads), C.c_bool(use_cg),
                                           C.c_bool(precondition_cg), C.c_int(max_cg_steps), _ptr(BtB))
        return BtB

    def optimizeA_explicit(self, A, B, csr, lam, lam_last=None, k=None, scale_lam=False,
                           scale_bias_const=False, nthreads=1, use_cg=True, precondition_cg=False,
                           max_cg_steps=3, weight=None, wsum=None):
        """weight: observation weights in the entry order of ``csr``; wsum: per-row lambda multipliers under scale_lam (the
        driver's wsumA; None: the row's own sum, common.c:696-707)."""
        assert A.dtype == self.dtype and B.dtype == self.dtype and A.flags.c_contiguous and B.flags.c_contiguous
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        lam_last = lam if lam_last is None else lam_last
        p, i, v = csr
        if weight is not None:
            weight = np.ascontiguousarray(weight, self.dtype)
            wsum = None if wsum is None else np.ascontiguousarray(wsum, self.dtype)
            self.lib.oracle_set_row_weights(_ptr(weight), _ptr(wsum))
        self.lib.oracle_optimizeA_explicit(_ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb),
                                           C.c_int(m), C.c_int(n), C.c_int(k), _ptr(p), _ptr(i), _ptr(v),
                                           self._r(lam), self._r(lam_last), C.c_bool(scale_lam),
                                           C.c_bool(scale_bias_const), C.c_int(nthreads), C.c_bool(use_cg),
                                           C.c_bool(precondition_cg), C.c_int(max_cg_steps))

    def optimizeA_naz_weighted(self, A, B, csr, weight, lam, lam_last=None, k=None, wsum=None, bias_BtX=None, bias_X=None,
                               bias_X_glob=0.0, scale_lam=False, scale_bias_const=False, use_cg=False, precondition_cg=False,
                               max_cg_steps=3, nthreads=1):
        """optimizeA Case 4 with NA_as_zero and observation weights (common.c:3209-3302 + the NA_as_zero / weight branches of
        factors_closed_form): weight in the entry order of ``csr``."""
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        lam_last = lam if lam_last is None else lam_last
        p, i, v = csr
        weight = np.ascontiguousarray(weight, self.dtype)
        wsum = None if wsum is None else np.ascontiguousarray(wsum, self.dtype)
        bias_BtX = None if bias_BtX is None else np.ascontiguousarray(bias_BtX, self.dtype)
        bias_X = None if bias_X is None else np.ascontiguousarray(bias_X, self.dtype)
        self.lib.oracle_optimizeA_naz_weighted(_ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb), C.c_int(m), C.c_int(n), C.c_int(k),
                                               _ptr(p), _ptr(i), _ptr(v), _ptr(weight), _ptr(wsum), _ptr(bias_BtX), _ptr(bias_X),
                                               self._r(bias_X_glob), self._r(lam), self._r(lam_last), C.c_bool(scale_lam),
                                               C.c_bool(scale_bias_const), C.c_bool(use_cg), C.c_bool(precondition_cg),
                                               C.c_int(max_cg_steps), C.c_int(nthreads))

    def optimizeA_dense_full(self, A, B, Xfull, lam, lam_last=None, k=None, do_B=False,
                             scale_lam=False, nthreads=1):
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        lam_last = lam if lam_last is None else lam_last
        ldX = Xfull.shape[1]
        self.lib.oracle_optimizeA_dense_full(_ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb),
                                             C.c_int(m), C.c_int(n), C.c_int(k), _ptr(Xfull),
                                             C.c_size_t(ldX), C.c_bool(do_B), self._r(lam),
                                             self._r(lam_last), C.c_bool(scale_lam), C.c_int(nthreads))

    def optimizeA_collective_chol(self, A, B, Cm, csr, U, lam, w_user=1.0, lam_last=None, k=None,
                                  k_main=0, k_user=0, k_item=0, scale_lam=False,
                                  scale_lam_sideinfo=False, nthreads=1, m_u=None):
        m, lda = A.shape
        n, ldb = B.shape
        p = Cm.shape[0]
        lam_last = lam if lam_last is None else lam_last
        m_u = U.shape[0] if m_u is None else m_u
        pp, i, v = csr
        self.lib.oracle_optimizeA_collective_chol(
            _ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb), _ptr(Cm),
            C.c_int(m), C.c_int(m_u), C.c_int(n), C.c_int(p),
            C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            _ptr(pp), _ptr(i), _ptr(v), _ptr(U), self._r(lam), self._r(w_user), self._r(lam_last),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_int(nthreads))

    def optimizeA_collective_sparse(self, A, B, Cm, csr, U_csr, lam, w_user=1.0, lam_last=None, k=None, k_main=0,
                                    k_user=0, k_item=0, scale_lam=False, scale_lam_sideinfo=False, implicit=False,
                                    nthreads=1, use_cg=False, precondition_cg=False, max_cg_steps=3):
        m, lda = A.shape
        n, ldb = B.shape
        lam_last = lam if lam_last is None else lam_last
        up, ui, uv = U_csr
        if use_cg:
            self.lib.oracle_optimizeA_collective_sparse_cg(
                _ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb), _ptr(Cm), C.c_int(m), C.c_int(len(up) - 1), C.c_int(n),
                C.c_int(Cm.shape[0]), C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
                _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]), _ptr(up), _ptr(ui), _ptr(uv),
                self._r(lam), self._r(w_user), self._r(lam_last), C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo),
                C.c_bool(implicit), C.c_int(max_cg_steps), C.c_bool(precondition_cg), C.c_int(nthreads))
            return
        self.lib.oracle_optimizeA_collective_sparse_chol(
            _ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb), _ptr(Cm), C.c_int(m), C.c_int(len(up) - 1), C.c_int(n),
            C.c_int(Cm.shape[0]), C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]), _ptr(up), _ptr(ui), _ptr(uv),
            self._r(lam), self._r(w_user), self._r(lam_last), C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo),
            C.c_bool(implicit), C.c_int(nthreads))

    def _factors_multiple_sparse(self, implicit, B, row, col, val, m, k, Cm, U_coo, biasB=None, glob_mean=0.0,
                                 user_bias=False, lam=1.0, lam_bias=None, alpha=1.0, k_main=0, k_user=0, k_item=0,
                                 scale_lam=False, w_main=1.0, w_user=1.0, nthreads=1):
        """New rows with SPARSE side information: per row this is the collective closed form with the row's attributes as
        rank-1 terms (collective_factors_warm / _cold with u_vec_sp -> collective_closed_form_block[_implicit],
        collective.c:3555-4087), i.e. oracle_optimizeA_collective_sparse_chol on [B | 1] after the preprocessing of
        factors_collective_*_single (:10575-10863).  Not for scale_lam_sideinfo / bias-less scale_lam (quirks Q11, Q12)."""
        n = B.shape[0]
        ur, uc, uv, m_u, p = U_coo
        mm = max(m, m_u)
        val = np.asarray(val, self.dtype)
        if implicit:
            x = val * self.dtype(alpha) if alpha != 1 else val
            lam_x = lam                                              # :11270-11280: the X block keeps the lam of the call
        else:
            x = val - ((np.asarray(biasB, self.dtype)[col] + self.dtype(glob_mean)) if biasB is not None else self.dtype(glob_mean))
        csr, _ = self.coo_to_csr_and_csc(row, col, x.astype(self.dtype), mm, n)
        ucsr, _ = self.coo_to_csr_and_csc(ur, uc, uv, m_u, p)
        lam_bias = lam if lam_bias is None else lam_bias
        if w_main != 1:
            lam, lam_bias, w_user = lam / w_main, lam_bias / w_main, w_user / w_main
        ub = 1 if (user_bias and not implicit) else 0
        Bp = np.ascontiguousarray(np.hstack([B, np.ones((n, 1), self.dtype)]) if ub else B, self.dtype)
        A = np.zeros((mm, k_user + k + k_main + ub), self.dtype)
        if implicit and lam_x != lam:
            raise NotImplementedError("w_main != 1 with sparse side information: the X block keeps the unscaled lam")
        self.optimizeA_collective_sparse(A, Bp, Cm, csr, ucsr, lam, w_user=w_user, lam_last=lam_bias if ub else lam, k=k,
                                         k_main=k_main + ub, k_user=k_user, k_item=k_item, scale_lam=scale_lam,
                                         implicit=implicit, nthreads=nthreads)
        if ub:
            return np.ascontiguousarray(A[:, :-1]), A[:, -1].copy()
        return A, None

    def fit_als_sparse_sideinfo(self, A, B, row, col, val, k, implicit, U_coo=None, I_coo=None, Cm=None, Dm=None,
                                biasA=None, biasB=None, user_bias=False, item_bias=False, center=False, lam=1.0, alpha=1.0,
                                scale_lam=False, scale_lam_sideinfo=False, k_main=0, k_user=0, k_item=0, w_main=1.0,
                                w_user=1.0, w_item=1.0, niter=3, nthreads=1, use_cg=False, max_cg_steps=3,
                                precondition_cg=False, finalize_chol=False, NA_as_zero_X=False):
        """Whole fit with sparse side information (U_coo / I_coo = (row, col, val, rows, cols)).  NA_as_zero_X: explicit model, closed form."""
        if NA_as_zero_X:
            self.lib.oracle_set_sparse_fit_NA_as_zero_X(C.c_bool(True))
        m, n = A.shape[0], B.shape[0]
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        su, si = _coo_args(U_coo, self.dtype), _coo_args(I_coo, self.dtype)
        m_u, p = (0, 0) if U_coo is None else (U_coo[3], U_coo[4])
        n_i, q = (0, 0) if I_coo is None else (I_coo[3], I_coo[4])
        if p and Cm is None: Cm = np.zeros((p, k_user + k), self.dtype)
        if q and Dm is None: Dm = np.zeros((q, k_item + k), self.dtype)
        biasA = np.zeros(m, self.dtype) if biasA is None else biasA
        biasB = np.zeros(n, self.dtype) if biasB is None else biasB
        gm = np.zeros(1, self.dtype)
        ret = self.lib.oracle_fit_als_sparse_sideinfo(
            C.c_bool(implicit), _ptr(biasA), _ptr(biasB), _ptr(A), _ptr(B), _ptr(Cm), _ptr(Dm), _ptr(gm),
            C.c_int(m), C.c_int(n), C.c_int(k), _ptr(row), _ptr(col), _ptr(val), C.c_size_t(len(val)),
            C.c_bool(user_bias), C.c_bool(item_bias), C.c_bool(center), self._r(lam), self._r(alpha),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo),
            su[0], su[1], su[2], su[3], C.c_int(m_u), C.c_int(p), si[0], si[1], si[2], si[3], C.c_int(n_i), C.c_int(q),
            C.c_int(k_main), C.c_int(k_user), C.c_int(k_item), self._r(w_main), self._r(w_user), self._r(w_item),
            C.c_int(niter), C.c_int(nthreads), C.c_bool(use_cg), C.c_int(max_cg_steps), C.c_bool(precondition_cg),
            C.c_bool(finalize_chol))
        return dict(ret=ret, A=A, B=B, C=Cm, D=Dm, biasA=biasA, biasB=biasB, glob_mean=gm[0])

    def calc_mean_and_center(self, X, nthreads=1):
        self.lib.oracle_calc_mean_and_center.restype = self.real
        return self.lib.oracle_calc_mean_and_center(_ptr(X), C.c_size_t(len(X)), C.c_int(nthreads))

    def initialize_biases_twosided(self, m, n, csr, csc, lam_user, lam_item, scale_lam):
        biasA = np.zeros(m, self.dtype); biasB = np.zeros(n, self.dtype)
        self.lib.oracle_initialize_biases_twosided(C.c_int(m), C.c_int(n), _ptr(csr[0]), _ptr(csr[1]),
                                                   _ptr(csr[2]), _ptr(csc[0]), _ptr(csc[1]), _ptr(csc[2]),
                                                   self._r(lam_user), self._r(lam_item),
                                                   C.c_bool(scale_lam), _ptr(biasA), _ptr(biasB))
        return biasA, biasB

    def factors_explicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, biasB=None,
                                  glob_mean=0.0, user_bias=False, lam=1.0, lam_bias=None, k_main=0, k_user=0,
                                  k_item=0, scale_lam=False, scale_lam_sideinfo=False, scale_bias_const=False,
                                  scaling_biasA=1.0, w_main=1.0, w_user=1.0, nthreads=1, TransCtCinvCt=None, U_coo=None):
        """Restatement of factors_collective_explicit_multiple; same arguments as Reference.factors_explicit_multiple."""
        if U_coo is not None:
            return self._factors_multiple_sparse(False, B, row, col, val, m, k, Cm, U_coo, biasB=biasB, glob_mean=glob_mean,
                                                 user_bias=user_bias, lam=lam, lam_bias=lam_bias, k_main=k_main, k_user=k_user,
                                                 k_item=k_item, scale_lam=scale_lam, w_main=w_main, w_user=w_user, nthreads=nthreads)
        n = B.shape[0]
        csr, _ = self.coo_to_csr_and_csc(row, col, val, m, n)
        m_u = 0 if U is None else U.shape[0]
        p = 0 if U is None else U.shape[1]
        mm = max(m, m_u)
        A = np.full((mm, k_user + k + k_main), np.nan, self.dtype)
        biasA = np.full(mm, np.nan, self.dtype) if user_bias else None
        Uc = None if U is None else np.ascontiguousarray(U, self.dtype)
        self.lib.oracle_factors_explicit_multiple(
            _ptr(A), _ptr(biasA), C.c_int(m), _ptr(Uc), C.c_int(m_u), C.c_int(p), _ptr(Cm),
            self._r(glob_mean), _ptr(biasB), _ptr(U_colmeans), _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]),
            C.c_int(n), _ptr(B), C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            self._r(lam), self._r(lam if lam_bias is None else lam_bias),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_bool(scale_bias_const), self._r(scaling_biasA),
            self._r(w_main), self._r(w_user), _ptr(TransCtCinvCt), C.c_int(nthreads))
        return A, biasA

    def factors_implicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, lam=1.0,
                                  alpha=1.0, k_main=0, k_user=0, k_item=0, w_main=1.0, w_user=1.0,
                                  w_main_multiplier=1.0, apply_log_transf=False, nthreads=1, BtB=None, U_coo=None):
        """Restatement of factors_collective_implicit_multiple; same arguments as Reference.factors_implicit_multiple."""
        if U_coo is not None:
            return self._factors_multiple_sparse(True, B, row, col, val, m, k, Cm, U_coo, lam=lam, alpha=alpha, k_main=k_main,
                                                 k_user=k_user, k_item=k_item, w_main=w_main * w_main_multiplier, w_user=w_user,
                                                 nthreads=nthreads)[0]
        n = B.shape[0]
        csr, _ = self.coo_to_csr_and_csc(row, col, val, m, n)
        m_u = 0 if U is None else U.shape[0]
        p = 0 if U is None else U.shape[1]
        A = np.full((max(m, m_u), k_user + k + k_main), np.nan, self.dtype)
        Uc = None if U is None else np.ascontiguousarray(U, self.dtype)
        self.lib.oracle_factors_implicit_multiple(
            _ptr(A), C.c_int(m), _ptr(Uc), C.c_int(m_u), C.c_int(p), _ptr(Cm), _ptr(U_colmeans),
            _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]), C.c_int(n), _ptr(B),
            C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            self._r(lam), self._r(alpha), self._r(w_main), self._r(w_user), self._r(w_main_multiplier),
            C.c_bool(apply_log_transf), _ptr(BtB), C.c_int(nthreads))
        return A

    def fit_implicit_als(self, A, B, row, col, val, lam=1.0, alpha=1.0, apply_log_transf=False,
                         niter=10, nthreads=1, use_cg=True, max_cg_steps=3, precondition_cg=False,
                         finalize_chol=False):
        m, k = A.shape
        n = B.shape[0]
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        return self.lib.oracle_fit_implicit_als(_ptr(A), _ptr(B), C.c_int(m), C.c_int(n), C.c_int(k),
                                                _ptr(row), _ptr(col), _ptr(val), C.c_size_t(len(val)),
                                                self._r(lam), self._r(alpha), C.c_bool(apply_log_transf),
                                                C.c_int(niter), C.c_int(nthreads), C.c_bool(use_cg),
                                                C.c_int(max_cg_steps), C.c_bool(precondition_cg),
                                                C.c_bool(finalize_chol))

    def fit_implicit_als_sideinfo(self, A, B, row, col, val, k, Cm=None, Dm=None, U=None, II=None, lam=1.0,
                                  alpha=1.0, apply_log_transf=False, k_main=0, k_user=0, k_item=0, w_main=1.0,
                                  w_user=1.0, w_item=1.0, niter=10, nthreads=1, use_cg=False, max_cg_steps=3,
                                  precondition_cg=False, finalize_chol=False, m=None, n=None):
        m = A.shape[0] if m is None else m; n = B.shape[0] if n is None else n
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        m_u, p = (0, 0) if U is None else U.shape
        n_i, q = (0, 0) if II is None else II.shape
        Ucm = np.zeros(max(p, 1), self.dtype); Icm = np.zeros(max(q, 1), self.dtype)
        if U is not None and Cm is None:
            Cm = np.zeros((p, k_user + k), self.dtype)
        if II is not None and Dm is None:
            Dm = np.zeros((q, k_item + k), self.dtype)
        ret = self.lib.oracle_fit_implicit_als_sideinfo(
            _ptr(A), _ptr(B), _ptr(Cm), _ptr(Dm), _ptr(Ucm), _ptr(Icm), C.c_int(m), C.c_int(n), C.c_int(k),
            _ptr(row), _ptr(col), _ptr(val), C.c_size_t(len(val)),
            _ptr(U), C.c_int(m_u), C.c_int(p), _ptr(II), C.c_int(n_i), C.c_int(q),
            C.c_int(k_main), C.c_int(k_user), C.c_int(k_item), self._r(w_main), self._r(w_user), self._r(w_item),
            self._r(lam), self._r(alpha), C.c_bool(apply_log_transf), C.c_int(niter), C.c_int(nthreads),
            C.c_bool(use_cg), C.c_int(max_cg_steps), C.c_bool(precondition_cg), C.c_bool(finalize_chol))
        return dict(ret=ret, A=A, B=B, C=Cm, D=Dm, U_colmeans=Ucm, I_colmeans=Icm)

    def set_closed_form_rows(self, maskA=None, maskB=None):
        """Per-row solver choice of the NEXT explicit fit (dense X, optimizeA Case 2): non-zero = closed form inside a CG
        half-step.  The arrays must stay alive until that fit returns."""
        self._cfA = None if maskA is None else np.ascontiguousarray(maskA, np.uint8)
        self._cfB = None if maskB is None else np.ascontiguousarray(maskB, np.uint8)
        self.lib.oracle_set_closed_form_rows(_ptr(self._cfA), _ptr(self._cfB))

    def set_lambda_multipliers(self, multA=None, multB=None):
        """Per-row lambda multipliers of the NEXT explicit fit under scale_lam (dense X: n for a row that misses few entries, its
        present entries otherwise); that fit must be given (unit) weights.  The arrays must stay alive until it returns."""
        self._lmA = None if multA is None else np.ascontiguousarray(multA, self.dtype)
        self._lmB = None if multB is None else np.ascontiguousarray(multB, self.dtype)
        self.lib.oracle_set_lambda_multipliers(_ptr(self._lmA), _ptr(self._lmB))

    def set_sideinfo_dense_rules(self, cfC=None, multC=None, cfD=None, multD=None):
        """Per-attribute rules of the NEXT fit_als_sparse_sideinfo call when it stands for DENSE side information with NaN: closed
        form inside a CG half-step (cf*, one byte per attribute) and the lambda multipliers under scale_lam (mult*).  The arrays
        must stay alive until that fit returns."""
        self._scf = [None if a is None else np.ascontiguousarray(a, np.uint8) for a in (cfC, cfD)]
        self._smu = [None if a is None else np.ascontiguousarray(a, self.dtype) for a in (multC, multD)]
        self.lib.oracle_set_sideinfo_dense_rules(_ptr(self._scf[0]), _ptr(self._smu[0]), _ptr(self._scf[1]), _ptr(self._smu[1]))

    def set_zero_rows(self, rowsA=None, rowsB=None):
        """Rows of A / B the NEXT fit sets to zero after their update (NA_as_zero_U / _I: rows with neither an entry of X nor of
        the side information, which the reference does not solve).  The arrays must stay alive until that fit returns."""
        self._zrA = None if rowsA is None else np.ascontiguousarray(rowsA, np.int32)
        self._zrB = None if rowsB is None else np.ascontiguousarray(rowsB, np.int32)
        self.lib.oracle_set_zero_rows(_ptr(self._zrA), C.c_int(0 if self._zrA is None else len(self._zrA)),
                                      _ptr(self._zrB), C.c_int(0 if self._zrB is None else len(self._zrB)))

    def fit_explicit_als(self, A, B, row, col, val, k, biasA=None, biasB=None, Cm=None, Dm=None,
                         U=None, II=None, user_bias=True, item_bias=True, center=True, lam=10.0,
                         scale_lam=False, scale_lam_sideinfo=False, k_main=0, k_user=0, k_item=0,
                         w_user=1.0, w_item=1.0, niter=10, nthreads=1, use_cg=True, max_cg_steps=3,
                         precondition_cg=False, finalize_chol=True, init_biases=False, m=None, n=None,
                         add_implicit_features=False, w_implicit=1.0, w_main=1.0, weight=None, NA_as_zero_X=False):
        m = A.shape[0] if m is None else m; n = B.shape[0] if n is None else n       # shape of X (A, B may have more rows)
        if NA_as_zero_X:
            self.lib.oracle_set_fit_NA_as_zero_X(C.c_bool(True))
        if weight is not None:
            weight = np.ascontiguousarray(weight, self.dtype)
            assert len(weight) == len(val)
            self.lib.oracle_set_fit_weights(_ptr(weight))
        if w_main != 1.0:                                                            # collective.c:7497-7521
            lam, w_user, w_item, w_implicit = lam / w_main, w_user / w_main, w_item / w_main, w_implicit / w_main
        Ai = np.zeros((A.shape[0], k + k_main), self.dtype) if add_implicit_features else None
        Bi = np.zeros((B.shape[0], k + k_main), self.dtype) if add_implicit_features else None
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        biasA = np.zeros(A.shape[0], self.dtype) if biasA is None else biasA
        biasB = np.zeros(B.shape[0], self.dtype) if biasB is None else biasB
        glob_mean = np.zeros(1, self.dtype)
        m_u, p = (0, 0) if U is None else U.shape
        n_i, q = (0, 0) if II is None else II.shape
        Ucm = np.zeros(max(p, 1), self.dtype); Icm = np.zeros(max(q, 1), self.dtype)
        if U is not None and Cm is None:
            Cm = np.zeros((p, k_user + k), self.dtype)
        if II is not None and Dm is None:
            Dm = np.zeros((q, k_item + k), self.dtype)
        ret = self.lib.oracle_fit_explicit_als_implicit_features(
            _ptr(biasA), _ptr(biasB), _ptr(A), _ptr(B), _ptr(Cm), _ptr(Dm), _ptr(Ai), _ptr(Bi), self._r(w_implicit),
            _ptr(glob_mean), _ptr(Ucm),
            _ptr(Icm), C.c_int(m), C.c_int(n), C.c_int(k), _ptr(row), _ptr(col), _ptr(val),
            C.c_size_t(len(val)), C.c_bool(user_bias), C.c_bool(item_bias), C.c_bool(center),
            self._r(lam), C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo),
            _ptr(U), C.c_int(m_u), C.c_int(p), _ptr(II), C.c_int(n_i), C.c_int(q),
            C.c_int(k_main), C.c_int(k_user), C.c_int(k_item), self._r(w_user), self._r(w_item),
            C.c_int(niter), C.c_int(nthreads), C.c_bool(use_cg), C.c_int(max_cg_steps),
            C.c_bool(precondition_cg), C.c_bool(finalize_chol), C.c_bool(init_biases))
        return dict(ret=ret, A=A, B=B, C=Cm, D=Dm, biasA=biasA, biasB=biasB, glob_mean=glob_mean[0],
                    U_colmeans=Ucm, I_colmeans=Icm, Ai=Ai, Bi=Bi)


def ref_available(dtype=np.float64):
    return os.path.exists(os.path.join(_HERE, "_ref", "libcmfrec_ref_%s.so" % _suffix(dtype)))


class Reference:
    """The real cmfrec (oracle/_ref), internal operators + the two fit_*_als entry points.

    Signatures: /root/reference/src/cmfrec.h:986-1027 (optimizeA, optimizeA_implicit),
    :1646-1683 (optimizeA_collective), :1851-1921 (fit_collective_{explicit,implicit}_als)."""

    def __init__(self, dtype=np.float64):
        self.dtype = _np_dtype(dtype)
        path = os.path.join(_HERE, "_ref", "libcmfrec_ref_%s.so" % _suffix(dtype))
        if not os.path.exists(path):
            if not build_ref():
                raise FileNotFoundError(path)
        _preload_openblas()
        self.lib = C.CDLL(path)
        self.real = c_real[self.dtype]

    def _r(self, x):
        return self.real(float(x))

    def _scratch(self, nelem):
        return np.zeros(int(nelem), self.dtype)

    def coo_to_csr_and_csc(self, row, col, val, m, n):
        nnz = len(val)
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        csr_p = np.zeros(m + 1, np.uint64); csr_i = np.zeros(nnz, np.int32); csr_v = np.zeros(nnz, self.dtype)
        csc_p = np.zeros(n + 1, np.uint64); csc_i = np.zeros(nnz, np.int32); csc_v = np.zeros(nnz, self.dtype)
        self.lib.coo_to_csr_and_csc(_ptr(row), _ptr(col), _ptr(val), None, C.c_int(m), C.c_int(n),
                                    C.c_size_t(nnz), _ptr(csr_p), _ptr(csr_i), _ptr(csr_v),
                                    _ptr(csc_p), _ptr(csc_i), _ptr(csc_v), None, None, C.c_int(1))
        return (csr_p, csr_i, csr_v), (csc_p, csc_i, csc_v)

    def optimizeA_implicit(self, A, B, csr, lam, k=None, nthreads=1, use_cg=True,
                           precondition_cg=False, max_cg_steps=3, return_BtB=False):
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        p, i, v = csr
        BtB = np.zeros((k, k), self.dtype)
        buf = self._scratch(k * k * (nthreads + 2) + 16 * k * nthreads + 1024)
        self.lib.optimizeA_implicit(_ptr(A), C.c_size_t(lda), _ptr(B), C.c_size_t(ldb),
                                    C.c_int(m), C.c_int(n), C.c_int(k), _ptr(p), _ptr(i), _ptr(v),
                                    self._r(lam), self._r(0.), C.c_int(nthreads), C.c_bool(False),
                                    C.c_bool(use_cg), C.c_bool(precondition_cg), C.c_int(max_cg_steps),
                                    C.c_bool(False), C.c_int(100), _ptr(BtB), _ptr(buf), None)
        return BtB if return_BtB else None

    def optimizeA(self, A, B, csr=None, Xfull=None, lam=1.0, lam_last=None, k=None, scale_lam=False,
                  scale_bias_const=False, do_B=False, nthreads=1, use_cg=True, precondition_cg=False,
                  max_cg_steps=3, full_dense=False, weight=None, wsum=None, NA_as_zero=False, bias_BtX=None, bias_X=None,
                  bias_X_glob=0.0):
        """Case 4 (csr given) or Case 1 (Xfull given, full_dense=True) of optimizeA.  weight (entry order of csr) / wsum: the
        observation weights and the driver's lambda multipliers (wsumA).  NA_as_zero (+ bias_BtX [k], bias_X [n], bias_X_glob):
        sparse X whose absent entries are zeros -- Case 3 without weights, Case 4's weighted branches with them."""
        weight = None if weight is None else np.ascontiguousarray(weight, self.dtype)
        wsum = None if wsum is None else np.ascontiguousarray(wsum, self.dtype)
        bias_BtX = None if bias_BtX is None else np.ascontiguousarray(bias_BtX, self.dtype)
        bias_X = None if bias_X is None else np.ascontiguousarray(bias_X, self.dtype)
        m, lda = A.shape
        n, ldb = B.shape
        k = min(lda, ldb) if k is None else k
        lam_last = lam if lam_last is None else lam_last
        p, i, v = (None, None, None) if csr is None else csr
        ldX = 0 if Xfull is None else Xfull.shape[1]
        filled = C.c_bool(False)
        buf = self._scratch(k * k * (nthreads + 4) + (n + m) * (nthreads + 2) + 16 * k * nthreads + 4096)
        cnt_NA = np.zeros(max(m, n) + 1, np.int32)
        self.lib.optimizeA(_ptr(A), C.c_int(lda), _ptr(B), C.c_int(ldb), C.c_int(m), C.c_int(n), C.c_int(k),
                           _ptr(p), _ptr(i), _ptr(v), _ptr(Xfull), C.c_int(ldX),
                           C.c_bool(full_dense), C.c_bool(False), C.c_bool(full_dense),
                           _ptr(cnt_NA), _ptr(weight), C.c_bool(NA_as_zero),
                           self._r(lam), self._r(lam_last), self._r(0.), self._r(0.),
                           C.c_bool(scale_lam), C.c_bool(scale_bias_const), _ptr(wsum),
                           C.c_bool(do_B), C.c_int(nthreads), C.c_bool(False),
                           C.c_bool(use_cg), C.c_bool(precondition_cg), C.c_int(max_cg_steps),
                           C.c_bool(False), C.c_int(100),
                           None, _ptr(bias_BtX), _ptr(bias_X), self._r(bias_X_glob), None, self._r(1.),
                           C.c_bool(False), None, C.byref(filled), _ptr(buf), None)

    def optimizeA_collective(self, A, B, Cm, csr, U, lam, w_user=1.0, lam_last=None, k=None,
                             k_main=0, k_user=0, k_item=0, scale_lam=False, scale_lam_sideinfo=False,
                             nthreads=1, use_cg=False, m_u=None, U_csr=None, precondition_cg=False):
        """U dense [m_u, p], or sparse: U=None and U_csr=(indptr[m_u+1], indices, values)."""
        m, lda = A.shape
        n, ldb = B.shape
        p = Cm.shape[0]
        lam_last = lam if lam_last is None else lam_last
        if U_csr is not None:
            return self._optimizeA_collective_sparse(A, B, Cm, csr, U_csr, lam, w_user, lam_last, k, k_main, k_user, k_item,
                                                     scale_lam, scale_lam_sideinfo, nthreads, use_cg, precondition_cg)
        m_u = U.shape[0] if m_u is None else m_u
        pp, i, v = csr
        k_totA = k_user + k + k_main
        buf = self._scratch(k_totA * k_totA * (nthreads + 6) + (n + m) * (nthreads + 2) + 4096)
        cnt_NA_u = np.zeros(max(m, m_u) + 1, np.int32)
        flags = [C.c_bool(False) for _ in range(5)]
        self.lib.optimizeA_collective(
            _ptr(A), C.c_int(lda), _ptr(B), C.c_int(ldb), _ptr(Cm), None,
            C.c_int(m), C.c_int(m_u), C.c_int(n), C.c_int(p),
            C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            _ptr(pp), _ptr(i), _ptr(v), None, C.c_int(0),
            C.c_bool(False), C.c_bool(False), C.c_bool(False), None, None, C.c_bool(False),
            None, C.c_int(0), C.c_int(0), C.c_bool(False),
            None, None, None, _ptr(U), _ptr(cnt_NA_u), None,
            C.c_bool(True), C.c_bool(False), C.c_bool(True), C.c_bool(False),
            self._r(lam), self._r(w_user), self._r(1.), self._r(lam_last), self._r(0.), self._r(0.),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_bool(False), None,
            C.c_bool(False), C.c_int(nthreads), C.c_bool(False),
            C.c_bool(use_cg), C.c_int(3), C.c_bool(False), C.c_bool(False), C.c_int(100),
            None, None, None, self._r(0.), C.c_bool(False),
            None, None, None, None, None,
            C.byref(flags[0]), C.byref(flags[1]), C.byref(flags[2]), C.byref(flags[3]), C.byref(flags[4]),
            _ptr(buf), None)

    def factors_explicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, biasB=None,
                                  glob_mean=0.0, user_bias=False, lam=1.0, lam_bias=None, k_main=0, k_user=0,
                                  k_item=0, scale_lam=False, scale_lam_sideinfo=False, scale_bias_const=False,
                                  scaling_biasA=1.0, w_main=1.0, w_user=1.0, nthreads=1, n=None, TransCtCinvCt=None,
                                  U_coo=None, l1_lam=0.0, l1_lam_bias=None):
        """factors_collective_explicit_multiple (src/cmfrec.h:2004-2047, collective.c:10865-11174): sparse X of the
        new rows as COO, optional dense U.  Returns (A, biasA or None)."""
        n_max, ldb = B.shape
        n = n_max if n is None else n
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype).copy()
        nnz = len(val)
        m_u = 0 if U is None else U.shape[0]
        p = 0 if U is None else U.shape[1]
        # sparse side information goes in as CSR: the COO branch of the reference converts with m instead of m_u rows into
        # an m_u+1 allocation (collective.c:10970-10986) and corrupts the heap unless m_u == m
        ucsr = (None, None, None)
        if U_coo is not None:
            m_u, p = U_coo[3], U_coo[4]
            ucsr, _ = self.coo_to_csr_and_csc(U_coo[0], U_coo[1], np.ascontiguousarray(U_coo[2], self.dtype), m_u, p)
        mm = max(m, m_u)
        A = np.full((mm, k_user + k + k_main), np.nan, self.dtype)
        biasA = np.full(mm, np.nan, self.dtype) if user_bias else None
        lam_unique = None
        if lam_bias is not None and lam_bias != lam:
            lam_unique = np.zeros(6, self.dtype); lam_unique[0] = lam_bias; lam_unique[2] = lam
        l1_unique = None
        if l1_lam_bias is not None and l1_lam_bias != l1_lam:
            l1_unique = np.zeros(6, self.dtype); l1_unique[0] = l1_lam_bias; l1_unique[2] = l1_lam
        Uc = None if U is None else np.ascontiguousarray(U, self.dtype).copy()
        rc = self.lib.factors_collective_explicit_multiple(
            _ptr(A), _ptr(biasA), C.c_int(m),
            _ptr(Uc), C.c_int(m_u), C.c_int(p),
            C.c_bool(False), C.c_bool(False), C.c_bool(False),
            None, None, None, C.c_size_t(0), _ptr(ucsr[0]), _ptr(ucsr[1]), _ptr(ucsr[2]),
            None, C.c_int(0), C.c_int(0),
            _ptr(Cm), None,
            self._r(glob_mean), _ptr(biasB), _ptr(U_colmeans),
            _ptr(val), _ptr(row), _ptr(col), C.c_size_t(nnz),
            None, None, None,
            None, C.c_int(n), None,
            _ptr(B), None, C.c_bool(False),
            C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            self._r(lam), _ptr(lam_unique), self._r(l1_lam), _ptr(l1_unique),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_bool(scale_bias_const), self._r(scaling_biasA),
            self._r(w_main), self._r(w_user), self._r(1.),
            C.c_int(n_max), C.c_bool(True),
            None, None, None, None, None, _ptr(TransCtCinvCt), None, None, None,
            C.c_int(nthreads))
        assert rc == 0, rc
        return A, biasA

    def factors_implicit_multiple(self, B, row, col, val, m, k, Cm=None, U=None, U_colmeans=None, lam=1.0,
                                  alpha=1.0, k_main=0, k_user=0, k_item=0, w_main=1.0, w_user=1.0,
                                  w_main_multiplier=1.0, apply_log_transf=False, nthreads=1, BtB=None, U_coo=None, l1_lam=0.0):
        """factors_collective_implicit_multiple (src/cmfrec.h:2048-2071, collective.c:11176-11340)."""
        n, ldb = B.shape
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype).copy()
        nnz = len(val)
        m_u = 0 if U is None else U.shape[0]
        p = 0 if U is None else U.shape[1]
        ucsr = (None, None, None)
        if U_coo is not None:                     # as CSR, see factors_explicit_multiple
            m_u, p = U_coo[3], U_coo[4]
            ucsr, _ = self.coo_to_csr_and_csc(U_coo[0], U_coo[1], np.ascontiguousarray(U_coo[2], self.dtype), m_u, p)
        mm = max(m, m_u)
        A = np.full((mm, k_user + k + k_main), np.nan, self.dtype)
        Uc = None if U is None else np.ascontiguousarray(U, self.dtype).copy()
        rc = self.lib.factors_collective_implicit_multiple(
            _ptr(A), C.c_int(m),
            _ptr(Uc), C.c_int(m_u), C.c_int(p),
            C.c_bool(False), C.c_bool(False),
            None, None, None, C.c_size_t(0), _ptr(ucsr[0]), _ptr(ucsr[1]), _ptr(ucsr[2]),
            _ptr(val), _ptr(row), _ptr(col), C.c_size_t(nnz),
            None, None, None,
            _ptr(B), C.c_int(n), _ptr(Cm), _ptr(U_colmeans),
            C.c_int(k), C.c_int(k_user), C.c_int(k_item), C.c_int(k_main),
            self._r(lam), self._r(l1_lam), self._r(alpha), self._r(w_main), self._r(w_user), self._r(w_main_multiplier),
            C.c_bool(apply_log_transf),
            None, _ptr(BtB), None, None,
            C.c_int(nthreads))
        assert rc == 0, rc
        return A

    def optimizeA_collective_implicit_sparse(self, A, B, Cm, csr, U_csr, lam, w_user=1.0, k=None, k_main=0, k_user=0,
                                             k_item=0, nthreads=1, use_cg=False, precondition_cg=False):
        """optimizeA_collective_implicit (src/cmfrec.h:1442-1467) with sparse U, Cholesky.  A [m, k_user+k+k_main] and
        B [n, k_item+k+k_main] are contiguous (the function takes no leading dimensions)."""
        m, ka = A.shape
        n, kb = B.shape
        assert ka == k_user + k + k_main and kb == k_item + k + k_main
        p = Cm.shape[0]
        up, ui, uv = U_csr
        pp, i, v = csr
        k_totA = k_user + k + k_main
        BtB = np.zeros((k + k_main) ** 2, self.dtype)
        buf = self._scratch(k_totA * k_totA * (nthreads + 8) + (n + m + p) * (nthreads + 2) + 4096)
        flags = [C.c_bool(False) for _ in range(4)]
        self.lib.optimizeA_collective_implicit(
            _ptr(A), _ptr(B), _ptr(Cm), C.c_int(m), C.c_int(len(up) - 1), C.c_int(n), C.c_int(p),
            C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            _ptr(pp), _ptr(i), _ptr(v), _ptr(up), _ptr(ui), _ptr(uv), None, None, None,
            C.c_bool(False), C.c_bool(False), C.c_bool(False),
            self._r(lam), self._r(0.), self._r(w_user), C.c_int(nthreads), C.c_bool(False),
            C.c_bool(use_cg), C.c_int(3), C.c_bool(precondition_cg), C.c_bool(False), C.c_int(100),
            _ptr(BtB), None, None, None, None,
            C.byref(flags[0]), C.byref(flags[1]), C.byref(flags[2]), C.byref(flags[3]),
            _ptr(buf), None)

    def _optimizeA_collective_sparse(self, A, B, Cm, csr, U_csr, lam, w_user, lam_last, k, k_main, k_user, k_item,
                                     scale_lam, scale_lam_sideinfo, nthreads, use_cg=False, precondition_cg=False):
        m, lda = A.shape
        n, ldb = B.shape
        p = Cm.shape[0]
        up, ui, uv = U_csr
        m_u = len(up) - 1
        pp, i, v = csr
        k_totA = k_user + k + k_main
        buf = self._scratch(k_totA * k_totA * (nthreads + 6) + (n + m + p) * (nthreads + 2) + 4096)
        flags = [C.c_bool(False) for _ in range(5)]
        self.lib.optimizeA_collective(
            _ptr(A), C.c_int(lda), _ptr(B), C.c_int(ldb), _ptr(Cm), None,
            C.c_int(m), C.c_int(m_u), C.c_int(n), C.c_int(p),
            C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            _ptr(pp), _ptr(i), _ptr(v), None, C.c_int(0),
            C.c_bool(False), C.c_bool(False), C.c_bool(False), None, None, C.c_bool(False),
            None, C.c_int(0), C.c_int(0), C.c_bool(False),
            _ptr(up), _ptr(ui), _ptr(uv), None, None, None,
            C.c_bool(False), C.c_bool(False), C.c_bool(False), C.c_bool(False),
            self._r(lam), self._r(w_user), self._r(1.), self._r(lam_last), self._r(0.), self._r(0.),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_bool(False), None,
            C.c_bool(False), C.c_int(nthreads), C.c_bool(False),
            C.c_bool(use_cg), C.c_int(3), C.c_bool(precondition_cg), C.c_bool(False), C.c_int(100),
            None, None, None, self._r(0.), C.c_bool(False),
            None, None, None, None, None,
            C.byref(flags[0]), C.byref(flags[1]), C.byref(flags[2]), C.byref(flags[3]), C.byref(flags[4]),
            _ptr(buf), None)

    def calc_mean_and_center(self, row, col, X, m, n, nthreads=1):
        """Returns (glob_mean, centred copy of X)."""
        Xp = (C.c_void_p * 1)(X.ctypes.data)
        gm = np.zeros(1, self.dtype)
        mod_x = C.c_bool(False); mod_xf = C.c_bool(False)
        Xc = X.copy()
        Xp[0] = Xc.ctypes.data
        self.lib.calc_mean_and_center(_ptr(row), _ptr(col), Xp, C.c_size_t(len(X)), None, None,
                                      C.c_int(m), C.c_int(n), None, None, None, None, None, None, None,
                                      C.c_bool(False), C.c_bool(False), C.c_bool(True), C.c_int(nthreads),
                                      _ptr(gm), C.byref(mod_x), C.byref(mod_xf), C.c_bool(True))
        return gm[0], Xc

    def initialize_biases_twosided(self, m, n, csr, csc, lam_user, lam_item, scale_lam, nthreads=1):
        biasA = np.zeros(m, self.dtype); biasB = np.zeros(n, self.dtype)
        self.lib.initialize_biases_twosided(
            None, None, None, None, C.c_int(m), C.c_int(n), C.c_bool(False), C.c_bool(False), C.c_double(0.),
            _ptr(csr[0]), _ptr(csr[1]), _ptr(csr[2]), _ptr(csc[0]), _ptr(csc[1]), _ptr(csc[2]),
            None, None, None, None, self._r(lam_user), self._r(lam_item), C.c_bool(scale_lam),
            None, None, _ptr(biasA), _ptr(biasB), C.c_int(nthreads))
        return biasA, biasB

    def random_parallel(self, sizeA, sizeB, seed, normal, nthreads=1):
        class ArraysToFill(C.Structure):
            _fields_ = [("A", C.c_void_p), ("sizeA", C.c_size_t), ("B", C.c_void_p), ("sizeB", C.c_size_t)]
        A = np.zeros(sizeA, self.dtype); B = np.zeros(max(sizeB, 1), self.dtype)
        arr = ArraysToFill(A.ctypes.data, sizeA, B.ctypes.data if sizeB else None, sizeB)
        self.lib.random_parallel.argtypes = [ArraysToFill, C.c_int, C.c_bool, C.c_int]
        self.lib.random_parallel(arr, seed, normal, nthreads)
        return A, B[:sizeB]

    def fit_collective_implicit_als(self, A, B, row, col, val, k, lam=1.0, alpha=1.0, niter=10,
                                    nthreads=1, use_cg=True, max_cg_steps=3, precondition_cg=False,
                                    finalize_chol=False, reset_values=False, seed=1,
                                    apply_log_transf=False, Cm=None, Dm=None, U=None, II=None,
                                    k_main=0, k_user=0, k_item=0, w_main=1.0, w_user=1.0, w_item=1.0,
                                    precompute=False, m=None, n=None, U_coo=None, I_coo=None, nonneg=False,
                                    nonneg_C=False, nonneg_D=False, max_cd_steps=100, l1_lam=0.0, lam_unique=None,
                                    l1_lam_unique=None, adjust_weight=False, NA_as_zero_U=False, NA_as_zero_I=False):
        """U_coo / I_coo = (row, col, val, rows, cols): sparse side information instead of dense U / II (NA_as_zero_U / _I: its
        absent entries are zeros instead of missing)."""
        m = A.shape[0] if m is None else m; n = B.shape[0] if n is None else n       # shape of X (A, B may have more rows)
        lam6 = None if lam_unique is None else np.ascontiguousarray(lam_unique, self.dtype)
        l16 = None if l1_lam_unique is None else np.ascontiguousarray(l1_lam_unique, self.dtype)
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        wmm = np.zeros(1, self.dtype)
        m_u, p = (0, 0) if U is None else U.shape
        n_i, q = (0, 0) if II is None else II.shape
        su, si = _coo_args(U_coo, self.dtype), _coo_args(I_coo, self.dtype)
        if U_coo is not None: m_u, p = U_coo[3], U_coo[4]
        if I_coo is not None: n_i, q = I_coo[3], I_coo[4]
        Ucm = np.zeros(max(p, 1), self.dtype); Icm = np.zeros(max(q, 1), self.dtype)
        if p and Cm is None:
            Cm = np.zeros((p, k_user + k), self.dtype)
        if q and Dm is None:
            Dm = np.zeros((q, k_item + k), self.dtype)
        kq = k_user + k + k_main
        pre = dict(BtB=np.zeros((k + k_main, k + k_main), self.dtype), BeTBe=np.zeros((kq, kq), self.dtype),
                   BeTBeChol=np.zeros((kq, kq), self.dtype), CtUbias=np.zeros(max(k_user + k, 1), self.dtype)) if precompute else None
        ret = self.lib.fit_collective_implicit_als(
            _ptr(A), _ptr(B), _ptr(Cm), _ptr(Dm), C.c_bool(reset_values), C.c_int(seed), _ptr(Ucm), _ptr(Icm),
            C.c_int(m), C.c_int(n), C.c_int(k), _ptr(row), _ptr(col), _ptr(val), C.c_size_t(len(val)),
            self._r(lam), _ptr(lam6), self._r(l1_lam), _ptr(l16),
            _ptr(U), C.c_int(m_u), C.c_int(p), _ptr(II), C.c_int(n_i), C.c_int(q),
            *su[:4], *si[:4],
            C.c_bool(NA_as_zero_U), C.c_bool(NA_as_zero_I), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            self._r(w_main), self._r(w_user), self._r(w_item), _ptr(wmm),
            self._r(alpha), C.c_bool(adjust_weight), C.c_bool(apply_log_transf),
            C.c_int(niter), C.c_int(nthreads), C.c_bool(False), C.c_bool(False),
            C.c_bool(use_cg), C.c_int(max_cg_steps), C.c_bool(precondition_cg), C.c_bool(finalize_chol),
            C.c_bool(nonneg), C.c_int(max_cd_steps), C.c_bool(nonneg_C), C.c_bool(nonneg_D),
            C.c_bool(precompute), _ptr(pre["BtB"]) if pre else None, _ptr(pre["BeTBe"]) if pre else None,
            _ptr(pre["BeTBeChol"]) if pre else None, _ptr(pre["CtUbias"]) if pre else None)
        if U is None and II is None and not precompute and U_coo is None and I_coo is None and not adjust_weight:
            return ret
        return dict(ret=ret, A=A, B=B, C=Cm, D=Dm, U_colmeans=Ucm, I_colmeans=Icm, pre=pre, w_main_multiplier=wmm[0])

    def fit_collective_explicit_als(self, A, B, row, col, val, k, biasA=None, biasB=None, Cm=None,
                                    Dm=None, U=None, II=None, user_bias=True, item_bias=True,
                                    center=True, lam=10.0, scale_lam=False, scale_lam_sideinfo=False,
                                    k_main=0, k_user=0, k_item=0, w_user=1.0, w_item=1.0, niter=10,
                                    nthreads=1, use_cg=True, max_cg_steps=3, precondition_cg=False,
                                    finalize_chol=True, reset_values=False, seed=1, precompute=False, m=None, n=None,
                                    U_coo=None, I_coo=None, nonneg=False, nonneg_C=False, nonneg_D=False, max_cd_steps=100,
                                    l1_lam=0.0, add_implicit_features=False, w_implicit=1.0, w_main=1.0, lam_unique=None,
                                    l1_lam_unique=None, scale_bias_const=False, weight=None, NA_as_zero_X=False, Xfull=None,
                                    center_U=True, center_I=True, NA_as_zero_U=False, NA_as_zero_I=False):
        """U_coo / I_coo = (row, col, val, rows, cols): sparse side information instead of dense U / II (NA_as_zero_U / _I:
        its absent entries are zeros instead of missing).
        center_U / center_I = False: no column means are handed over, the reference then uses U / II as given.
        weight: observation weights, one per entry of X.  Xfull: dense X [m, n] with NaN for the missing entries instead
        of the triplet (row / col / val are then ignored; weight, if given, is [m, n] too)."""
        m = A.shape[0] if m is None else m; n = B.shape[0] if n is None else n
        weight = None if weight is None else np.ascontiguousarray(weight, self.dtype)
        if Xfull is not None:
            Xfull = np.array(Xfull, self.dtype, order="C", copy=True)        # the reference centres it in place
            row = col = np.empty(0, np.int32); val = np.empty(0, self.dtype)
        lam6 = None if lam_unique is None else np.ascontiguousarray(lam_unique, self.dtype)
        l16 = None if l1_lam_unique is None else np.ascontiguousarray(l1_lam_unique, self.dtype)
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, self.dtype)
        Ai = np.zeros((A.shape[0], k + k_main), self.dtype) if add_implicit_features else None
        Bi = np.zeros((B.shape[0], k + k_main), self.dtype) if add_implicit_features else None
        biasA = np.zeros(A.shape[0], self.dtype) if biasA is None else biasA
        biasB = np.zeros(B.shape[0], self.dtype) if biasB is None else biasB
        glob_mean = np.zeros(1, self.dtype)
        m_u, p = (0, 0) if U is None else U.shape
        n_i, q = (0, 0) if II is None else II.shape
        su, si = _coo_args(U_coo, self.dtype), _coo_args(I_coo, self.dtype)
        if U_coo is not None: m_u, p = U_coo[3], U_coo[4]
        if I_coo is not None: n_i, q = I_coo[3], I_coo[4]
        Ucm = np.zeros(max(p, 1), self.dtype); Icm = np.zeros(max(q, 1), self.dtype)
        if p and Cm is None:
            Cm = np.zeros((p, k_user + k), self.dtype)
        if q and Dm is None:
            Dm = np.zeros((q, k_item + k), self.dtype)
        sbA = np.zeros(1, self.dtype); sbB = np.zeros(1, self.dtype)
        kp = k + k_main + int(user_bias); kc = k_user + k; kq = k_user + kp
        pre = None
        if precompute:
            pre = dict(B_plus_bias=np.zeros((n, k_item + k + k_main + 1), self.dtype), BtB=np.zeros((kp, kp), self.dtype),
                       TransBtBinvBt=np.zeros((n, kp), self.dtype), BtXbias=np.zeros(kp, self.dtype),
                       BeTBeChol=np.zeros((kq, kq), self.dtype), BiTBi=np.zeros((k + k_main, k + k_main), self.dtype),
                       TransCtCinvCt=np.zeros((max(Cm.shape[0] if Cm is not None else 1, 1), max(kc, 1)), self.dtype),
                       CtCw=np.zeros((max(kc, 1), max(kc, 1)), self.dtype), CtUbias=np.zeros(max(kc, 1), self.dtype))
        pp = (lambda key: _ptr(pre[key])) if pre else (lambda key: None)
        ret = self.lib.fit_collective_explicit_als(
            _ptr(biasA), _ptr(biasB), _ptr(A), _ptr(B), _ptr(Cm), _ptr(Dm), _ptr(Ai), _ptr(Bi),
            C.c_bool(add_implicit_features), C.c_bool(reset_values), C.c_int(seed),
            _ptr(glob_mean), _ptr(Ucm) if center_U else None, _ptr(Icm) if center_I else None,
            C.c_int(m), C.c_int(n), C.c_int(k), _ptr(row), _ptr(col), _ptr(val), C.c_size_t(len(val)),
            _ptr(Xfull), _ptr(weight), C.c_bool(user_bias), C.c_bool(item_bias), C.c_bool(center),
            self._r(lam), _ptr(lam6), self._r(l1_lam), _ptr(l16),
            C.c_bool(scale_lam), C.c_bool(scale_lam_sideinfo), C.c_bool(scale_bias_const), _ptr(sbA), _ptr(sbB),
            _ptr(U), C.c_int(m_u), C.c_int(p), _ptr(II), C.c_int(n_i), C.c_int(q),
            *su[:4], *si[:4],
            C.c_bool(NA_as_zero_X), C.c_bool(NA_as_zero_U), C.c_bool(NA_as_zero_I),
            C.c_int(k_main), C.c_int(k_user), C.c_int(k_item),
            self._r(w_main), self._r(w_user), self._r(w_item), self._r(w_implicit),
            C.c_int(niter), C.c_int(nthreads), C.c_bool(False), C.c_bool(False),
            C.c_bool(use_cg), C.c_int(max_cg_steps), C.c_bool(precondition_cg), C.c_bool(finalize_chol),
            C.c_bool(nonneg), C.c_int(max_cd_steps), C.c_bool(nonneg_C), C.c_bool(nonneg_D),
            C.c_bool(precompute), C.c_bool(True), pp("B_plus_bias"), pp("BtB"), pp("TransBtBinvBt"), pp("BtXbias"),
            pp("BeTBeChol"), pp("BiTBi"), pp("TransCtCinvCt"), pp("CtCw"), pp("CtUbias"))
        return dict(ret=ret, A=A, B=B, C=Cm, D=Dm, biasA=biasA, biasB=biasB, glob_mean=glob_mean[0],
                    U_colmeans=Ucm, I_colmeans=Icm, pre=pre, Ai=Ai, Bi=Bi, scaling_biasA=sbA[0], scaling_biasB=sbB[0])
