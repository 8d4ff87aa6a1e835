"""CMF / CMF_implicit estimators: the ``fit()`` surface of the reference's Python classes
(/root/reference/cmfrec/__init__.py:2881-2896 ``CMF.__init__``, :4673-4683 ``CMF_implicit.__init__``,
:938-1181 ``_fit_common``, :3150-3248 / :4882-4928 ``_fit``), driving the HIP library through
the reference's own C signatures (include/cmfrec_hip.h level 1).

Only the ALS path is offered (``method="als"``); options outside SURVEY.md section 8 raise
``NotImplementedError`` / ``ValueError`` instead of silently doing something else.
Inputs: ``X`` as a SciPy COO matrix (any sparse format is converted), or a ``(row, col, value)``
triplet plus ``shape``; dense side information ``U`` [m_u, p] / ``I`` [n_i, q] as NumPy arrays.
"""
import ctypes as C
import multiprocessing

import numpy as np

from . import _lib


def _coo_triplet(X, shape=None):
    if isinstance(X, tuple):
        row, col, val = X
        if shape is None:
            shape = (int(np.max(row)) + 1, int(np.max(col)) + 1)
    else:
        import scipy.sparse as sp
        if not sp.issparse(X):
            raise TypeError("'X' must be a SciPy sparse matrix or a (row, col, value) triplet")
        X = X.tocoo()
        row, col, val, shape = X.row, X.col, X.data, X.shape
    m, n = int(shape[0]), int(shape[1])
    if max(m, n) > np.iinfo(np.int32).max:      # reference guard, __init__.py:1146-1149
        raise ValueError("Error: dimensions of 'X' are too large for 32-bit indices.")
    return (np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32), val, m, n)


class _Base:
    def _setup(self, use_float, nthreads, n_jobs):
        self.use_float = bool(use_float)
        self.dtype_ = np.float32 if self.use_float else np.float64
        if n_jobs is not None:
            nthreads = n_jobs
        if nthreads is None or nthreads < 1:
            nthreads = multiprocessing.cpu_count()
        self.nthreads = int(nthreads)
        self.is_fitted_ = False

    def _lib(self):
        return _lib.load(self.dtype_), _lib.real(self.dtype_)


def _topN(self, users, n, exclude, biasB):
    """Shared body of ``CMF.topN_batch`` / ``CMF_implicit.topN_batch``."""
    from . import ops
    users = np.atleast_1d(np.asarray(users, np.int64))
    A = np.ascontiguousarray(self.A_[users][:, self.k_user:])
    B = np.ascontiguousarray(self.B_[:, self.k_item:])
    excl = None
    if exclude is not None:                       # scipy CSR / (indptr, indices) over ALL users: take the asked rows
        ip, ix = (exclude.indptr, exclude.indices) if hasattr(exclude, "indptr") else exclude
        ip = np.asarray(ip, np.int64); ix = np.asarray(ix)
        lens = ip[users + 1] - ip[users]
        ep = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        take = np.concatenate([np.arange(ip[u], ip[u + 1]) for u in users]) if lens.sum() else np.zeros(0, np.int64)
        excl = (ep, ix[take].astype(np.int32))
    return ops.topN_batch(A, B, n_top=n, biasB=biasB, exclude=excl)


def _new_rows(X, n, dt):
    """COO triplet of new rows: ``X`` is a SciPy sparse matrix [m_x, n], a (row, col, val) triplet or None."""
    if X is None:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, dt), 0
    if isinstance(X, tuple):
        row, col, val = X
        row = np.asarray(row); col = np.asarray(col)
        if len(row) and (row.min() < 0 or col.min() < 0):
            raise ValueError("'X' has negative indices")
        if len(col) and col.max() >= n:
            raise ValueError("'X' has more columns than the model has items")
        m_x = int(np.max(row)) + 1 if len(row) else 0
    else:
        X = X.tocoo()
        if X.shape[1] > n:
            raise ValueError("'X' has more columns than the model has items")
        row, col, val, m_x = X.row, X.col, X.data, X.shape[0]
    return (np.ascontiguousarray(row, np.int32), np.ascontiguousarray(col, np.int32),
            np.ascontiguousarray(val, dt), int(m_x))


def _side_info(M, dt):
    """Side information for fit(): dense array -> (array, None); SciPy sparse matrix -> (None, (row, col, val, rows, cols))."""
    if M is None:
        return None, None
    if hasattr(M, "tocoo"):
        M = M.tocoo()
        return None, (np.ascontiguousarray(M.row, np.int32), np.ascontiguousarray(M.col, np.int32),
                      np.ascontiguousarray(M.data, dt), int(M.shape[0]), int(M.shape[1]))
    return np.ascontiguousarray(M, dt), None


def _penalty(value, name):
    """``lambda_`` / ``l1_lambda``: a number, or six numbers (user bias, item bias, A, B, C, D) like the reference
    (cmfrec/__init__.py:99-101).  Returns (scalar, array-or-None); with an array the scalar handed to C is 0, as the
    reference does (cmfrec/__init__.py:3179-3180)."""
    if np.isscalar(value):
        return float(value), None
    arr = np.asarray(value, np.float64).reshape(-1)
    if arr.shape[0] != 6:
        raise ValueError("'%s' must be a single number or an array with 6 entries." % name)
    return 0.0, arr


class CMF_implicit(_Base):
    """Implicit-feedback model (iALS / WRMF), reference class ``CMF_implicit``."""

    def __init__(self, k=50, lambda_=1e0, alpha=1., use_cg=True, k_user=0, k_item=0, k_main=0,
                 w_main=1., w_user=10., w_item=10., l1_lambda=0., center_U=True, center_I=True,
                 niter=10, NA_as_zero_user=False, NA_as_zero_item=False, nonneg=False, nonneg_C=False,
                 nonneg_D=False, max_cd_steps=100, apply_log_transf=False,
                 precompute_for_predictions=True, use_float=True, max_cg_steps=3,
                 precondition_cg=False, finalize_chol=False, random_state=1, verbose=False,
                 produce_dicts=False, handle_interrupt=True, nthreads=-1, n_jobs=None):
        # sparse side information whose absent entries are zeros (not missing): cmfrec/__init__.py:4688-4690.  The C library
        # runs it as the zero-filled dense matrix it denotes (fit.hip, ZeroFilledSide)
        self.NA_as_zero_user = bool(NA_as_zero_user); self.NA_as_zero_item = bool(NA_as_zero_item)
        self.k = int(k); self.alpha = float(alpha); self.use_cg = bool(use_cg)
        self.lambda_, self._lam6 = _penalty(lambda_, "lambda_")
        self.k_user = int(k_user); self.k_item = int(k_item); self.k_main = int(k_main)
        self.w_main = float(w_main); self.w_user = float(w_user); self.w_item = float(w_item)
        self.l1_lambda = l1_lambda; self.niter = int(niter); self.nonneg = bool(nonneg)
        self.center_U = bool(center_U); self.center_I = bool(center_I)
        self.apply_log_transf = bool(apply_log_transf)
        self.precompute_for_predictions = bool(precompute_for_predictions)
        self.max_cg_steps = int(max_cg_steps); self.precondition_cg = bool(precondition_cg)
        self.finalize_chol = bool(finalize_chol); self.random_state = int(random_state)
        self.verbose = bool(verbose); self.handle_interrupt = bool(handle_interrupt)
        self._setup(use_float, nthreads, n_jobs)
        self.nonneg_C = bool(nonneg_C); self.nonneg_D = bool(nonneg_D); self.max_cd_steps = int(max_cd_steps)
        self.l1_lambda, self._l16 = _penalty(l1_lambda, "l1_lambda")

    def fit(self, X, U=None, I=None, shape=None, A0=None, B0=None):
        """Fits the model.  ``A0``/``B0`` (optional) inject the start values instead of drawing
        them from ``random_state`` (C argument ``reset_values=false``).  ``U`` / ``I``: dense side
        information without missing values, or SciPy sparse matrices (missing = absent)."""
        row, col, val, m, n = _coo_triplet(X, shape)
        lib, R = self._lib()
        dt = self.dtype_
        lam6 = None if self._lam6 is None else np.ascontiguousarray(self._lam6, dt)
        l16 = None if self._l16 is None else np.ascontiguousarray(self._l16, dt)
        val = np.ascontiguousarray(val, dt)
        Uc, Us = _side_info(U, dt)
        Ic, Is = _side_info(I, dt)
        m_u, p = (0, 0) if Uc is None else Uc.shape
        n_i, q = (0, 0) if Ic is None else Ic.shape
        if Us is not None: m_u, p = Us[3], Us[4]
        if Is is not None: n_i, q = Is[3], Is[4]
        spU = (None, None, None, C.c_size_t(0)) if Us is None else (_lib.ptr(Us[0]), _lib.ptr(Us[1]), _lib.ptr(Us[2]), C.c_size_t(len(Us[2])))
        spI = (None, None, None, C.c_size_t(0)) if Is is None else (_lib.ptr(Is[0]), _lib.ptr(Is[1]), _lib.ptr(Is[2]), C.c_size_t(len(Is[2])))
        ka, kb = self.k_user + self.k + self.k_main, self.k_item + self.k + self.k_main
        reset = A0 is None
        A = np.empty((max(m, m_u), ka), dt) if reset else np.array(A0, dt, order="C", copy=True)     # m_max rows
        B = np.empty((max(n, n_i), kb), dt) if (reset or B0 is None) else np.array(B0, dt, order="C", copy=True)
        if not reset and B0 is None:
            B[:] = 0
        Cm = np.zeros((p, self.k_user + self.k), dt) if p else None
        Dm = np.zeros((q, self.k_item + self.k), dt) if q else None
        Ucm = np.zeros(max(p, 1), dt); Icm = np.zeros(max(q, 1), dt)
        wmm = np.zeros(1, dt)
        pre = self.precompute_for_predictions
        kq = self.k_user + self.k + self.k_main
        BtB = np.zeros((self.k + self.k_main,) * 2, dt) if pre else None
        BeTBe = np.zeros((kq, kq), dt) if (pre and p) else None
        BeTBeChol = np.zeros((kq, kq), dt) if (pre and p) else None
        CtUbias = np.zeros(self.k_user + self.k, dt) if (pre and p and Us is not None and self.NA_as_zero_user) else None
        rc = lib.fit_collective_implicit_als(
            _lib.ptr(A), _lib.ptr(B), _lib.ptr(Cm), _lib.ptr(Dm), C.c_bool(reset), C.c_int(self.random_state),
            _lib.ptr(Ucm) if (p and self.center_U) else None, _lib.ptr(Icm) if (q and self.center_I) else None,
            C.c_int(m), C.c_int(n), C.c_int(self.k), _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
            C.c_size_t(len(val)), R(self.lambda_), _lib.ptr(lam6), R(self.l1_lambda), _lib.ptr(l16),
            _lib.ptr(Uc), C.c_int(m_u), C.c_int(p), _lib.ptr(Ic), C.c_int(n_i), C.c_int(q),
            *spU, *spI,
            C.c_bool(self.NA_as_zero_user), C.c_bool(self.NA_as_zero_item), C.c_int(self.k_main), C.c_int(self.k_user), C.c_int(self.k_item),
            R(self.w_main), R(self.w_user), R(self.w_item), _lib.ptr(wmm),
            R(self.alpha), C.c_bool(bool(getattr(self, "_adjust_weight", False))),   # the reference's estimator always passes False (__init__.py:4753)
            C.c_bool(self.apply_log_transf), C.c_int(self.niter), C.c_int(self.nthreads),
            C.c_bool(self.verbose), C.c_bool(self.handle_interrupt), C.c_bool(self.use_cg),
            C.c_int(self.max_cg_steps), C.c_bool(self.precondition_cg), C.c_bool(self.finalize_chol),
            C.c_bool(self.nonneg), C.c_int(self.max_cd_steps), C.c_bool(self.nonneg_C), C.c_bool(self.nonneg_D),
            C.c_bool(pre), _lib.ptr(BtB), _lib.ptr(BeTBe), _lib.ptr(BeTBeChol), _lib.ptr(CtUbias))
        # an interrupted fit returns 3 with usable factors; like the reference's wrapper, raise only when the caller did
        # not ask for the interrupt to be handled (cmfrec/wrapper_untyped.pxi: `if ret_code == 3 and not handle_interrupt`)
        _lib.check(rc, lib, "fit_collective_implicit_als", interrupt_ok=self.handle_interrupt)
        self.A_, self.B_ = A, B
        self.C_ = Cm if Cm is not None else np.empty((0, 0), dt)
        self.D_ = Dm if Dm is not None else np.empty((0, 0), dt)
        self._U_colmeans, self._I_colmeans = Ucm[:p], Icm[:q]
        self._w_main_multiplier = float(wmm[0])
        # precomputed matrices for predictions on new data, reference attribute names (cmfrec/__init__.py:4893-4927)
        e = np.empty((0, 0), dt)
        self._BtB = BtB if BtB is not None else e
        self._BeTBe = BeTBe if BeTBe is not None else e
        self._BeTBeChol = BeTBeChol if BeTBeChol is not None else e
        self._CtUbias = CtUbias if CtUbias is not None else np.empty(0, dt)
        self.is_fitted_ = True
        return self

    def factors_multiple(self, X=None, U=None):
        """Factors of new users from their interactions ``X`` [m_x, n] (sparse) and / or dense attributes ``U`` [m_u, p]
        (reference ``CMF_implicit.factors_multiple``, cmfrec/__init__.py:5313; C function
        factors_collective_implicit_multiple).  Returns ``A`` [max(m_x, m_u), k_user+k+k_main]."""
        if X is None and U is None:
            raise ValueError("Must pass at least one of 'X', 'U'.")
        lam6 = None if self._lam6 is None else np.ascontiguousarray(self._lam6, self.dtype_)
        l1 = self.l1_lambda if self._l16 is None else float(self._l16[2])       # the reference passes l1_lambda[2], like lambda_
        lib, R = self._lib()
        dt = self.dtype_
        n = self.B_.shape[0]
        row, col, val, m_x = _new_rows(X, n, dt)
        Uc = None if (U is None or not self.C_.shape[0]) else np.ascontiguousarray(U, dt)
        m_u, p = (0, 0) if Uc is None else Uc.shape
        A = np.empty((max(m_x, m_u), self.k_user + self.k + self.k_main), dt)
        has = lambda M: M is not None and M.shape[0] > 0
        rc = lib.factors_collective_implicit_multiple(
            _lib.ptr(A), C.c_int(m_x), _lib.ptr(Uc), C.c_int(m_u), C.c_int(p), C.c_bool(False), C.c_bool(self.nonneg),
            None, None, None, C.c_size_t(0), None, None, None,
            _lib.ptr(val), _lib.ptr(row), _lib.ptr(col), C.c_size_t(len(val)), None, None, None,
            _lib.ptr(self.B_), C.c_int(n), _lib.ptr(self.C_) if p else None,
            _lib.ptr(self._U_colmeans) if (p and len(self._U_colmeans)) else None,
            C.c_int(self.k), C.c_int(self.k_user), C.c_int(self.k_item), C.c_int(self.k_main),
            R(self.lambda_ if lam6 is None else float(lam6[2])), R(l1), R(self.alpha), R(self.w_main), R(self.w_user),
            R(self._w_main_multiplier),
            C.c_bool(self.apply_log_transf),
            _lib.ptr(self._BeTBe) if has(self._BeTBe) else None, _lib.ptr(self._BtB) if has(self._BtB) else None,
            _lib.ptr(self._BeTBeChol) if has(self._BeTBeChol) else None, None, C.c_int(self.nthreads))
        _lib.check(rc, lib, "factors_collective_implicit_multiple")
        return A

    def topN_batch(self, users, n=10, exclude=None):
        """Top-``n`` item ids and scores (A_u . B_i) for a batch of users, ranked on the GPU; ``exclude``: CSR of items
        to skip per user (e.g. the training matrix).  Batch form of the reference's ``topN`` (common.c:5127-5380)."""
        return _topN(self, users, n, exclude, None)

    def predict(self, user, item):
        """A_u . B_i for paired user / item ids (reference predict_multiple, common.c:5066-5106)."""
        user = np.asarray(user); item = np.asarray(item)
        return np.einsum("ij,ij->i", self.A_[user, self.k_user:], self.B_[item, self.k_item:])


class CMF(_Base):
    """Explicit-feedback collective model, reference class ``CMF`` (ALS only)."""

    def __init__(self, k=40, lambda_=1e+1, method="als", use_cg=True, user_bias=True, item_bias=True,
                 center=True, add_implicit_features=False, scale_lam=False, scale_lam_sideinfo=False,
                 scale_bias_const=False, k_user=0, k_item=0, k_main=0, w_main=1., w_user=1., w_item=1.,
                 w_implicit=0.5, l1_lambda=0., center_U=True, center_I=True, maxiter=800, niter=10,
                 parallelize="separate", corr_pairs=4, max_cg_steps=3, precondition_cg=False,
                 finalize_chol=True, NA_as_zero=False, NA_as_zero_user=False, NA_as_zero_item=False,
                 nonneg=False, nonneg_C=False, nonneg_D=False, max_cd_steps=100,
                 precompute_for_predictions=True, include_all_X=True, use_float=True, random_state=1,
                 verbose=False, print_every=10, handle_interrupt=True, produce_dicts=False, nthreads=-1,
                 n_jobs=None):
        if method != "als":
            raise NotImplementedError("only method='als' is implemented in cmfrec_amd")
        # sparse side information whose absent entries are zeros (not missing): cmfrec/__init__.py:2903-2905.  The C library
        # runs it as the zero-filled dense matrix it denotes (fit.hip, ZeroFilledSide)
        self.NA_as_zero_user = bool(NA_as_zero_user); self.NA_as_zero_item = bool(NA_as_zero_item)
        # NA_as_zero (absent entries of a sparse X are zeros): without side information or with dense complete / sparse side
        # information on exactly the rows / columns of X, closed form or CG, with add_implicit_features too; the matrices for
        # predictions on new data (with BtXbias) for the model without side information and implicit features only.  With
        # observation weights (fit(..., W=)): the same models without implicit features; start values given by the caller
        # (A0 / B0 / biasA0 / biasB0) when the model has biases -- the reference's own bias start values are not defined for that
        # combination (common.c:4727-4731).  What is refused and why: DESIGN.md section 7.
        self.NA_as_zero = bool(NA_as_zero)
        # (precompute_for_predictions with NA_as_zero: the model without side information -- checked in fit(), where the data is known)
        self.scale_bias_const = bool(scale_bias_const)
        if add_implicit_features and (nonneg or not np.isscalar(l1_lambda) or l1_lambda):
            raise NotImplementedError("add_implicit_features together with nonneg / l1_lambda is not implemented in "
                                      "cmfrec_amd (the reference itself crashes on it)")
        self.add_implicit_features = bool(add_implicit_features)
        self.l1_lambda, self._l16 = _penalty(l1_lambda, "l1_lambda")
        self.nonneg = bool(nonneg); self.nonneg_C = bool(nonneg_C); self.nonneg_D = bool(nonneg_D)
        self.max_cd_steps = int(max_cd_steps)
        self.k = int(k); self.use_cg = bool(use_cg)
        self.lambda_, self._lam6 = _penalty(lambda_, "lambda_")
        self.user_bias = bool(user_bias); self.item_bias = bool(item_bias); self.center = bool(center)
        self.center_U = bool(center_U); self.center_I = bool(center_I)
        self.scale_lam = bool(scale_lam); self.scale_lam_sideinfo = bool(scale_lam_sideinfo)
        self.k_user = int(k_user); self.k_item = int(k_item); self.k_main = int(k_main)
        self.w_main = float(w_main); self.w_user = float(w_user); self.w_item = float(w_item)
        self.w_implicit = float(w_implicit); self.niter = int(niter); self.max_cg_steps = int(max_cg_steps)
        self.precondition_cg = bool(precondition_cg); self.finalize_chol = bool(finalize_chol)
        self.precompute_for_predictions = bool(precompute_for_predictions)
        self.random_state = int(random_state); self.verbose = bool(verbose)
        self.handle_interrupt = bool(handle_interrupt)
        self._setup(use_float, nthreads, n_jobs)

    def fit(self, X, U=None, I=None, shape=None, A0=None, B0=None, biasA0=None, biasB0=None, W=None):
        """``X``: SciPy sparse matrix / COO triplet, or a dense 2-D array with NaN for the missing entries (the plain model:
        no side information).  ``W``: observation weights, one per entry of ``X`` in its COO order (reference
        ``CMF.fit(X, ..., W=...)``, cmfrec/__init__.py:3066; an array(nnz,) for sparse ``X``, an array(m, n) for dense ``X``)."""
        lib, R = self._lib()
        dt = self.dtype_
        Xfull = None
        if isinstance(X, np.ndarray):
            if X.ndim != 2:
                raise ValueError("a dense 'X' must be a 2-D array")
            if self.NA_as_zero:
                raise ValueError("'NA_as_zero' is for a sparse 'X'.")
            Xfull = np.ascontiguousarray(X, dt)
            m, n = Xfull.shape
            row = col = np.empty(0, np.int32); val = np.empty(0, dt)
        else:
            row, col, val, m, n = _coo_triplet(X, shape)
        Wv = None
        if W is not None:
            Wv = np.ascontiguousarray(np.asarray(W).reshape(-1), dt)
            if Wv.shape[0] != (len(val) if Xfull is None else Xfull.size):
                raise ValueError("'W' must have the same number of entries as 'X'.")
        lam6 = None if self._lam6 is None else np.ascontiguousarray(self._lam6, dt)
        l16 = None if self._l16 is None else np.ascontiguousarray(self._l16, dt)
        val = np.ascontiguousarray(val, dt)
        Uc, Us = _side_info(U, dt)
        Ic, Is = _side_info(I, dt)
        m_u, p = (0, 0) if Uc is None else Uc.shape
        n_i, q = (0, 0) if Ic is None else Ic.shape
        if Us is not None: m_u, p = Us[3], Us[4]
        if Is is not None: n_i, q = Is[3], Is[4]
        spU = (None, None, None, C.c_size_t(0)) if Us is None else (_lib.ptr(Us[0]), _lib.ptr(Us[1]), _lib.ptr(Us[2]), C.c_size_t(len(Us[2])))
        spI = (None, None, None, C.c_size_t(0)) if Is is None else (_lib.ptr(Is[0]), _lib.ptr(Is[1]), _lib.ptr(Is[2]), C.c_size_t(len(Is[2])))
        use_cg = self.use_cg
        ka, kb = self.k_user + self.k + self.k_main, self.k_item + self.k + self.k_main
        reset = A0 is None
        A = np.empty((max(m, m_u), ka), dt) if reset else np.array(A0, dt, order="C", copy=True)
        B = np.empty((max(n, n_i), kb), dt) if (reset or B0 is None) else np.array(B0, dt, order="C", copy=True)
        if not reset and B0 is None:
            B[:] = 0
        biasA = np.zeros(max(m, m_u), dt) if biasA0 is None else np.array(biasA0, dt, copy=True)
        biasB = np.zeros(max(n, n_i), dt) if biasB0 is None else np.array(biasB0, dt, copy=True)
        Cm = np.zeros((p, self.k_user + self.k), dt) if p else None
        Dm = np.zeros((q, self.k_item + self.k), dt) if q else None
        glob_mean = np.zeros(1, dt); Ucm = np.zeros(max(p, 1), dt); Icm = np.zeros(max(q, 1), dt)
        sbA = np.zeros(1, dt); sbB = np.zeros(1, dt)
        # with implicit features the matrices for new-row predictions are not produced (factors_multiple is unavailable)
        pre = self.precompute_for_predictions and not self.add_implicit_features
        imp = self.add_implicit_features
        Ai = np.zeros((max(m, m_u), self.k + self.k_main), dt) if imp else None
        Bi = np.zeros((max(n, n_i), self.k + self.k_main), dt) if imp else None
        kp = self.k + self.k_main + int(self.user_bias); kc = self.k_user + self.k; kq = self.k_user + kp
        Bpb = np.zeros((max(n, n_i), kb + 1), dt) if (pre and self.user_bias) else None
        BtB = np.zeros((kp, kp), dt) if pre else None
        TBt = np.zeros((max(n, n_i), kp), dt) if pre else None
        BeChol = np.zeros((kq, kq), dt) if (pre and p) else None
        TCt = np.zeros((p, kc), dt) if (pre and p) else None
        CtCw = np.zeros((kc, kc), dt) if (pre and p) else None
        CtUbias = np.zeros(kc, dt) if (pre and p and Us is not None and self.NA_as_zero_user) else None
        if self.NA_as_zero and pre and (p or q or imp):
            raise NotImplementedError("NA_as_zero with side information or implicit features: precompute_for_predictions is not "
                                      "implemented in cmfrec_amd (pass False)")
        BtXbias = np.zeros(kp, dt) if (pre and self.NA_as_zero) else None       # reference attribute BtXbias_ (cmfrec/__init__.py)
        rc = lib.fit_collective_explicit_als(
            _lib.ptr(biasA), _lib.ptr(biasB), _lib.ptr(A), _lib.ptr(B), _lib.ptr(Cm), _lib.ptr(Dm), _lib.ptr(Ai), _lib.ptr(Bi),
            C.c_bool(imp), C.c_bool(reset), C.c_int(self.random_state), _lib.ptr(glob_mean),
            # no column means = the side information is used as given (cmfrec/__init__.py passes an empty array for center_U=False)
            _lib.ptr(Ucm) if (p and self.center_U) else None, _lib.ptr(Icm) if (q and self.center_I) else None,
            C.c_int(m), C.c_int(n), C.c_int(self.k), _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
            C.c_size_t(len(val)), _lib.ptr(Xfull), _lib.ptr(Wv), C.c_bool(self.user_bias), C.c_bool(self.item_bias),
            C.c_bool(self.center), R(self.lambda_), _lib.ptr(lam6), R(self.l1_lambda), _lib.ptr(l16), C.c_bool(self.scale_lam),
            C.c_bool(self.scale_lam_sideinfo), C.c_bool(self.scale_bias_const), _lib.ptr(sbA), _lib.ptr(sbB),
            _lib.ptr(Uc), C.c_int(m_u), C.c_int(p), _lib.ptr(Ic), C.c_int(n_i), C.c_int(q),
            *spU, *spI,
            C.c_bool(self.NA_as_zero), C.c_bool(self.NA_as_zero_user), C.c_bool(self.NA_as_zero_item),
            C.c_int(self.k_main), C.c_int(self.k_user), C.c_int(self.k_item),
            R(self.w_main), R(self.w_user), R(self.w_item), R(self.w_implicit),
            C.c_int(self.niter), C.c_int(self.nthreads), C.c_bool(self.verbose), C.c_bool(self.handle_interrupt),
            C.c_bool(use_cg), C.c_int(self.max_cg_steps), C.c_bool(self.precondition_cg),
            C.c_bool(self.finalize_chol), C.c_bool(self.nonneg), C.c_int(self.max_cd_steps), C.c_bool(self.nonneg_C),
            C.c_bool(self.nonneg_D),
            C.c_bool(pre), C.c_bool(True), _lib.ptr(Bpb), _lib.ptr(BtB), _lib.ptr(TBt), _lib.ptr(BtXbias), _lib.ptr(BeChol), None,
            _lib.ptr(TCt), _lib.ptr(CtCw), _lib.ptr(CtUbias))
        _lib.check(rc, lib, "fit_collective_explicit_als", interrupt_ok=self.handle_interrupt)
        # precomputed matrices for predictions on new data, reference attribute names (cmfrec/__init__.py:3211-3247)
        e = np.empty((0, 0), dt)
        self._B_plus_bias = Bpb if Bpb is not None else e
        self._BtB = BtB if BtB is not None else e
        self._TransBtBinvBt = TBt if TBt is not None else e
        self._BtXbias = BtXbias if BtXbias is not None else np.empty(0, dt)
        self._BeTBeChol = BeChol if BeChol is not None else e
        self._TransCtCinvCt = TCt if TCt is not None else e
        self._CtUbias = CtUbias if CtUbias is not None else np.empty(0, dt)
        self._CtCw = CtCw if CtCw is not None else e
        self._scaling_biasA, self._scaling_biasB = float(sbA[0]), float(sbB[0])    # cmfrec/__init__.py: scaling_biasA_ / _B_
        self.A_, self.B_ = A, B
        self.Ai_ = Ai if imp else e                      # cmfrec/__init__.py:3195-3196
        self.Bi_ = Bi if imp else e
        self.C_ = Cm if Cm is not None else np.empty((0, 0), dt)
        self.D_ = Dm if Dm is not None else np.empty((0, 0), dt)
        self.user_bias_ = biasA if self.user_bias else np.empty(0, dt)
        self.item_bias_ = biasB if self.item_bias else np.empty(0, dt)
        self.glob_mean_ = float(glob_mean[0])
        self._U_colmeans = Ucm[:p] if self.center_U else Ucm[:0]
        self._I_colmeans = Icm[:q] if self.center_I else Icm[:0]
        self.is_fitted_ = True
        return self

    def factors_multiple(self, X=None, U=None, return_bias=False):
        """Factors (and bias) of new users from their ratings ``X`` [m_x, n] (sparse) and / or dense attributes ``U``
        [m_u, p] (reference ``CMF.factors_multiple``, cmfrec/__init__.py:3706; C function
        factors_collective_explicit_multiple).  Returns ``A`` [max(m_x, m_u), k_user+k+k_main], or ``(A, bias)``."""
        if X is None and U is None:
            raise ValueError("Must pass at least one of 'X', 'U'.")
        lam6 = None if self._lam6 is None else np.ascontiguousarray(self._lam6, self.dtype_)
        l16 = None if self._l16 is None else np.ascontiguousarray(self._l16, self.dtype_)
        if self.add_implicit_features:
            raise NotImplementedError("factors_multiple with add_implicit_features is not implemented in cmfrec_amd")
        lib, R = self._lib()
        dt = self.dtype_
        n = self.B_.shape[0]
        row, col, val, m_x = _new_rows(X, n, dt)
        Uc = None if (U is None or not self.C_.shape[0]) else np.ascontiguousarray(U, dt)
        m_u, p = (0, 0) if Uc is None else Uc.shape
        mm = max(m_x, m_u)
        A = np.empty((mm, self.k_user + self.k + self.k_main), dt)
        biasA = np.empty(mm, dt) if self.user_bias else None
        has = lambda M: M is not None and M.shape[0] > 0
        rc = lib.factors_collective_explicit_multiple(
            _lib.ptr(A), _lib.ptr(biasA), C.c_int(m_x), _lib.ptr(Uc), C.c_int(m_u), C.c_int(p),
            C.c_bool(False), C.c_bool(False), C.c_bool(self.nonneg),
            None, None, None, C.c_size_t(0), None, None, None, None, C.c_int(0), C.c_int(0),
            _lib.ptr(self.C_) if p else None, None, R(self.glob_mean_),
            _lib.ptr(self.item_bias_) if self.item_bias else None,
            _lib.ptr(self._U_colmeans) if (p and len(self._U_colmeans)) else None,
            _lib.ptr(val), _lib.ptr(row), _lib.ptr(col), C.c_size_t(len(val)), None, None, None,
            None, C.c_int(n), None, _lib.ptr(self.B_), None, C.c_bool(False),
            C.c_int(self.k), C.c_int(self.k_user), C.c_int(self.k_item), C.c_int(self.k_main),
            R(self.lambda_), _lib.ptr(lam6), R(self.l1_lambda), _lib.ptr(l16), C.c_bool(self.scale_lam), C.c_bool(self.scale_lam_sideinfo),
            C.c_bool(self.scale_bias_const), R(self._scaling_biasA if self.scale_bias_const else 1.), R(self.w_main), R(self.w_user),
            R(self.w_implicit),
            C.c_int(n), C.c_bool(True),
            None, None, None, None, None,
            _lib.ptr(self._TransCtCinvCt) if (p and has(self._TransCtCinvCt)) else None, None, None, None,
            C.c_int(self.nthreads))
        _lib.check(rc, lib, "factors_collective_explicit_multiple")
        if return_bias:
            return A, biasA
        return A

    def topN_batch(self, users, n=10, exclude=None):
        """Top-``n`` item ids and scores for a batch of users, ranked on the GPU by A_u . B_i + item_bias[i] (the user
        bias and the global mean do not change the order; the reference adds them to the scores, common.c:5339-5345,
        and so does this method).  ``exclude``: CSR of items to skip per user."""
        ids, sc = _topN(self, users, n, exclude, self.item_bias_ if self.item_bias else None)
        users = np.atleast_1d(np.asarray(users, np.int64))
        sc = sc + self.glob_mean_
        if self.user_bias:
            sc = sc + np.asarray(self.user_bias_)[users][:, None]
        return ids, sc

    def predict(self, user, item):
        """glob_mean + biasA[u] + biasB[i] + A_u . B_i (reference predict_multiple, common.c:5098-5106)."""
        user = np.asarray(user); item = np.asarray(item)
        ku, ki = self.k_user, self.k_item
        out = np.einsum("ij,ij->i", self.A_[user, ku:], self.B_[item, ki:]) + self.glob_mean_
        if self.user_bias:
            out = out + self.user_bias_[user]
        if self.item_bias:
            out = out + self.item_bias_[item]
        return out
