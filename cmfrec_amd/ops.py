"""Operator-level entry points (level 2 of include/cmfrec_hip.h): one factor update with host
buffers.  Argument names and meaning follow the reference's internal operators
(/root/reference/src/cmfrec.h:986-1027, :1646-1683)."""
import ctypes as C

import numpy as np

from . import _lib


def _prep(A, B):
    if A.dtype != B.dtype or A.dtype.type not in (np.float64, np.float32):
        raise TypeError("A and B must share dtype float64 or float32")
    if not (A.flags.c_contiguous and B.flags.c_contiguous):
        raise ValueError("A and B must be C-contiguous (row-major)")
    return _lib.load(A.dtype), _lib.real(A.dtype)


def _csr(csr, dtype):
    p, i, v = csr
    return (np.ascontiguousarray(p, np.uint64), np.ascontiguousarray(i, np.int32), np.ascontiguousarray(v, dtype))


def optimizeA_implicit(A, B, csr, lam, k=None, use_cg=True, precondition_cg=False, max_cg_steps=3,
                       return_BtB=False):
    """In-place iALS half-step of A[m,lda] given B[n,ldb] (reference optimizeA_implicit)."""
    lib, R = _prep(A, B)
    m, lda = A.shape
    n, ldb = B.shape
    k = min(lda, ldb) if k is None else k
    p, i, v = _csr(csr, A.dtype)
    BtB = np.zeros((k, k), A.dtype) if return_BtB else None
    rc = lib.cmfrec_hip_optimizeA_implicit(_lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), C.c_int(m),
                                           C.c_int(n), C.c_int(k), _lib.ptr(p), _lib.ptr(i), _lib.ptr(v), R(lam),
                                           C.c_bool(use_cg), C.c_bool(precondition_cg), C.c_int(max_cg_steps),
                                           _lib.ptr(BtB))
    _lib.check(rc, lib, "optimizeA_implicit")
    return BtB


def optimizeA_explicit(A, B, csr, lam, lam_last=None, k=None, bias_sub=None, scale_lam=False,
                       scale_bias_const=False, use_cg=True, precondition_cg=False, max_cg_steps=3, weight=None, wsum=None):
    """In-place explicit half-step on sparse X (reference optimizeA, Case 4).  weight: observation weights in the entry
    order of ``csr``; wsum: the per-row lambda multipliers of scale_lam (default: every row's own sum of weights)."""
    lib, R = _prep(A, B)
    m, lda = A.shape
    n, ldb = B.shape
    k = min(lda, ldb) if k is None else k
    lam_last = lam if lam_last is None else lam_last
    p, i, v = _csr(csr, A.dtype)
    bs = None if bias_sub is None else np.ascontiguousarray(bias_sub, A.dtype)
    if weight is not None:
        w = np.ascontiguousarray(weight, A.dtype)
        ws = None if wsum is None else np.ascontiguousarray(wsum, A.dtype)
        assert len(w) == len(v) and (ws is None or len(ws) == m)
        rc = lib.cmfrec_hip_optimizeA_explicit_weighted(_lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), C.c_int(m),
                                                        C.c_int(n), C.c_int(k), _lib.ptr(p), _lib.ptr(i), _lib.ptr(v), _lib.ptr(w),
                                                        _lib.ptr(ws), _lib.ptr(bs), R(lam), R(lam_last), C.c_bool(scale_lam),
                                                        C.c_bool(scale_bias_const), C.c_bool(use_cg), C.c_bool(precondition_cg),
                                                        C.c_int(max_cg_steps))
        _lib.check(rc, lib, "optimizeA_explicit_weighted")
        return
    rc = lib.cmfrec_hip_optimizeA_explicit(_lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), C.c_int(m),
                                           C.c_int(n), C.c_int(k), _lib.ptr(p), _lib.ptr(i), _lib.ptr(v), _lib.ptr(bs),
                                           R(lam), R(lam_last), C.c_bool(scale_lam), C.c_bool(scale_bias_const),
                                           C.c_bool(use_cg), C.c_bool(precondition_cg), C.c_int(max_cg_steps))
    _lib.check(rc, lib, "optimizeA_explicit")


def optimizeA_dense_full(A, B, Xfull, lam, lam_last=None, k=None, do_B=False, scale_lam=False):
    """Dense full update A = X B (BtB + lam)^-1 (reference optimizeA, Case 1): the C / D step."""
    lib, R = _prep(A, B)
    m, lda = A.shape
    n, ldb = B.shape
    k = min(lda, ldb) if k is None else k
    lam_last = lam if lam_last is None else lam_last
    Xf = np.ascontiguousarray(Xfull, A.dtype)
    rc = lib.cmfrec_hip_optimizeA_dense_full(_lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), C.c_int(m),
                                             C.c_int(n), C.c_int(k), _lib.ptr(Xf), C.c_size_t(Xf.shape[1]),
                                             C.c_bool(do_B), R(lam), R(lam_last), C.c_bool(scale_lam))
    _lib.check(rc, lib, "optimizeA_dense_full")


def optimizeA_collective(A, B, Cm, csr, U, lam, w_user=1.0, lam_last=None, k=None, k_main=0, k_user=0,
                         k_item=0, bias_sub=None, scale_lam=False, scale_lam_sideinfo=False, m_u=None):
    """Collective half-step with dense U, Cholesky (reference optimizeA_collective, general branch)."""
    lib, R = _prep(A, B)
    m, lda = A.shape
    n, ldb = B.shape
    pdim = Cm.shape[0]
    lam_last = lam if lam_last is None else lam_last
    m_u = U.shape[0] if m_u is None else m_u
    p, i, v = _csr(csr, A.dtype)
    Cc = np.ascontiguousarray(Cm, A.dtype)
    Uc = np.ascontiguousarray(U, A.dtype)
    bs = None if bias_sub is None else np.ascontiguousarray(bias_sub, A.dtype)
    rc = lib.cmfrec_hip_optimizeA_collective(
        _lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), _lib.ptr(Cc), C.c_int(m), C.c_int(m_u),
        C.c_int(n), C.c_int(pdim), C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item), _lib.ptr(p),
        _lib.ptr(i), _lib.ptr(v), _lib.ptr(bs), _lib.ptr(Uc), R(lam), R(w_user), R(lam_last), C.c_bool(scale_lam),
        C.c_bool(scale_lam_sideinfo))
    _lib.check(rc, lib, "optimizeA_collective")


def optimizeA_collective_sparse(A, B, Cm, csr, U_csr, lam, w_user=1.0, lam_last=None, k=None, k_main=0, k_user=0, k_item=0,
                                bias_sub=None, scale_lam=False, scale_lam_sideinfo=False, implicit=False):
    """Collective half-step with SPARSE side information, Cholesky: ``U_csr`` = (indptr[m_u+1], indices, values) over the
    first m_u rows (reference optimizeA_collective / optimizeA_collective_implicit with U_csr, !NA_as_zero_U)."""
    lib, R = _prep(A, B)
    m, lda = A.shape
    n, ldb = B.shape
    lam_last = lam if lam_last is None else lam_last
    p, i, v = _csr(csr, A.dtype)
    up, ui, uv = _csr(U_csr, A.dtype)
    Cm = np.ascontiguousarray(Cm, A.dtype)
    bs = None if bias_sub is None else np.ascontiguousarray(bias_sub, A.dtype)
    rc = lib.cmfrec_hip_optimizeA_collective_sparse(
        _lib.ptr(A), C.c_size_t(lda), _lib.ptr(B), C.c_size_t(ldb), _lib.ptr(Cm), C.c_int(m), C.c_int(len(up) - 1), C.c_int(n),
        C.c_int(Cm.shape[0]), C.c_int(k), C.c_int(k_main), C.c_int(k_user), C.c_int(k_item), _lib.ptr(p), _lib.ptr(i), _lib.ptr(v),
        _lib.ptr(bs), _lib.ptr(up), _lib.ptr(ui), _lib.ptr(uv), R(lam), R(w_user), R(lam_last), C.c_bool(scale_lam),
        C.c_bool(scale_lam_sideinfo), C.c_bool(implicit))
    _lib.check(rc, lib, "optimizeA_collective_sparse")


def topN_batch(A, B, n_top=10, biasB=None, exclude=None):
    """Top-N item ids (and scores) for every row of ``A``: score = A_u . B_i (+ biasB[i]), descending, ties by lower
    id; ``exclude`` = (indptr, indices) CSR of items to skip per user (sorted here).  Batch counterpart of the
    reference's per-user ``topN`` (src/common.c:5127-5380)."""
    A = np.ascontiguousarray(A); B = np.ascontiguousarray(B, A.dtype)
    lib, R = _prep(A, B)
    nu, lda = A.shape
    n, ldb = B.shape
    ids = np.empty((nu, n_top), np.int32); sc = np.empty((nu, n_top), A.dtype)
    bs = None if biasB is None else np.ascontiguousarray(biasB, A.dtype)
    ep = ei = None
    if exclude is not None:
        ep = np.ascontiguousarray(exclude[0], np.uint64)
        ei = np.ascontiguousarray(exclude[1], np.int32)
        owner = np.repeat(np.arange(nu), np.diff(ep.astype(np.int64)))
        ei = np.ascontiguousarray(ei[np.lexsort((ei, owner))])          # each list ascending (the kernel binary-searches it)
    rc = lib.cmfrec_hip_topN_batch(_lib.ptr(A), C.c_size_t(lda), C.c_int(nu), _lib.ptr(B), C.c_size_t(ldb), C.c_int(n),
                                   C.c_int(lda), _lib.ptr(bs), _lib.ptr(ep), _lib.ptr(ei), C.c_int(n_top), _lib.ptr(ids),
                                   _lib.ptr(sc))
    _lib.check(rc, lib, "topN_batch")
    return ids, sc
