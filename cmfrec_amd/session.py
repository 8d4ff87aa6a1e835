"""Python handle of the device-resident ALS session (level 3 of include/cmfrec_hip.h)."""
import ctypes as C

import numpy as np

from . import _lib


class AlsSession:
    """Factor matrices, CSR/CSC of X and side information resident in HBM; one ``update`` = one
    half-step of the reference's ALS loop (src/collective.c:8334-8898 / :9827-10045)."""

    def __init__(self, m, n, k, implicit, dtype=np.float64, lam=1.0, use_cg=True, max_cg_steps=3,
                 k_main=0, k_user=0, k_item=0, user_bias=False, item_bias=False, scale_lam=False,
                 scale_lam_sideinfo=False, p=0, q=0, m_u=0, n_i=0, w_user=1.0, w_item=1.0,
                 row_range=None, col_range=None, device=-1, precondition_cg=False, m_x=0, n_x=0):
        self.dtype = np.dtype(dtype).type
        self.lib = _lib.load(self.dtype)
        M = _lib.Model if self.dtype is np.float64 else _lib.ModelF
        rb, re = (0, m) if row_range is None else row_range
        cb, ce = (0, n) if col_range is None else col_range
        self._row_range, self._col_range = (int(rb), int(re)), (int(cb), int(ce))
        self.model = M(implicit=int(implicit), m=m, n=n, k=k, k_main=k_main, k_user=k_user, k_item=k_item,
                       user_bias=int(user_bias), item_bias=int(item_bias), scale_lam=int(scale_lam),
                       scale_lam_sideinfo=int(scale_lam_sideinfo), use_cg=int(use_cg), precondition_cg=int(precondition_cg),
                       max_cg_steps=max_cg_steps, p=p, q=q, m_u=m_u, n_i=n_i, lam=lam, w_user=w_user,
                       w_item=w_item, row_begin=rb, row_end=re, col_begin=cb, col_end=ce, m_x=m_x, n_x=n_x)
        self.m, self.n, self.k = m, n, k
        self.k_totA = k_user + k + k_main
        self.k_totB = k_item + k + k_main
        self.has_bias = bool(user_bias or item_bias)
        self.handle = self.lib.cmfrec_hip_session_create(C.byref(self.model), C.c_int(device))
        if not self.handle:
            msg = self.lib.cmfrec_hip_last_error()
            raise RuntimeError("cmfrec_hip_session_create failed: %s" % (msg.decode() if msg else "?"))
        self.handle = C.c_void_p(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.cmfrec_hip_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _c(self, a, dt=None):
        return None if a is None else np.ascontiguousarray(a, self.dtype if dt is None else dt)

    def set_X(self, csr, csc):
        """csr / csc = (indptr uint64 rebased to the local block, indices int32 global, values[, observation weights])."""
        if len(csr) > 3 or len(csc) > 3:
            keep = [self._c(csr[0], np.uint64), self._c(csr[1], np.int32), self._c(csr[2]), self._c(csr[3]),
                    self._c(csc[0], np.uint64), self._c(csc[1], np.int32), self._c(csc[2]), self._c(csc[3])]
            _lib.check(self.lib.cmfrec_hip_session_set_X_weighted(self.handle, *[_lib.ptr(a) for a in keep]), self.lib, "set_X")
            return
        keep = [self._c(csr[0], np.uint64), self._c(csr[1], np.int32), self._c(csr[2]),
                self._c(csc[0], np.uint64), self._c(csc[1], np.int32), self._c(csc[2])]
        _lib.check(self.lib.cmfrec_hip_session_set_X(self.handle, *[_lib.ptr(a) for a in keep]), self.lib, "set_X")

    def set_A_parts(self, csr, nparts):
        """Cuts the local rows of A into ``nparts`` contiguous parts (same ``csr`` as given to set_X): update('A') then
        completes part by part, so each part's all-gather can overlap the kernels of the next ones (distributed.py)."""
        keep = [self._c(csr[0], np.uint64), self._c(csr[1], np.int32), self._c(csr[2])]
        _lib.check(self.lib.cmfrec_hip_session_set_A_parts(self.handle, *[_lib.ptr(a) for a in keep], C.c_int(int(nparts))),
                   self.lib, "set_A_parts")

    def part_ranges(self):
        """[(begin, end)] local row offsets of the parts of A ([] when A is not split)."""
        out = []
        for c in range(self.lib.cmfrec_hip_session_nparts(self.handle)):
            b = C.c_int(0); e = C.c_int(0)
            _lib.check(self.lib.cmfrec_hip_session_part_range(self.handle, C.c_int(c), C.byref(b), C.byref(e)), self.lib,
                       "part_range")
            out.append((b.value, e.value))
        return out

    def stream_wait_part(self, part, raw_stream):
        """Makes the HIP stream ``raw_stream`` (an integer handle) wait until part ``part`` of the last update('A') is done."""
        _lib.check(self.lib.cmfrec_hip_session_stream_wait_part(self.handle, C.c_int(part), C.c_void_p(raw_stream)), self.lib,
                   "stream_wait_part")

    def set_nonneg(self, nonneg=True, nonneg_C=False, nonneg_D=False, max_cd_steps=100):
        """Non-negative factors: coordinate descent (reference solve_nonneg) instead of the Cholesky / CG solves."""
        _lib.check(self.lib.cmfrec_hip_session_set_nonneg(self.handle, C.c_int(int(nonneg)), C.c_int(int(nonneg_C)),
                                                          C.c_int(int(nonneg_D)), C.c_int(int(max_cd_steps))), self.lib, "set_nonneg")

    def set_implicit_features(self, w_implicit=0.5, Ai=None, Bi=None):
        """Implicit features of the explicit model (reference ``add_implicit_features``): ``iterate`` then updates Bi and
        Ai between D and B and the A / B updates carry the extra term.  Call after ``set_X``."""
        R = _lib.real(self.dtype)
        cv = lambda M: None if M is None else np.ascontiguousarray(M, self.dtype)
        Ai, Bi = cv(Ai), cv(Bi)
        _lib.check(self.lib.cmfrec_hip_session_set_implicit_features(self.handle, R(w_implicit), _lib.ptr(Ai), _lib.ptr(Bi)),
                   self.lib, "set_implicit_features")
        self._kk_imp = True

    def get_implicit_features(self, m, n, kk):
        """(Ai [m, kk], Bi [n, kk]) with kk = k + k_main."""
        Ai = np.empty((m, kk), self.dtype); Bi = np.empty((n, kk), self.dtype)
        _lib.check(self.lib.cmfrec_hip_session_get_implicit_features(self.handle, _lib.ptr(Ai), _lib.ptr(Bi)), self.lib,
                   "get_implicit_features")
        return Ai, Bi

    def set_lam_unique(self, lam_unique=None, l1_lam_unique=None, max_cd_steps=100):
        """Per-matrix penalties: six values each (user bias, item bias, A, B, C, D), already divided by w_main."""
        cv = lambda a: None if a is None else np.ascontiguousarray(a, self.dtype)
        l6, l16 = cv(lam_unique), cv(l1_lam_unique)
        _lib.check(self.lib.cmfrec_hip_session_set_lam_unique(self.handle, _lib.ptr(l6), _lib.ptr(l16), C.c_int(int(max_cd_steps))),
                   self.lib, "set_lam_unique")

    def set_NA_as_zero_X(self, on=True, center=False, glob_mean=0.0):
        """The main matrix missing-as-zero (explicit model; reference NA_as_zero_X, src/collective.c:8334-8898): X must have been set
        uncentred.  With observation weights the rows' lambda multipliers under scale_lam count the absent entries; the session
        rebuilds them whenever X is uploaded again or the flag changes."""
        R = _lib.real(self.dtype)
        _lib.check(self.lib.cmfrec_hip_session_set_NA_as_zero_X(self.handle, C.c_int(int(on)), C.c_int(int(center)), R(glob_mean)),
                   self.lib, "set_NA_as_zero_X")

    def init_biases(self, lam_user, lam_item):
        """Bias start values on the device (reference initialize_biases_*, src/common.c:4410-4909)."""
        R = _lib.real(self.dtype)
        _lib.check(self.lib.cmfrec_hip_session_init_biases(self.handle, R(lam_user), R(lam_item)), self.lib, "init_biases")

    def set_X_coo(self, row, col, val, alpha=1.0, subtract=0.0, weight=None):
        """COO triplet (int32 row / col ids, values); CSR and CSC are built on the device with the
        reference's entry order (stable in COO order, src/helpers.c:1375-1491).  weight: one observation weight per entry
        (explicit model), or None."""
        R = _lib.real(self.dtype)
        keep = [self._c(row, np.int32), self._c(col, np.int32), self._c(val), self._c(weight)]
        if weight is not None and len(keep[3]) != len(keep[2]):
            raise ValueError("weight must have one entry per entry of X")
        _lib.check(self.lib.cmfrec_hip_session_set_X_coo_weighted(self.handle, *[_lib.ptr(a) for a in keep],
                                                                  C.c_size_t(len(keep[2])), R(subtract), R(alpha)), self.lib, "set_X_coo")

    def set_X_coo_device(self, which, key, other, val, alpha=1.0, subtract=0.0):
        """One shard from a COO triplet resident in HBM (torch CUDA tensors, int32 / int32 / values): which = 'r' builds
        the CSR of the local rows (key = row - row_begin, other = global column), 'c' the CSC of the local columns
        (key = column - col_begin, other = global row).  Multi-GPU set-up path (distributed.py)."""
        import torch
        R = _lib.real(self.dtype)
        tdt = torch.float64 if self.dtype is np.float64 else torch.float32
        assert key.is_cuda and other.is_cuda and val.is_cuda and key.dtype == torch.int32 and other.dtype == torch.int32 and val.dtype == tdt
        key, other, val = key.contiguous(), other.contiguous(), val.contiguous()
        if val.numel():
            # the device build counts with atomics on counts[key] and sizes its sort from the row count: an index outside the
            # shard would corrupt HBM silently (the host entry points validate theirs the same way)
            lo, hi = (self._row_range if which == "r" else self._col_range)
            n_other = self.n if which == "r" else self.m
            kmin, kmax = torch.aminmax(key)
            omin, omax = torch.aminmax(other)
            if int(kmin) < 0 or int(kmax) >= hi - lo or int(omin) < 0 or int(omax) >= n_other:
                raise ValueError("set_X_coo_device: index outside the shard (key in [0, %d), other in [0, %d))" % (hi - lo, n_other))
        torch.cuda.current_stream().synchronize()      # the tensors were produced on torch's stream, the session has its own
        _lib.check(self.lib.cmfrec_hip_session_set_X_coo_device(
            self.handle, C.c_int(ord(which)), C.c_void_p(key.data_ptr()), C.c_void_p(other.data_ptr()), C.c_void_p(val.data_ptr()),
            C.c_size_t(val.numel()), R(subtract), R(alpha)), self.lib, "set_X_coo_device")

    def set_A_parts_resident(self, nparts):
        """set_A_parts on the CSR already resident in the session (shards built on the device)."""
        _lib.check(self.lib.cmfrec_hip_session_set_A_parts(self.handle, None, None, None, C.c_int(int(nparts))), self.lib,
                   "set_A_parts")

    def get_X(self, which):
        """(indptr uint64, indices int32, values, order int32) of the resident CSR ('r') / CSC ('c')."""
        rows = self.m if which == "r" else self.n
        indptr = np.empty(rows + 1, np.uint64)
        _lib.check(self.lib.cmfrec_hip_session_get_X(self.handle, C.c_int(ord(which)), _lib.ptr(indptr), None, None, None),
                   self.lib, "get_X")
        nnz = int(indptr[-1])
        ind = np.empty(nnz, np.int32); val = np.empty(nnz, self.dtype); order = np.empty(rows, np.int32)
        _lib.check(self.lib.cmfrec_hip_session_get_X(self.handle, C.c_int(ord(which)), None, _lib.ptr(ind), _lib.ptr(val),
                                                     _lib.ptr(order)), self.lib, "get_X")
        return indptr, ind, val, order

    def set_factors(self, A=None, B=None, biasA=None, biasB=None, Cm=None, Dm=None):
        keep = [self._c(x) for x in (A, B, biasA, biasB, Cm, Dm)]
        _lib.check(self.lib.cmfrec_hip_session_set_factors(self.handle, *[_lib.ptr(a) for a in keep]), self.lib,
                   "set_factors")

    def get_factors(self):
        mdl = self.model
        out = dict(A=np.empty((self.m, self.k_totA), self.dtype), B=np.empty((self.n, self.k_totB), self.dtype))
        out["biasA"] = np.empty(self.m, self.dtype) if mdl.user_bias else None
        out["biasB"] = np.empty(self.n, self.dtype) if mdl.item_bias else None
        out["C"] = np.empty((mdl.p, mdl.k_user + mdl.k), self.dtype) if mdl.p else None
        out["D"] = np.empty((mdl.q, mdl.k_item + mdl.k), self.dtype) if mdl.q else None
        _lib.check(self.lib.cmfrec_hip_session_get_factors(
            self.handle, *[_lib.ptr(out[x]) for x in ("A", "B", "biasA", "biasB", "C", "D")]), self.lib, "get_factors")
        return out

    def set_sideinfo(self, U=None, II=None):
        keep = [self._c(U), self._c(II)]
        _lib.check(self.lib.cmfrec_hip_session_set_sideinfo(self.handle, *[_lib.ptr(a) for a in keep]), self.lib,
                   "set_sideinfo")

    def set_sideinfo_local(self, U=None, II=None):
        """Row-block shards: only the block's rows of the (centred) side information -- U rows [row_begin, min(row_end, m_u)),
        I rows [col_begin, min(col_end, n_i)).  Arrays on the host, or contiguous torch tensors of the session's dtype that
        already live in HBM (their device pointers are handed over; the session copies them)."""
        keep, ptrs = [], []
        for a in (U, II):
            if a is not None and hasattr(a, "data_ptr"):           # torch tensor (device or host)
                if not a.is_contiguous() or np.dtype(str(a.dtype).replace("torch.", "")) != np.dtype(self.dtype):
                    raise ValueError("device side information must be contiguous and of the session's dtype")
                keep.append(a); ptrs.append(C.c_void_p(a.data_ptr()))
            else:
                b = self._c(a)
                keep.append(b); ptrs.append(_lib.ptr(b))
        _lib.check(self.lib.cmfrec_hip_session_set_sideinfo_local(self.handle, *ptrs), self.lib, "set_sideinfo_local")
        if any(hasattr(a, "data_ptr") for a in keep if a is not None):
            self.sync()                                            # the copies out of the caller's tensors are done

    def sideinfo_partial(self, which):
        """First half of the C / D update of a shard: partial sums over the local rows; returns (device pointer, elements) of
        the buffer [F_loc^T F_loc | U_loc^T F_loc] for the caller's all-reduce."""
        _lib.check(self.lib.cmfrec_hip_session_sideinfo_partial(self.handle, C.c_int(ord(which))), self.lib, "sideinfo_partial")
        p, elems, _ = self.device_ptr("P")
        return p, elems

    def sideinfo_finish(self, which):
        _lib.check(self.lib.cmfrec_hip_session_sideinfo_finish(self.handle, C.c_int(ord(which))), self.lib, "sideinfo_finish")

    def update(self, which, use_cholesky=False):
        _lib.check(self.lib.cmfrec_hip_session_update(self.handle, C.c_int(ord(which)), C.c_int(int(use_cholesky))),
                   self.lib, "update(%s)" % which)

    def after_gather(self, which):
        _lib.check(self.lib.cmfrec_hip_session_after_gather(self.handle, C.c_int(ord(which))), self.lib, "after_gather")

    def iterate(self, niter, finalize_chol=False):
        _lib.check(self.lib.cmfrec_hip_session_iterate(self.handle, C.c_int(niter), C.c_int(int(finalize_chol))),
                   self.lib, "iterate")

    def sync(self):
        _lib.check(self.lib.cmfrec_hip_session_sync(self.handle), self.lib, "sync")

    def device_ptr(self, which):
        rows = C.c_size_t(0); ld = C.c_size_t(0)
        p = self.lib.cmfrec_hip_session_device_ptr(self.handle, C.c_int(ord(which)), C.byref(rows), C.byref(ld))
        return p, rows.value, ld.value

    def stream(self):
        return self.lib.cmfrec_hip_session_stream(self.handle)

    def kernel_time(self, which):
        ms = C.c_double(0); cnt = C.c_long(0)
        _lib.check(self.lib.cmfrec_hip_session_kernel_time(self.handle, C.c_int(ord(which)), C.byref(ms), C.byref(cnt)),
                   self.lib, "kernel_time")
        return ms.value, cnt.value

    def bin_stats(self, which, bin_):
        """(ms, launches, rows, nnz) of the CG row kernel launches of nnz-bin ``bin_`` (0 heavy, 1 medium, 2 light)."""
        ms = C.c_double(0); cnt = C.c_long(0); rows = C.c_long(0); nnz = C.c_ulonglong(0)
        _lib.check(self.lib.cmfrec_hip_session_bin_stats(self.handle, C.c_int(ord(which)), C.c_int(bin_), C.byref(ms),
                                                         C.byref(cnt), C.byref(rows), C.byref(nnz)), self.lib, "bin_stats")
        return ms.value, cnt.value, rows.value, nnz.value

    def bin_overlaps(self, which, bin_):
        """True when the launches of that bin run beside other kernels, so that bin_stats' time is not a kernel duration."""
        return bool(self.lib.cmfrec_hip_session_bin_overlaps(self.handle, C.c_int(ord(which)), C.c_int(bin_)))

    def vh_min(self, which):
        """Rows of X's CSR ('A') / CSC ('B') shard with at least this many entries are split rows (entries sorted by opposing index)."""
        return int(self.lib.cmfrec_hip_session_vh_min(self.handle, C.c_int(ord(which))))

    def lowrank_info(self):
        """(rows, eig) of the most recent collective Cholesky half-step: rows solved by the low-rank kernels (0: path not taken),
        eigen-decomposition 3 = tridiagonalisation + QL (eig_kernels.hpp), 2 = one-workgroup Jacobi kernel."""
        rows, eig = C.c_int(0), C.c_int(0)
        self.lib.cmfrec_hip_session_lowrank_info(self.handle, C.byref(rows), C.byref(eig))
        return int(rows.value), int(eig.value)

    def vh_mode(self, which):
        """0: no split rows on that side; 1: streamed per CG pass; 2: single gather + CG on the row's Gramian."""
        return int(self.lib.cmfrec_hip_session_vh_mode(self.handle, C.c_int(ord(which))))

    def reset_timers(self):
        self.lib.cmfrec_hip_session_reset_timers(self.handle)

    def reload_switches(self):
        """Read the CMFREC_HIP_* environment switches again (they are read when a session is created)."""
        try:
            fn = self.lib.cmfrec_hip_reload_switches
        except AttributeError:      # a comparison build of an earlier round (CMFREC_HIP_LIBDIR): it reads the environment at every launch
            return
        fn()
