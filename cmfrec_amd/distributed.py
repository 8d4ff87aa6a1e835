"""Row-block multi-GPU ALS: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed path; rows of a factor matrix are independent
given the opposing matrix (every `omp parallel for` on the path is over rows), so users and items
are split into contiguous blocks, each rank updates its own block of A / B on its GPU and the
updated blocks are all-gathered after every half-step so that every rank holds full replicas as
gather sources.  The only collective on the data path is that all-gather.

``ShardedAls`` is written against a small engine protocol so the partitioning / gathering logic
can be exercised on CPU with gloo (tests/test_distributed_gloo.py supplies an oracle-backed
engine); ``GpuEngine`` is the real one (AlsSession, HBM-resident).
"""
import numpy as np


def balanced_boundaries(counts, parts):
    """Contiguous split of rows into ``parts`` blocks with (nearly) equal total nnz.
    counts: nnz per row.  Returns parts+1 boundaries, non-decreasing, first 0, last len(counts)."""
    counts = np.asarray(counts, np.int64)
    n = len(counts)
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = csum[-1]
    bounds = [0]
    for p in range(1, parts):
        target = total * p / parts
        b = int(np.searchsorted(csum, target, side="left"))
        b = max(bounds[-1], min(b, n))
        bounds.append(b)
    bounds.append(n)
    return bounds


def equal_boundaries(n, parts):
    step = -(-n // parts)
    return [min(i * step, n) for i in range(parts)] + [n]


def dealt_item_order(counts, parts):
    """Item blocks that are EQUAL in rows and balanced in nnz at once: the items in descending order of their entry count
    (stable) are dealt to the ranks like cards, in snake order (0 .. parts-1, parts-1 .. 0, ...), and rank r's items get the
    new ids r * blk + 0, 1, 2, ... in the order dealt.  Contiguous nnz-balanced blocks (balanced_boundaries) are unequal in
    rows, which forces the all-gather of B through a padded staging buffer and world - 1 strided copies per half-step
    (ShardedAls.allgather); with equal blocks it lands in the replica directly, like A's.
    Returns (new_id [n] int64, blk): new_id[item] in [0, parts * blk); the ids no item maps to (parts * blk - n of them, the
    last slots of some blocks) are rows of B without entries."""
    counts = np.asarray(counts, np.int64)
    n = len(counts)
    blk = -(-n // parts)
    order = np.argsort(-counts, kind="stable")
    pos = np.arange(n)
    rnd, j = pos // parts, pos % parts
    rank = np.where(rnd % 2 == 0, j, parts - 1 - j)
    new_id = np.empty(n, np.int64)
    new_id[order] = rank * blk + rnd
    return new_id, blk


def shard_coo_by_items(row_global, col, val, n, rank, world, group=None, col_bounds=None):
    """Set-up of the item side of a row-block run without ever holding the whole matrix on one rank: every rank brings
    the entries of ITS user block (global row ids, global column ids); item blocks are cut nnz-balanced from the
    all-reduced per-item counts, and every entry travels to the rank that owns its column in one
    all-to-all (variable splits).  Entries arrive ordered by source rank and, inside a source, in their original order,
    i.e. in the order of the concatenated COO -- the order the reference's stable counting sort would see.
    Tensors may live on any device (gloo on CPU in the tests, RCCL on the GPUs).
    Returns (col_bounds [world+1], rows_of_my_items (global), cols_of_my_items (global), vals)."""
    import torch
    import torch.distributed as dist
    if col_bounds is None:
        counts = torch.bincount(col.long(), minlength=n)
        if world > 1:
            dist.all_reduce(counts, group=group)
        col_bounds = balanced_boundaries(counts.cpu().numpy(), world)
    if world == 1:
        return col_bounds, row_global, col, val
    inner = torch.as_tensor(col_bounds[1:-1], device=col.device, dtype=col.dtype)
    dest = torch.bucketize(col, inner, right=True)          # owner of the column: number of inner boundaries <= col
    order = torch.argsort(dest, stable=True)
    send = torch.bincount(dest, minlength=world)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    in_splits, out_splits = send.tolist(), recv.tolist()

    def xchg(t):
        out = t.new_empty(int(sum(out_splits)))
        dist.all_to_all_single(out, t[order].contiguous(), out_splits, in_splits, group=group)
        return out

    return col_bounds, xchg(row_global), xchg(col), xchg(val)


class ShardedAls:
    """ALS loop over row-block shards.  engine protocol:
        engine.update(which, use_cholesky=False)   -- recompute the LOCAL block of 'A' or 'B'
        engine.full(which)      -> 2-D torch tensor [rows, ld] (the local replica, updated in place)
        engine.ranges(which)    -> list of (begin, end) per rank
        engine.after_gather(which)
        engine.pre_collective() / engine.post_collective()  -- stream hand-over hooks
      optional, for overlapping the all-gather with the update (allgather_parts):
        engine.parts(which) -> [(r0, r1)] local row offsets in completion order, engine.comm_stream(),
        engine.wait_part(which, part, stream), engine.join_comm(stream)
    """

    def __init__(self, engine, rank, world, group=None, exchange=None):
        """exchange: how the updated row blocks travel after a half-step --
          "collective" (default): `all_gather_into_tensor` (RCCL picks ring / tree), through a padded staging buffer when the
                                  blocks are unequal;
          "p2p"                 : every rank sends its block to each peer and receives each peer's block STRAIGHT INTO its
                                  rows of the replica, all 2 (world - 1) transfers in one `batch_isend_irecv` group -- on an
                                  xGMI node every pair of GPUs has a link of its own, so the group is one hop on seven links at
                                  once (2.1 ms against the ring's 14.6 ms bound for config 4's A), and unequal
                                  blocks need neither padding nor copy kernels.
        Default from CMFREC_ALLGATHER.  Which one wins on an 8-GPU node is a measurement this repository could not take (one
        GPU per call); both give bit-identical replicas (tests/test_distributed_gloo.py, tests/test_gpu_two_ranks.py)."""
        import os
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self._stage = {}
        self.exchange = exchange or os.environ.get("CMFREC_ALLGATHER", "collective")
        if self.exchange not in ("collective", "p2p"):
            raise ValueError("exchange must be 'collective' or 'p2p'")

    def _p2p_exchange(self, full, ranges, r0=None, r1=None):
        """Direct placement: rows [b + r0, b + r1) of every rank's block (the whole block when r0 is None) from their owner
        into this rank's replica.  Peers are visited in the order rank + 1, rank + 2, ... so that no two ranks start on the
        same peer."""
        import torch.distributed as dist
        me, world = self.rank, self.world

        def rows(r):
            b, e = ranges[r]
            return (b, e) if r0 is None else (b + r0, min(b + r1, e))

        def peer(r):
            return dist.get_global_rank(self.group, r) if self.group is not None else r

        ops = []
        sb, se = rows(me)
        for off in range(1, world):
            dst, src = (me + off) % world, (me - off) % world
            if se > sb:
                ops.append(dist.P2POp(dist.isend, full[sb:se], peer(dst), self.group))
            rb, re = rows(src)
            if re > rb:
                ops.append(dist.P2POp(dist.irecv, full[rb:re], peer(src), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()          # RCCL: the current stream waits (no host wait); gloo: the host does

    def allgather(self, which):
        import contextlib
        import torch
        import torch.distributed as dist
        eng = self.engine
        full = eng.full(which)
        ranges = eng.ranges(which)
        b0, b1 = ranges[self.rank]
        sizes = [e - b for b, e in ranges]
        # Engines that expose their stream (GpuEngine) get the collective ENQUEUED on it: kernels -> all-gather -> next
        # half-step are ordered by the stream, the host never waits inside the loop.  Others (the CPU test engine) keep
        # the pre / post hooks.
        ordered = eng.ordered_stream() if hasattr(eng, "ordered_stream") else None
        if ordered is None:
            eng.pre_collective()
        if self.world == 1 and not dist.is_initialized():
            if ordered is None:
                eng.post_collective()
            return
        ld = full.shape[1]
        with (torch.cuda.stream(ordered) if ordered is not None else contextlib.nullcontext()):
            if self.exchange == "p2p":
                self._p2p_exchange(full, ranges)
            elif len(set(sizes)) == 1 and sizes[0] * self.world == full.shape[0]:
                # equal blocks tiling the matrix exactly: gather straight into the replica
                local = full[b0:b1].clone()
                dist.all_gather_into_tensor(full.view(-1), local.view(-1), group=self.group)
            else:
                mx = max(sizes)
                key = (which, mx, ld)
                if key not in self._stage:
                    self._stage[key] = (torch.zeros((self.world, mx, ld), dtype=full.dtype, device=full.device),
                                        torch.zeros((mx, ld), dtype=full.dtype, device=full.device))
                stage, local = self._stage[key]
                local[: b1 - b0].copy_(full[b0:b1])
                dist.all_gather_into_tensor(stage.view(-1), local.view(-1), group=self.group)
                for r, (rb, re) in enumerate(ranges):
                    if r != self.rank and re > rb:
                        full[rb:re].copy_(stage[r, : re - rb])
        if ordered is None:
            eng.post_collective()

    def allgather_parts(self, which, parts):
        """All-gather of a block that its engine completes part by part (engine.parts): the collective of part c is
        issued on the engine's communication stream as soon as that part's rows are final, beside the kernels of the
        parts after it.  Needs equal blocks on all ranks (each part is one all_gather_into_tensor into a staging
        buffer [world, rows, ld], copied into the replica on the same stream)."""
        import contextlib
        import torch
        import torch.distributed as dist
        eng = self.engine
        full = eng.full(which)
        ranges = eng.ranges(which)
        b0, _ = ranges[self.rank]
        blk = ranges[0][1] - ranges[0][0]
        ld = full.shape[1]
        view = full.view(self.world, blk, ld)
        cs = eng.comm_stream()
        for c, (r0, r1) in enumerate(parts):
            if self.exchange == "p2p":
                eng.wait_part(which, c, cs)
                with (torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()):
                    self._p2p_exchange(full, ranges, r0, r1)
                continue
            key = (which, "part", c, r1 - r0, ld)
            if key not in self._stage:
                self._stage[key] = torch.empty((self.world, r1 - r0, ld), dtype=full.dtype, device=full.device)
            stage = self._stage[key]
            eng.wait_part(which, c, cs)
            with (torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()):
                dist.all_gather_into_tensor(stage.view(-1), full[b0 + r0:b0 + r1].reshape(-1), group=self.group)
                if self.rank > 0:
                    view[:self.rank, r0:r1].copy_(stage[:self.rank])
                if self.rank + 1 < self.world:
                    view[self.rank + 1:, r0:r1].copy_(stage[self.rank + 1:])
        eng.join_comm(cs)

    def half_step(self, which, use_cholesky=False):
        import torch.distributed as dist
        self.engine.update(which, use_cholesky)
        parts = self.engine.parts(which) if hasattr(self.engine, "parts") else None
        sizes = {e - b for b, e in self.engine.ranges(which)}
        if parts and len(parts) > 1 and len(sizes) == 1 and dist.is_initialized() and \
                next(iter(sizes)) * self.world == self.engine.full(which).shape[0]:
            self.allgather_parts(which, parts)
        else:
            self.allgather(which)
        self.engine.after_gather(which)

    def iteration(self, use_cholesky=False):
        # reference order: B then A (src/collective.c:9924-10022)
        self.half_step("B", use_cholesky)
        self.half_step("A", use_cholesky)

    # ---- explicit / collective model: side information sharded with the rows ----
    def sideinfo_step(self, which):
        """C ('C') or D ('D') update of the collective model (optimizeA Case 1, src/common.c:2793-2991) with U / I cut
        into the same row blocks as A / B: every rank adds up  F_loc^T F_loc  and  U_loc^T F_loc  over ITS rows
        (engine.sideinfo_partial), one all-reduce makes the sums global, and every rank solves the same small system
        (engine.sideinfo_finish) -- C / D stay identical replicas without ever being communicated."""
        import contextlib
        import torch
        import torch.distributed as dist
        eng = self.engine
        buf = eng.sideinfo_partial(which)              # flat tensor, updated in place
        ordered = eng.ordered_stream() if hasattr(eng, "ordered_stream") else None
        if self.world > 1 or dist.is_initialized():
            with (torch.cuda.stream(ordered) if ordered is not None else contextlib.nullcontext()):
                dist.all_reduce(buf, group=self.group)
        eng.sideinfo_finish(which)

    def iteration_collective(self, use_cholesky=True):
        """One ALS iteration of the explicit model with side information, reference order C, D, B, A
        (src/collective.c:8334-8898).  Bias columns ride in the gathered rows (engine.after_gather splits them off)."""
        eng = self.engine
        if eng.has_sideinfo("C"):
            self.sideinfo_step("C")
        if eng.has_sideinfo("D"):
            self.sideinfo_step("D")
        self.half_step("B", use_cholesky)
        self.half_step("A", use_cholesky)


class _DevArray:
    """Zero-copy view of session-owned HBM for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": np.dtype(dtype).str,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class GpuEngine:
    """One rank's shard: an AlsSession that owns rows [row_begin,row_end) of A and columns
    [col_begin,col_end) of B, plus torch views of its replicas for the RCCL all-gather."""

    def __init__(self, session, row_ranges, col_ranges):
        import torch
        self.session = session
        self._ranges = {"A": row_ranges, "B": col_ranges}
        self._full = {}
        for which in ("A", "B"):
            ptr, rows, ld = session.device_ptr(which)
            self._full[which] = torch.as_tensor(_DevArray(ptr, (rows, ld), session.dtype), device="cuda")

    @classmethod
    def from_user_block(cls, m_blk, n, k, row, col, val, A0_blk, lam, max_cg_steps, rank, world, device,
                        dtype=np.float64, a_parts=1):
        """Implicit model, weak-scaling layout of bench.py: every rank brings its own user block
        (local rows, global item ids); the CSC shard of the rank's item block is assembled from all
        ranks' triplets with one all-gather at set-up."""
        import torch
        import torch.distributed as dist
        from .session import AlsSession
        m = m_blk * world
        row_ranges = [(r * m_blk, (r + 1) * m_blk) for r in range(world)]
        cb = equal_boundaries(n, world)
        col_ranges = [(cb[r], cb[r + 1]) for r in range(world)]
        c0, c1 = col_ranges[rank]
        dev = torch.device("cuda", device)
        g_row = torch.as_tensor(row.astype(np.int64) + rank * m_blk, device=dev).to(torch.int32)
        g_col = torch.as_tensor(col, device=dev)
        g_val = torch.as_tensor(val.astype(dtype), device=dev)
        nn = len(val)
        all_row = torch.empty(world * nn, dtype=torch.int32, device=dev)
        all_col = torch.empty(world * nn, dtype=torch.int32, device=dev)
        all_val = torch.empty(world * nn, dtype=g_val.dtype, device=dev)
        dist.all_gather_into_tensor(all_row, g_row)
        dist.all_gather_into_tensor(all_col, g_col)
        dist.all_gather_into_tensor(all_val, g_val)
        keep = (all_col >= c0) & (all_col < c1)
        crow = all_row[keep].cpu().numpy(); ccol = (all_col[keep] - c0).cpu().numpy(); cval = all_val[keep].cpu().numpy()
        del all_row, all_col, all_val, keep
        order = np.argsort(row, kind="stable")
        csr_p = np.zeros(m_blk + 1, np.uint64); np.cumsum(np.bincount(row, minlength=m_blk), out=csr_p[1:])
        csr = (csr_p, col[order].astype(np.int32), val[order].astype(dtype))
        order = np.argsort(ccol, kind="stable")
        csc_p = np.zeros(c1 - c0 + 1, np.uint64); np.cumsum(np.bincount(ccol, minlength=c1 - c0), out=csc_p[1:])
        csc = (csc_p, crow[order].astype(np.int32), cval[order].astype(dtype))
        sess = AlsSession(m, n, k, implicit=True, dtype=dtype, lam=lam, use_cg=True, max_cg_steps=max_cg_steps,
                          row_range=row_ranges[rank], col_range=col_ranges[rank], device=device)
        sess.set_X(csr, csc)
        if a_parts > 1:
            sess.set_A_parts(csr, a_parts)
        eng = cls(sess, row_ranges, col_ranges)
        # start values: every rank contributes its user block of A; B starts at zero (collective.c:9765-9768)
        fullA = eng.full("A")
        fullA[rank * m_blk:(rank + 1) * m_blk].copy_(torch.as_tensor(np.ascontiguousarray(A0_blk, dtype), device=dev))
        eng.full("B").zero_()
        torch.cuda.synchronize()
        ShardedAls(eng, rank, world).allgather("A")
        return eng

    @classmethod
    def from_device_coo(cls, m, n, k, row_local, col, val, row_ranges, rank, world, device, dtype=np.float32, lam=5.0,
                        max_cg_steps=3, a_parts=1, group=None, item_blocks="contiguous"):
        """Implicit model from a COO block that lives in HBM (torch CUDA tensors; BASELINE config 4 is generated shard-wise
        on the device): this rank's USER block (rows local to row_ranges[rank], global item ids).  Item blocks are cut
        nnz-balanced, the entries of a rank's items arrive through one all-to-all (shard_coo_by_items), CSR and CSC are
        built on the device.  No rank ever holds the whole matrix.
        item_blocks: "contiguous" -- blocks of consecutive item ids, balanced in nnz, unequal in rows; "dealt" -- the items are
        renumbered (dealt_item_order) so that the blocks are equal in rows AND balanced: B then has world * blk >= n rows in
        the new numbering, engine.item_ids maps an item to its row (engine.items_in_order(B) puts a replica back)."""
        import torch
        import torch.distributed as dist
        from .session import AlsSession
        r0, r1 = row_ranges[rank]
        row_local = row_local.to(torch.int32); col = col.to(torch.int32)
        item_ids, col_bounds_in = None, None
        if item_blocks == "dealt" and world > 1:
            counts = torch.bincount(col.long(), minlength=n)
            dist.all_reduce(counts, group=group)
            item_ids, blk = dealt_item_order(counts.cpu().numpy(), world)
            col = torch.as_tensor(item_ids, device=col.device)[col.long()].to(torch.int32)
            n = blk * world
            col_bounds_in = [r * blk for r in range(world + 1)]
        elif item_blocks not in ("contiguous", "dealt"):
            raise ValueError("item_blocks must be 'contiguous' or 'dealt'")
        col_bounds, crow, ccol, cval = shard_coo_by_items(row_local + r0, col, val, n, rank, world, group=group, col_bounds=col_bounds_in)
        col_ranges = [(int(col_bounds[r]), int(col_bounds[r + 1])) for r in range(world)]
        c0, c1 = col_ranges[rank]
        sess = AlsSession(m, n, k, implicit=True, dtype=dtype, lam=lam, use_cg=True, max_cg_steps=max_cg_steps,
                          row_range=row_ranges[rank], col_range=col_ranges[rank], device=device)
        sess.set_X_coo_device("r", row_local, col, val)
        sess.set_X_coo_device("c", (ccol - c0).to(torch.int32), crow.to(torch.int32), cval)
        del crow, ccol, cval
        if a_parts > 1:
            sess.set_A_parts_resident(a_parts)
        eng = cls(sess, row_ranges, col_ranges)
        eng.item_ids = item_ids
        return eng

    item_ids = None         # "dealt" item blocks: item -> row of B (numpy int64 [n_items]); None: the identity

    def items_in_order(self, B):
        """Rows of a replica of B (array [rows, ...]) in the caller's item numbering."""
        return B if self.item_ids is None else B[self.item_ids]

    def full(self, which):
        return self._full[which]

    def ranges(self, which):
        return self._ranges[which]

    @classmethod
    def from_collective_block(cls, m, n, k, row_local, col, val, row_ranges, rank, world, device, U_local=None, I_local=None,
                              p=0, q=0, m_u=0, n_i=0, dtype=np.float64, lam=1.0, w_user=1.0, w_item=1.0, user_bias=False,
                              item_bias=False, scale_lam=False, group=None):
        """Explicit model with dense side information, sharded: this rank's user block of X (device COO, rows local), its
        rows of U and -- after the item blocks are known -- its rows of I (a callable ``I_local(c0, c1)`` or the array).
        X is expected centred (and the bias start values set through the replicas) by the caller."""
        import torch
        from .session import AlsSession
        r0, r1 = row_ranges[rank]
        row_local = row_local.to(torch.int32); col = col.to(torch.int32)
        col_bounds, crow, ccol, cval = shard_coo_by_items(row_local + r0, col, val, n, rank, world, group=group)
        col_ranges = [(int(col_bounds[r]), int(col_bounds[r + 1])) for r in range(world)]
        c0, c1 = col_ranges[rank]
        sess = AlsSession(m, n, k, implicit=False, dtype=dtype, lam=lam, use_cg=False, user_bias=user_bias, item_bias=item_bias,
                          scale_lam=scale_lam, p=p, q=q, m_u=m_u, n_i=n_i, w_user=w_user, w_item=w_item,
                          row_range=row_ranges[rank], col_range=col_ranges[rank], device=device)
        sess.set_X_coo_device("r", row_local, col, val)
        sess.set_X_coo_device("c", (ccol - c0).to(torch.int32), crow.to(torch.int32), cval)
        Il = I_local(c0, c1) if callable(I_local) else I_local
        sess.set_sideinfo_local(U_local, Il)
        eng = cls(sess, row_ranges, col_ranges)
        eng._side = {"C": p > 0, "D": q > 0}
        return eng

    def has_sideinfo(self, which):
        return bool(getattr(self, "_side", {}).get(which, False))

    def sideinfo_partial(self, which):
        import torch
        ptr, elems = self.session.sideinfo_partial(which)
        return torch.as_tensor(_DevArray(ptr, (elems,), self.session.dtype), device="cuda")

    def sideinfo_finish(self, which):
        self.session.sideinfo_finish(which)

    def ordered_stream(self):
        """The session's own HIP stream as a torch stream: collectives enqueued on it are ordered after the kernels of the
        half-step before them and before those of the next one without any host synchronisation."""
        import torch
        if getattr(self, "_ext", None) is None:
            self._ext = torch.cuda.ExternalStream(int(self.session.stream()))
        return self._ext

    def update(self, which, use_cholesky=False):
        self.session.update(which, use_cholesky)

    def after_gather(self, which):
        self.session.after_gather(which)

    def parts(self, which):
        return self.session.part_ranges() if which == "A" else []

    def comm_stream(self):
        import torch
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream()
        return self._comm

    def wait_part(self, which, part, stream):
        self.session.stream_wait_part(part, stream.cuda_stream)

    def join_comm(self, stream):
        # the next half-step (enqueued on the session's stream) must see every part's gathered rows: the session's
        # stream waits for the communication stream, the host does not
        self.ordered_stream().wait_stream(stream)

    def pre_collective(self):
        # the session launches on its own stream; RCCL runs on torch's
        self.session.sync()

    def post_collective(self):
        import torch
        torch.cuda.current_stream().synchronize()
