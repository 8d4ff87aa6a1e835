"""CPU, world_size 2, gloo: the row-block sharding + all-gather orchestration of
cmfrec_amd.distributed.ShardedAls.  The per-shard half-step is supplied by an oracle-backed engine
(test infrastructure) so that the N>1 control path -- block ranges, even / uneven all-gather,
replica consistency -- is exercised without a GPU and must reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """Engine protocol of ShardedAls on CPU: full replicas as torch tensors, the local block
    recomputed with the oracle's optimizeA_implicit on the CSR/CSC shard."""

    def __init__(self, O, A, B, csr, csc, row_ranges, col_ranges, rank, lam):
        self.O, self.rank, self.lam = O, rank, lam
        self.A, self.B = A, B                       # numpy, shared memory with the torch views
        self.tA, self.tB = torch.from_numpy(A), torch.from_numpy(B)
        self._ranges = {"A": row_ranges, "B": col_ranges}
        r0, r1 = row_ranges[rank]; c0, c1 = col_ranges[rank]
        self.csr = self._slice(csr, r0, r1)
        self.csc = self._slice(csc, c0, c1)

    @staticmethod
    def _slice(csr, b, e):
        p, i, v = csr
        lo, hi = int(p[b]), int(p[e])
        return ((p[b:e + 1] - p[b]).astype(np.uint64), i[lo:hi].copy(), v[lo:hi].copy())

    def full(self, which):
        return self.tA if which == "A" else self.tB

    def ranges(self, which):
        return self._ranges[which]

    def update(self, which, use_cholesky=False):
        b, e = self._ranges[which][self.rank]
        if which == "A":
            blk = np.ascontiguousarray(self.A[b:e])
            self.O.optimizeA_implicit(blk, self.B, self.csr, self.lam, use_cg=not use_cholesky)
            self.A[b:e] = blk
        else:
            blk = np.ascontiguousarray(self.B[b:e])
            self.O.optimizeA_implicit(blk, self.A, self.csc, self.lam, use_cg=not use_cholesky)
            self.B[b:e] = blk

    def after_gather(self, which):
        pass

    # optional part-wise all-gather (ShardedAls.allgather_parts): on the CPU the parts are complete when update() returns
    nparts = 0

    def parts(self, which):
        if which != "A" or self.nparts <= 1:
            return []
        b, e = self._ranges["A"][self.rank]
        step = -(-(e - b) // self.nparts)
        return [(min(c * step, e - b), min((c + 1) * step, e - b)) for c in range(self.nparts)]

    def comm_stream(self):
        return None

    def wait_part(self, which, part, stream):
        pass

    joined = 0

    def join_comm(self, stream):
        self.joined += 1

    def pre_collective(self):
        pass

    def post_collective(self):
        pass


def _worker(rank, world, port, balanced, out_dir, m=301, nparts=0):
    from conftest import make_coo
    from oracle.bindings import Oracle
    from cmfrec_amd.distributed import ShardedAls, balanced_boundaries, equal_boundaries
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = Oracle(np.float64)
    n, k = 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    csr, csc = O.coo_to_csr_and_csc(row, col, val, m, n)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    if balanced:
        rb = balanced_boundaries(np.diff(csr[0].astype(np.int64)), world)
        cb = balanced_boundaries(np.diff(csc[0].astype(np.int64)), world)
    else:
        rb, cb = equal_boundaries(m, world), equal_boundaries(n, world)
    rr = [(rb[i], rb[i + 1]) for i in range(world)]
    cr = [(cb[i], cb[i + 1]) for i in range(world)]
    eng = OracleEngine(O, A, B, csr, csc, rr, cr, rank, 4.0)
    eng.nparts = nparts
    als = ShardedAls(eng, rank, world)
    if nparts > 1:
        assert len(eng.parts("A")) == nparts
    for _ in range(3):
        als.iteration()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=A, B=B, joined=eng.joined, exchange=als.exchange)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("exchange", ["collective", "p2p"])
@pytest.mark.parametrize("balanced", [False, True])
def test_two_rank_als_matches_single_process(tmp_path, balanced, exchange, oracles, monkeypatch):
    from conftest import make_coo
    world = 2
    monkeypatch.setenv("CMFREC_ALLGATHER", exchange)      # (inherited by the spawned ranks; "p2p": direct placement, no staging)
    mp.spawn(_worker, args=(world, _free_port(), balanced, str(tmp_path)), nprocs=world, join=True)
    O = oracles[np.float64]
    m, n, k = 301, 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    A = np.random.default_rng(3).standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    O.fit_implicit_als(A, B, row, col, val, lam=4.0, niter=3)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    # replicas agree bit-for-bit across ranks and equal the single-process fit exactly
    assert np.array_equal(r0["A"], r1["A"]) and np.array_equal(r0["B"], r1["B"])
    assert np.array_equal(r0["A"], A) and np.array_equal(r0["B"], B)
    assert str(r0["exchange"]) == exchange and str(r1["exchange"]) == exchange


def test_three_rank_p2p_unequal_blocks(tmp_path, oracles, monkeypatch):
    """Direct placement with three ranks and nnz-balanced (unequal) blocks: every rank sends to / receives from two peers in
    one group, no padding; replicas bit-identical to the single-process fit."""
    from conftest import make_coo
    world = 3
    monkeypatch.setenv("CMFREC_ALLGATHER", "p2p")
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path)), nprocs=world, join=True)
    O = oracles[np.float64]
    m, n, k = 301, 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    A = np.random.default_rng(3).standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    O.fit_implicit_als(A, B, row, col, val, lam=4.0, niter=3)
    for r in range(world):
        z = np.load(tmp_path / ("rank%d.npz" % r))
        assert str(z["exchange"]) == "p2p"
        assert np.array_equal(z["A"], A) and np.array_equal(z["B"], B)


def test_boundaries():
    from cmfrec_amd.distributed import balanced_boundaries, equal_boundaries
    assert equal_boundaries(10, 4) == [0, 3, 6, 9, 10]
    assert equal_boundaries(160112, 8)[-2:] == [140098, 160112]
    cnt = np.array([100, 1, 1, 1, 1, 1, 1, 94])
    b = balanced_boundaries(cnt, 2)
    assert b[0] == 0 and b[-1] == 8 and b == sorted(b)
    s0, s1 = cnt[b[0]:b[1]].sum(), cnt[b[1]:b[2]].sum()
    assert abs(int(s0) - int(s1)) <= 100
    assert balanced_boundaries(np.zeros(5, int), 3)[-1] == 5


@pytest.mark.parametrize("exchange", ["collective", "p2p"])
def test_two_rank_partwise_allgather(tmp_path, oracles, exchange, monkeypatch):
    """The overlap path of the multi-GPU bench: equal user blocks, the A block gathered in three parts
    (ShardedAls.allgather_parts); must still be the single-process fit, bit for bit."""
    from conftest import make_coo
    world, m = 2, 300
    monkeypatch.setenv("CMFREC_ALLGATHER", exchange)
    mp.spawn(_worker, args=(world, _free_port(), False, str(tmp_path), m, 3), nprocs=world, join=True)
    O = oracles[np.float64]
    n, k = 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    A = np.random.default_rng(3).standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    O.fit_implicit_als(A, B, row, col, val, lam=4.0, niter=3)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["A"], r1["A"]) and np.array_equal(r0["B"], r1["B"])
    assert np.array_equal(r0["A"], A) and np.array_equal(r0["B"], B)
    assert int(r0["joined"]) == 3 and int(r1["joined"]) == 3          # the part-wise path was really taken, once per iteration


def _worker_exchange(rank, world, port, out_dir, m=300):
    """Set-up path of bench.py --gpus N: every rank holds ONLY the entries of its user block; nnz-balanced item blocks
    come from the all-reduced counts and the CSC shard from the all-to-all (shard_coo_by_items)."""
    from conftest import make_coo
    from oracle.bindings import Oracle
    from cmfrec_amd.distributed import ShardedAls, shard_coo_by_items, balanced_boundaries
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = Oracle(np.float64)
    n, k = 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    m_blk = m // world
    r0, r1 = rank * m_blk, (rank + 1) * m_blk
    mine = (row >= r0) & (row < r1)                    # this rank's block, in the global COO order
    lrow, lcol, lval = row[mine], col[mine], val[mine]
    cb, crow, ccol, cval = shard_coo_by_items(torch.from_numpy(lrow.astype(np.int64)), torch.from_numpy(lcol.astype(np.int64)),
                                              torch.from_numpy(lval), n, rank, world)
    # the boundaries are the nnz-balanced ones of the whole matrix
    assert cb == balanced_boundaries(np.bincount(col, minlength=n), world)
    c0, c1 = cb[rank], cb[rank + 1]
    crow, ccol, cval = crow.numpy(), ccol.numpy(), cval.numpy()
    assert ((ccol >= c0) & (ccol < c1)).all()
    # exactly the entries of my item block, ordered by source rank and, inside a source, in COO order
    want = (col >= c0) & (col < c1)
    src = row[want] // m_blk
    order = np.argsort(src, kind="stable")
    assert np.array_equal(crow, row[want][order]) and np.array_equal(ccol, col[want][order]) and np.array_equal(cval, val[want][order])
    # local CSR / CSC shards from the local triplets only
    csr_l, _ = O.coo_to_csr_and_csc((lrow - r0).astype(np.int32), lcol.astype(np.int32), lval, m_blk, n)
    _, csc_l = O.coo_to_csr_and_csc(crow.astype(np.int32), (ccol - c0).astype(np.int32), cval, m, c1 - c0)
    rng = np.random.default_rng(3)
    A = rng.standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    rr = [(i * m_blk, (i + 1) * m_blk) for i in range(world)]
    cr = [(cb[i], cb[i + 1]) for i in range(world)]

    class LocalShardEngine(OracleEngine):
        def __init__(self):
            self.O, self.rank, self.lam = O, rank, 4.0
            self.A, self.B = A, B
            self.tA, self.tB = torch.from_numpy(A), torch.from_numpy(B)
            self._ranges = {"A": rr, "B": cr}
            self.csr, self.csc = csr_l, csc_l

    als = ShardedAls(LocalShardEngine(), rank, world)
    for _ in range(3):
        als.iteration()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=A, B=B)
    dist.destroy_process_group()


def test_two_rank_alltoall_setup(tmp_path, oracles):
    """No rank ever sees the whole matrix, item blocks are nnz-balanced: the three iterations still equal the
    single-process fit.  (Sums inside an item's column follow the exchanged order = the global COO order here,
    because user blocks are contiguous and arrive in rank order.)"""
    from conftest import make_coo
    world, m = 2, 300
    mp.spawn(_worker_exchange, args=(world, _free_port(), str(tmp_path), m), nprocs=world, join=True)
    O = oracles[np.float64]
    n, k = 200, 8
    row, col, val = make_coo(m, n, 5000, 77, heavy_row=(2, 150))
    A = np.random.default_rng(3).standard_normal((m, k)) * 0.01
    B = np.zeros((n, k))
    O.fit_implicit_als(A, B, row, col, val, lam=4.0, niter=3)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["A"], r1["A"]) and np.array_equal(r0["B"], r1["B"])
    # the entries of a column arrive grouped by source block instead of interleaved as in the global COO: the same sums
    # in another order
    assert np.abs(r0["A"] - A).max() < 1e-12 and np.abs(r0["B"] - B).max() < 1e-12


# ---- explicit / collective model: side information sharded with the rows --------------------------------------------
def _collective_problem():
    from conftest import make_coo
    m, n, k, p, q = 120, 90, 6, 5, 4
    row, col, val = make_coo(m, n, 2400, 21, counts=False, heavy_row=(3, 70), empty_rows=(9,))
    rng = np.random.default_rng(8)
    U = rng.standard_normal((m, p)); II = rng.standard_normal((n, q))
    U -= U.mean(0); II -= II.mean(0)           # the fit centres the side information by columns (common.c:4911-4997): done up front here
    A0 = rng.standard_normal((m, k)) * 0.1; B0 = rng.standard_normal((n, k)) * 0.1
    return m, n, k, p, q, row, col, val, U, II, A0, B0


class CollectiveOracleEngine(OracleEngine):
    """CPU engine of ShardedAls.iteration_collective: the local rows of A / B through the oracle's collective Cholesky
    operator (optimizeA_collective), the C / D update as partial sums over the local rows + the small solve."""

    def __init__(self, O, A, B, Cm, Dm, csr_l, csc_l, U_l, I_l, rr, cr, rank, lam, w_user, w_item, k):
        self.O, self.rank, self.lam, self.w_user, self.w_item, self.k = O, rank, lam, w_user, w_item, k
        self.A, self.B, self.Cm, self.Dm = A, B, Cm, Dm
        self.tA, self.tB = torch.from_numpy(A), torch.from_numpy(B)
        self._ranges = {"A": rr, "B": cr}
        self.csr, self.csc, self.U_l, self.I_l = csr_l, csc_l, U_l, I_l
        self._part = {}

    def has_sideinfo(self, which):
        return True

    def update(self, which, use_cholesky=True):
        b, e = self._ranges[which][self.rank]
        if which == "A":
            blk = np.ascontiguousarray(self.A[b:e])
            self.O.optimizeA_collective_chol(blk, self.B, self.Cm, self.csr, self.U_l, self.lam, w_user=self.w_user, k=self.k)
            self.A[b:e] = blk
        else:
            blk = np.ascontiguousarray(self.B[b:e])
            self.O.optimizeA_collective_chol(blk, self.A, self.Dm, self.csc, self.I_l, self.lam, w_user=self.w_item, k=self.k)
            self.B[b:e] = blk

    def sideinfo_partial(self, which):
        b, e = self._ranges["A" if which == "C" else "B"][self.rank]
        F = (self.A if which == "C" else self.B)[b:e]
        Ul = self.U_l if which == "C" else self.I_l
        buf = torch.from_numpy(np.concatenate([(F.T @ F).ravel(), (Ul.T @ F).ravel()]))
        self._part[which] = buf
        return buf

    def sideinfo_finish(self, which):
        kc = self.k
        buf = self._part[which].numpy()
        G = buf[:kc * kc].reshape(kc, kc); R = buf[kc * kc:].reshape(-1, kc)
        w = self.w_user if which == "C" else self.w_item
        M = (self.Cm if which == "C" else self.Dm)
        M[:] = np.linalg.solve(G + (self.lam / w) * np.eye(kc), R.T).T            # collective.c:8367, common.c:2824-2875


def _worker_collective(rank, world, port, out_dir):
    from oracle.bindings import Oracle
    from cmfrec_amd.distributed import ShardedAls, shard_coo_by_items
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = Oracle(np.float64)
    m, n, k, p, q, row, col, val, U, II, A0, B0 = _collective_problem()
    m_blk = m // world
    r0, r1 = rank * m_blk, (rank + 1) * m_blk
    mine = (row >= r0) & (row < r1)
    lrow, lcol, lval = row[mine], col[mine], val[mine]
    cb, crow, ccol, cval = shard_coo_by_items(torch.from_numpy(lrow.astype(np.int64)), torch.from_numpy(lcol.astype(np.int64)),
                                              torch.from_numpy(lval), n, rank, world)
    c0, c1 = cb[rank], cb[rank + 1]
    csr_l, _ = O.coo_to_csr_and_csc((lrow - r0).astype(np.int32), lcol.astype(np.int32), lval, m_blk, n)
    _, csc_l = O.coo_to_csr_and_csc(crow.numpy().astype(np.int32), (ccol.numpy() - c0).astype(np.int32), cval.numpy(), m, c1 - c0)
    A, B = A0.copy(), B0.copy()
    Cm, Dm = np.zeros((p, k)), np.zeros((q, k))
    rr = [(i * m_blk, (i + 1) * m_blk) for i in range(world)]
    cr = [(cb[i], cb[i + 1]) for i in range(world)]
    eng = CollectiveOracleEngine(O, A, B, Cm, Dm, csr_l, csc_l, np.ascontiguousarray(U[r0:r1]), np.ascontiguousarray(II[c0:c1]),
                                 rr, cr, rank, 0.7, 0.5, 2.0, k)
    als = ShardedAls(eng, rank, world)
    for _ in range(3):
        als.iteration_collective()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), A=A, B=B, C=Cm, D=Dm)
    dist.destroy_process_group()


def test_two_rank_collective_model(tmp_path, oracles):
    """Explicit model with dense side information on both sides, U / I sharded with the rows (a config-3-shaped toy):
    partial sums + all-reduce for C / D, all-gathers for A / B.  The two ranks end with identical replicas of everything;
    against the single-process fit only the summation order of the C / D partials differs."""
    world = 2
    mp.spawn(_worker_collective, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    O = oracles[np.float64]
    m, n, k, p, q, row, col, val, U, II, A0, B0 = _collective_problem()
    A, B = A0.copy(), B0.copy()
    Cm, Dm = np.zeros((p, k)), np.zeros((q, k))
    O.fit_explicit_als(A, B, row, col, val, k, Cm=Cm, Dm=Dm, U=U, II=II, user_bias=False, item_bias=False, center=False,
                       lam=0.7, w_user=0.5, w_item=2.0, niter=3, use_cg=False, finalize_chol=False)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    for key in ("A", "B", "C", "D"):
        assert np.array_equal(r0[key], r1[key]), key
    for key, ref in (("A", A), ("B", B), ("C", Cm), ("D", Dm)):
        assert np.abs(r0[key] - ref).max() / np.abs(ref).max() < 1e-10, key


def _worker_c5_setup(rank, world, port, out_dir, scale):
    """The set-up path of `bench.py --workload c5 --gpus N` (BASELINE.json configs[4]) on the CPU: this rank's shard drawn by
    bench.c5_shard_data, the item side through shard_coo_by_items (all-reduced item counts, nnz-balanced cut, one all-to-all)."""
    import bench
    from cmfrec_amd.distributed import shard_coo_by_items
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = bench.c5_shard_data(scale, rank, world, torch.device("cpu"))
    r0, r1 = d["row_ranges"][rank]
    cb, crow, ccol, cval = shard_coo_by_items(d["row"] + r0, d["col"], d["val"], d["n"], rank, world)
    c0, c1 = int(cb[rank]), int(cb[rank + 1])
    Il = d["item_rows"](c0, c1)
    np.savez(os.path.join(out_dir, "c5_rank%d.npz" % rank), cb=np.asarray(cb), crow=crow.numpy(), ccol=ccol.numpy(), cval=cval.numpy(),
             row=(d["row"] + r0).numpy(), col=d["col"].numpy(), val=d["val"].numpy(), U=d["U"].numpy(), Il=Il.numpy(),
             dims=np.array([d["m"], d["n"], d["nnz"], d["m_blk"]]))
    dist.destroy_process_group()


def test_two_rank_c5_setup(tmp_path):
    """World 2, gloo: every rank's user block keeps its entries, every entry reaches the rank that owns its item exactly once
    and in the order of the concatenated COO (source rank, then position), the item blocks are the same on both ranks and
    nnz-balanced, U rows are local, and the rows of I a rank draws are the rows of its item block."""
    import bench
    world, scale = 2, 2e-5                       # 2000 users x 1024 items, 40 k entries
    mp.spawn(_worker_c5_setup, args=(world, _free_port(), str(tmp_path), scale), nprocs=world, join=True)
    r = [np.load(tmp_path / ("c5_rank%d.npz" % i)) for i in range(world)]
    m, n, nnz, m_blk = [int(v) for v in r[0]["dims"]]
    assert np.array_equal(r[0]["cb"], r[1]["cb"]) and r[0]["cb"][0] == 0 and r[0]["cb"][-1] == n
    cb = r[0]["cb"]
    row = np.concatenate([r[i]["row"] for i in range(world)]); col = np.concatenate([r[i]["col"] for i in range(world)])
    val = np.concatenate([r[i]["val"] for i in range(world)])
    assert len(row) == nnz and row.min() >= 0 and row.max() < m
    for i in range(world):
        assert r[i]["row"].min() >= i * m_blk and r[i]["row"].max() < (i + 1) * m_blk          # the user block is local
        mine = (col >= cb[i]) & (col < cb[i + 1])
        # the entries of this rank's items, in the order of the concatenated COO
        assert np.array_equal(r[i]["crow"], row[mine]) and np.array_equal(r[i]["ccol"], col[mine]) and np.array_equal(r[i]["cval"], val[mine])
        assert r[i]["U"].shape == (m_blk, bench.C5_P)
        full_I = bench.c5_shard_data(scale, i, world, torch.device("cpu"))["item_rows"](0, n).numpy()
        assert np.array_equal(r[i]["Il"], full_I[cb[i]:cb[i + 1]])
    counts = np.bincount(col, minlength=n)
    share = counts[cb[0]:cb[1]].sum() / counts.sum()
    assert 0.35 < share < 0.65                                                                  # nnz-balanced item blocks
    assert abs(val.mean()) < 0.2 and set(np.unique(val + 2.75)) <= set(0.5 * np.arange(1, 11))
