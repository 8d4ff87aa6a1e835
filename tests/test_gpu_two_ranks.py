"""GPU, world_size 2 on ONE device: two processes, each with its own HIP session (row_range / col_range shards built by
GpuEngine), driven by ShardedAls exactly as `bench.py --gpus 2` drives them -- unequal nnz-balanced item blocks (the padded
all-gather), the A-step in four parts on the communication stream (allgather_parts with rank > 0), the item entries through
the all-to-all of shard_coo_by_items into set_X_coo_device, and for the explicit model sideinfo_partial / all-reduce /
sideinfo_finish with real partial sums.  RCCL refuses two ranks on one device, so the collectives go over gloo through a
host-staged adaptor (device tensor -> host -> gloo -> device, in the stream order the engine asked for); everything else --
sessions, kernels, streams, events -- is the production path.  Results against ONE plain session on the whole problem:
bit for bit for the implicit model (a row's arithmetic does not depend on which shard it lives in), 1e-10 for the
collective model (the partial sums of C / D are added in a different order)."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _host_staged_collectives():
    """torch.distributed's collectives for device tensors over a gloo group: staged through the host, on the current stream."""
    import torch
    import torch.distributed as dist
    real_ag, real_ar, real_a2a = dist.all_gather_into_tensor, dist.all_reduce, dist.all_to_all_single

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        if not out.is_cuda:
            return real_ag(out, inp, group=group)
        h_in = inp.detach().cpu().contiguous()                  # waits for the current stream's work on `inp`
        parts = [torch.empty_like(h_in) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, h_in, group=group)
        out.copy_(torch.cat([p.view(-1) for p in parts]).view(out.shape))

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group)
        h = t.detach().cpu()
        real_ar(h, op=op, group=group)
        t.copy_(h)

    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        if not out.is_cuda:
            return real_a2a(out, inp, output_split_sizes, input_split_sizes, group=group)
        world = dist.get_world_size(group)
        h_in = inp.detach().cpu().contiguous()
        ins = list(input_split_sizes) if input_split_sizes is not None else [h_in.shape[0] // world] * world
        outs = list(output_split_sizes) if output_split_sizes is not None else [out.shape[0] // world] * world
        # gloo has no all-to-all with splits for every dtype: every rank gathers every rank's send buffer and split table
        tabs = [None] * world
        dist.all_gather_object(tabs, (h_in, ins), group=group)
        me = dist.get_rank(group)
        pieces = []
        for src in range(world):
            buf, sp = tabs[src]
            off = int(sum(sp[:me]))
            pieces.append(buf[off:off + sp[me]])
        res = torch.cat(pieces) if pieces else h_in[:0]
        assert res.shape[0] == int(sum(outs))
        out.copy_(res.to(out.dtype))

    real_batch = dist.batch_isend_irecv

    def batch_isend_irecv(ops):
        """ShardedAls' direct placement (exchange="p2p"): sends from / receives into device rows, staged through the host."""
        if not any(op.tensor.is_cuda for op in ops):
            return real_batch(ops)
        host_ops, landed = [], []
        for op in ops:
            if op.op is dist.isend:
                host_ops.append(dist.P2POp(dist.isend, op.tensor.detach().cpu().contiguous(), op.peer, op.group))
            else:
                h = torch.empty(op.tensor.shape, dtype=op.tensor.dtype)
                host_ops.append(dist.P2POp(dist.irecv, h, op.peer, op.group))
                landed.append((op.tensor, h))
        reqs = real_batch(host_ops)

        class Done:
            def wait(self):
                for r in reqs:
                    r.wait()
                for dst, h in landed:
                    dst.copy_(h)

        return [Done()]

    dist.all_gather_into_tensor, dist.all_reduce, dist.all_to_all_single = all_gather_into_tensor, all_reduce, all_to_all_single
    dist.batch_isend_irecv = batch_isend_irecv


def _worker(rank, world, port, case, path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _host_staged_collectives()
    from cmfrec_amd.distributed import GpuEngine, ShardedAls
    d = np.load(path, allow_pickle=False)
    dev = torch.device("cuda", 0)
    m, n, k = int(d["m"]), int(d["n"]), int(d["k"])
    row, col, val = d["row"], d["col"], d["val"]
    blk = m // world
    row_ranges = [(r * blk, (r + 1) * blk) for r in range(world)]
    r0, r1 = row_ranges[rank]
    mine = (row >= r0) & (row < r1)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    try:
        if case in ("implicit", "implicit-dealt"):
            eng = GpuEngine.from_device_coo(m, n, k, t(row[mine] - r0), t(col[mine]), t(val[mine]), row_ranges, rank, world, 0,
                                            dtype=val.dtype.type, lam=5.0, max_cg_steps=3, a_parts=4,
                                            item_blocks="dealt" if case == "implicit-dealt" else "contiguous")
            sizes = {e - b for b, e in eng.ranges("B")}
            if case == "implicit":
                assert len(sizes) == 2, "the nnz-balanced item blocks of this test are meant to be unequal"
            else:
                assert len(sizes) == 1 and eng.full("B").shape[0] == 2 * next(iter(sizes)) >= n, "dealt item blocks are equal"
            assert len(eng.parts("A")) == 4
            eng.full("A").copy_(t(d["A0"])); eng.full("B").zero_()
            torch.cuda.synchronize()
            als = ShardedAls(eng, rank, world)
            for _ in range(3):
                als.iteration()
            eng.session.sync(); torch.cuda.synchronize()
            f = eng.session.get_factors()
            out = {"A": f["A"], "B": eng.items_in_order(f["B"])}
            if case == "implicit-dealt":
                cnt = np.bincount(col, minlength=n)
                blk = eng.full("B").shape[0] // world
                per_rank = np.bincount(eng.item_ids // blk, weights=cnt, minlength=world)
                out["nnz_per_rank"] = per_rank
        else:
            p, q = int(d["p"]), int(d["q"])
            U, II = d["U"], d["II"]
            biases = bool(d["biases"])
            eng = GpuEngine.from_collective_block(m, n, k, t(row[mine] - r0), t(col[mine]), t(val[mine]), row_ranges, rank, world, 0,
                                                  U_local=np.ascontiguousarray(U[r0:r1]), I_local=lambda c0, c1: np.ascontiguousarray(II[c0:c1]),
                                                  p=p, q=q, m_u=m, n_i=n, dtype=np.float64, lam=0.3, w_user=0.5, w_item=2.0,
                                                  user_bias=biases, item_bias=biases, scale_lam=True)
            fac = dict(A=d["A0"], B=d["B0"], biasA=np.zeros(m) if biases else None, biasB=np.zeros(n) if biases else None,
                       Cm=np.zeros((p, k)), Dm=np.zeros((q, k)))
            eng.session.set_factors(**fac)
            als = ShardedAls(eng, rank, world)
            for _ in range(3):
                als.iteration_collective()
            eng.session.sync(); torch.cuda.synchronize()
            f = eng.session.get_factors()
            out = {key: f[key] for key in ("A", "B", "C", "D") + (("biasA", "biasB") if biases else ())}
        np.savez(path + ".rank%d.npz" % rank, **out)
    finally:
        dist.destroy_process_group()


def _run_two_ranks(case, payload):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        path = os.path.join(td, "in.npz")
        np.savez(path, **payload)
        mp.spawn(_worker, args=(2, _free_port(), case, path), nprocs=2, join=True)
        return [dict(np.load(path + ".rank%d.npz" % r)) for r in range(2)]


@pytest.mark.parametrize("exchange", ["collective", "p2p"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_two_rank_implicit_hip_sessions(dtype, exchange, monkeypatch):
    """exchange="p2p": the updated blocks (B: unequal blocks; A: in four parts on the communication stream) travel by direct
    placement -- one batch of sends / receives per half-step or part, no staging buffer -- instead of all-gathers."""
    from cmfrec_amd.session import AlsSession
    monkeypatch.setenv("CMFREC_ALLGATHER", exchange)         # read by ShardedAls in the spawned ranks
    from conftest import make_coo
    m, n, k, nnz = 16000, 5001, 50, 500000
    row, col, val = make_coo(m, n, nnz, 7, heavy_row=(3, 450))      # uniform: ~31 entries per user, ~100 per item, one long row
    # entries of user block 0 first (stable): the all-to-all of the sharded set-up delivers a column's entries ordered by source
    # rank, so this is the COO order in which both runs see them -- and with it the order of every sum
    o = np.argsort(row // (m // 2), kind="stable")
    row, col, val = row[o], col[o], val[o].astype(dtype)
    assert np.bincount(row).max() < 500 and np.bincount(col).max() < 500     # no split rows: shards and whole take the same kernels
    A0 = (np.random.default_rng(1).random((m, k)) * 2.0 ** -7).astype(dtype)
    ref = AlsSession(m, n, k, implicit=True, dtype=dtype, lam=5.0, use_cg=True, max_cg_steps=3)
    ref.set_X_coo(row, col, val)
    ref.set_factors(A=A0, B=np.zeros((n, k), dtype))
    for _ in range(3):
        ref.update("B"); ref.update("A")
    fr = ref.get_factors()
    res = _run_two_ranks("implicit", dict(m=m, n=n, k=k, row=row, col=col, val=val, A0=A0))
    for r in range(2):                         # every rank ends with the full replicas, and they are the single session's
        assert np.array_equal(res[r]["A"], fr["A"]), "A, rank %d" % r
        assert np.array_equal(res[r]["B"], fr["B"]), "B, rank %d" % r


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_two_rank_implicit_dealt_item_blocks(dtype):
    """item_blocks="dealt": the items renumbered so that both ranks own the same number of rows of B (its all-gather then lands
    in the replica directly, like A's) with balanced entry counts.  A row's system does not depend on how the opposing rows are
    numbered, but the order in which a user's entries are summed follows the new item ids where the kernels sort by them
    (split rows) and B^T B is summed in the new order, so the comparison with the single session is to rounding -- which three
    iterations of three CG steps on rows of ~3000 entries amplify: 3e-11 in double precision, 4.5e-3 in single precision for a
    mere renumbering inside ONE session (tools/microbench/item_renumbering_sensitivity.py; with a Zipf(1.3) popularity single
    precision is 9 % away from itself and 19 % from double precision -- the data, not the sharding).  Skewed item popularity
    makes the balance matter."""
    from cmfrec_amd.session import AlsSession
    m, n, k, nnz = 16000, 5001, 32, 400000
    rng = np.random.default_rng(11)
    w = 1.0 / (np.arange(n) + 20.0)
    col = rng.choice(n, size=nnz, p=w / w.sum()).astype(np.int32)
    row = rng.integers(0, m, size=len(col)).astype(np.int32)
    key = np.unique(row.astype(np.int64) * n + col)
    row, col = (key // n).astype(np.int32), (key % n).astype(np.int32)
    o = np.argsort(row // (m // 2), kind="stable")
    row, col = row[o], col[o]
    val = rng.integers(1, 6, len(row)).astype(dtype)
    assert np.bincount(col).max() > 2000                      # split rows in both precisions
    A0 = (rng.random((m, k)) * 2.0 ** -7).astype(dtype)
    ref = AlsSession(m, n, k, implicit=True, dtype=dtype, lam=5.0, use_cg=True, max_cg_steps=3)
    ref.set_X_coo(row, col, val)
    ref.set_factors(A=A0, B=np.zeros((n, k), dtype))
    for _ in range(3):
        ref.update("B"); ref.update("A")
    fr = ref.get_factors()
    res = _run_two_ranks("implicit-dealt", dict(m=m, n=n, k=k, row=row, col=col, val=val, A0=A0))
    # (B^T B is summed over the items in their new order, split rows by new ids: three iterations of three CG steps carry that)
    tol = 2e-2 if dtype is np.float32 else 1e-7
    for r in range(2):
        assert res[r]["B"].shape == fr["B"].shape
        for key in ("A", "B"):
            assert np.abs(res[r][key] - fr[key]).max() <= tol * np.abs(fr[key]).max(), (key, r)
        per_rank = res[r]["nnz_per_rank"]
        assert abs(per_rank[0] - per_rank[1]) <= 0.02 * per_rank.sum(), per_rank          # the heaviest item alone holds far more than that


@pytest.mark.parametrize("biases", [False, True])
def test_two_rank_collective_hip_sessions(biases):
    from cmfrec_amd.session import AlsSession
    from conftest import make_coo, rel_err
    m, n, k, p, q = 3000, 1801, 24, 10, 7
    row, col, val = make_coo(m, n, 90000, 9, counts=False, heavy_row=(4, 1500), empty_rows=(8,))
    o = np.argsort(row // (m // 2), kind="stable")           # see test_two_rank_implicit_hip_sessions
    row, col, val = row[o], col[o], val[o]
    val = val - val.mean()
    rng = np.random.default_rng(2)
    U = rng.standard_normal((m, p)); U -= U.mean(0)
    II = rng.standard_normal((n, q)); II -= II.mean(0)
    A0 = rng.standard_normal((m, k)) * 0.05; B0 = rng.standard_normal((n, k)) * 0.05
    kw = dict(implicit=False, dtype=np.float64, lam=0.3, use_cg=False, user_bias=biases, item_bias=biases, scale_lam=True, p=p, m_u=m,
              q=q, n_i=n, w_user=0.5, w_item=2.0)
    ref = AlsSession(m, n, k, **kw)
    ref.set_X_coo(row, col, val)
    ref.set_sideinfo(U=U, II=II)
    ref.set_factors(A=A0, B=B0, biasA=np.zeros(m) if biases else None, biasB=np.zeros(n) if biases else None, Cm=np.zeros((p, k)),
                    Dm=np.zeros((q, k)))
    ref.iterate(3)
    fr = ref.get_factors()
    res = _run_two_ranks("collective", dict(m=m, n=n, k=k, p=p, q=q, row=row, col=col, val=val, U=U, II=II, A0=A0, B0=B0,
                                            biases=np.array(biases)))
    keys = ("A", "B", "C", "D") + (("biasA", "biasB") if biases else ())
    for key in keys:
        assert np.array_equal(res[0][key], res[1][key]), "replicas differ: " + key       # identical on both ranks
        assert rel_err(res[0][key], fr[key]) < 1e-10, key
