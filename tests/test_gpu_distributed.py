"""GPU, one rank: the multi-GPU engine of bench.py (GpuEngine + ShardedAls over RCCL) with the A block cut into parts
whose all-gathers overlap the following parts' kernels.  With one rank the collectives degenerate to copies, so this
pins the control path -- part boundaries, events between the session's stream and the communication stream, staging
copies -- against a plain single-session run.  (The 2-rank logic runs on CPU with gloo, test_distributed_gloo.py.)"""
import os
import socket

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("parts", [1, 4])
def test_one_rank_engine_matches_session(parts):
    import torch
    import torch.distributed as dist
    from cmfrec_amd.session import AlsSession
    from cmfrec_amd.distributed import GpuEngine, ShardedAls
    m, n, k, nnz = 20000, 6000, 50, 600000
    row, col, val = bench.synth_block(m, n, nnz, seed=5)
    A0 = np.random.default_rng(1).random((m, k)) * 2.0 ** -7
    ref = AlsSession(m, n, k, implicit=True, dtype=np.float64, lam=5.0, use_cg=True, max_cg_steps=3)
    ref.set_X(bench.to_csr(row, col, val, m), bench.to_csr(col, row, val, n))
    ref.set_factors(A=A0, B=np.zeros((n, k)))
    for _ in range(3):
        ref.update("B"); ref.update("A")
    f = ref.get_factors(); Ar, Br = f["A"], f["B"]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = GpuEngine.from_user_block(m, n, k, row, col, val, A0, lam=5.0, max_cg_steps=3, rank=0, world=1, device=0,
                                        a_parts=parts)
        assert len(eng.parts("A")) == (parts if parts > 1 else 0)
        als = ShardedAls(eng, 0, 1)
        for _ in range(3):
            als.iteration()
        torch.cuda.synchronize()
        f = eng.session.get_factors(); A, B = f["A"], f["B"]
    finally:
        dist.destroy_process_group()
    # the parts change which rows share a launch, not the arithmetic of a row
    assert np.array_equal(B, Br) and np.array_equal(A, Ar)


@pytest.mark.parametrize("parts", [1, 4])
def test_one_rank_device_coo_engine(parts):
    """The set-up path of `bench.py --gpus N` (BASELINE config 4): COO generated on the device, item blocks from the
    all-reduced counts, CSR / CSC built from device triplets, collectives enqueued on the session's stream (no host
    synchronisation in the loop), A-step in parts.  One rank through RCCL against a plain session on the same data."""
    import torch
    import torch.distributed as dist
    from cmfrec_amd.session import AlsSession
    from cmfrec_amd.distributed import GpuEngine, ShardedAls
    m, n, k, nnz = 24000, 7000, 64, 700000
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    row, col, val = bench.synth_block_torch(m, n, nnz, seed=41, item_seed=4, device=dev)
    hrow, hcol, hval = row.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    assert len(np.unique(hrow.astype(np.int64) * n + hcol)) == nnz and hval.min() >= 1
    A0 = (np.random.default_rng(1).random((m, k)) * 2.0 ** -7).astype(np.float32)
    ref = AlsSession(m, n, k, implicit=True, dtype=np.float32, lam=5.0, use_cg=True, max_cg_steps=3)
    ref.set_X_coo(hrow, hcol, hval)
    ref.set_factors(A=A0, B=np.zeros((n, k), np.float32))
    for _ in range(3):
        ref.update("B"); ref.update("A")
    f = ref.get_factors(); Ar, Br = f["A"], f["B"]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        eng = GpuEngine.from_device_coo(m, n, k, row, col, val, [(0, m)], 0, 1, 0, dtype=np.float32, lam=5.0, max_cg_steps=3,
                                        a_parts=parts)
        assert len(eng.parts("A")) == (parts if parts > 1 else 0)
        eng.full("A").copy_(torch.as_tensor(A0, device=dev)); eng.full("B").zero_()
        torch.cuda.synchronize()
        als = ShardedAls(eng, 0, 1)
        for _ in range(3):
            als.iteration()
        eng.session.sync(); torch.cuda.synchronize()
        f = eng.session.get_factors(); A, B = f["A"], f["B"]
    finally:
        dist.destroy_process_group()
    assert np.array_equal(B, Br) and np.array_equal(A, Ar)


@pytest.mark.parametrize("biases", [False, True])
def test_one_rank_collective_engine(biases):
    """Explicit model with dense side information on both sides as ONE row-block shard (U / I local, C / D update split into
    partial sums + all-reduce + finish, bias columns riding in the gathered rows): with one rank the partial sums are the
    whole sums, so three iterations must equal the plain session's bit for bit.  (Two ranks: tests/test_distributed_gloo.py.)"""
    import torch
    import torch.distributed as dist
    from cmfrec_amd.session import AlsSession
    from cmfrec_amd.distributed import GpuEngine, ShardedAls
    from conftest import make_coo
    m, n, k, p, q = 3000, 1800, 24, 10, 7
    row, col, val = make_coo(m, n, 90000, 9, counts=False, heavy_row=(4, 1500), empty_rows=(8,))
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
    fac = dict(A=A0, B=B0, biasA=np.zeros(m) if biases else None, biasB=np.zeros(n) if biases else None, Cm=np.zeros((p, k)),
               Dm=np.zeros((q, k)))
    ref.set_factors(**fac)
    ref.iterate(3)
    fr = ref.get_factors()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        eng = GpuEngine.from_collective_block(m, n, k, torch.as_tensor(row, device=dev), torch.as_tensor(col, device=dev),
                                              torch.as_tensor(val, device=dev), [(0, m)], 0, 1, 0, U_local=U, I_local=lambda c0, c1: II[c0:c1],
                                              p=p, q=q, m_u=m, n_i=n, dtype=np.float64, lam=0.3, w_user=0.5, w_item=2.0,
                                              user_bias=biases, item_bias=biases, scale_lam=True)
        eng.session.set_factors(**fac)
        als = ShardedAls(eng, 0, 1)
        for _ in range(3):
            als.iteration_collective()
        eng.session.sync(); torch.cuda.synchronize()
        fs = eng.session.get_factors()
    finally:
        dist.destroy_process_group()
    for key in ("A", "B", "C", "D") + (("biasA", "biasB") if biases else ()):
        assert np.array_equal(fs[key], fr[key]), key
