"""What ONE rank of `bench.py --gpus N` computes per iteration of BASELINE config 4 (10 M x 1 M, 5e8 entries, k = 64 fp32),
measured on one GPU without collectives: rank 0's shard is built as GpuEngine.from_device_coo builds it -- its own user block
as CSR over all items; its nnz-balanced item block as CSC over the users of ALL N blocks (the other ranks' blocks are drawn
here one after the other and only the entries of rank 0's items are kept) -- and update('B') / update('A') are timed, whole
block and in CMFREC_HIP_AG_PARTS parts.  Shows what the scaling run's ranks are busy with besides the all-gathers.
    python tools/microbench/c4_rank_of_n.py 8 [2 4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import bench
from cmfrec_amd.session import AlsSession
from cmfrec_amd.distributed import balanced_boundaries

dev = torch.device("cuda", 0)
K = bench.C4_K
for N in [int(a) for a in sys.argv[1:]] or [8]:
    m_blk, n, nnz_blk = bench.C4_M // N, bench.C4_N, bench.C4_NNZ // N
    m = m_blk * N
    t0 = time.time()
    counts = torch.zeros(n, dtype=torch.int64, device=dev)
    own = None
    for r in range(N):                                   # pass 1: item counts of the whole matrix -> nnz-balanced item blocks
        row, col, val = bench.synth_block_torch(m_blk, n, nnz_blk, seed=40 + r, item_seed=4, device=dev)
        counts += torch.bincount(col.long(), minlength=n)
        if r == 0:
            own = (row, col, val)
        else:
            del row, col, val
    cb = balanced_boundaries(counts.cpu().numpy(), N)
    c0, c1 = int(cb[0]), int(cb[1])
    rows, cols, vals = [], [], []
    for r in range(N):                                   # pass 2: the entries of rank 0's items, in source-rank order
        row, col, val = own if r == 0 else bench.synth_block_torch(m_blk, n, nnz_blk, seed=40 + r, item_seed=4, device=dev)
        keep = (col >= c0) & (col < c1)
        rows.append((row[keep].long() + r * m_blk).to(torch.int32)); cols.append((col[keep] - c0).to(torch.int32)); vals.append(val[keep])
        if r:
            del row, col, val
    crow = torch.cat(rows); ccol = torch.cat(cols); cval = torch.cat(vals)
    del rows, cols, vals
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    for parts in (1, int(os.environ.get("CMFREC_HIP_AG_PARTS", "4"))):
        s = AlsSession(m, n, K, implicit=True, dtype=np.float32, lam=bench.LAM, use_cg=True, max_cg_steps=3,
                       row_range=(0, m_blk), col_range=(c0, c1), device=0)
        s.set_X_coo_device("r", own[0], own[1], own[2])
        s.set_X_coo_device("c", ccol, crow, cval)
        if parts > 1:
            s.set_A_parts_resident(parts)
        g = torch.Generator(device=dev); g.manual_seed(1)
        A = torch.rand((m, K), generator=g, device=dev, dtype=torch.float32) * 2.0 ** -7
        B = torch.rand((n, K), generator=g, device=dev, dtype=torch.float32) * 2.0 ** -7
        s.set_factors(A=A.cpu().numpy(), B=B.cpu().numpy())
        del A, B
        for _ in range(2):
            s.update("B"); s.update("A")
        s.sync(); s.reset_timers()
        t1 = time.perf_counter()
        for _ in range(5):
            s.update("B"); s.update("A")
        s.sync()
        wall = (time.perf_counter() - t1) / 5
        a, ca = s.kernel_time("A"); b, cb_ = s.kernel_time("B")
        cnt = torch.bincount(ccol.long(), minlength=c1 - c0)
        print("N=%d, A-step in %d part(s): %.2f ms per iteration on rank 0 (B-step %.2f ms: %d items, %.1f M entries, %.0f %% of them in rows > 1024; "
              "A-step %.2f ms: %d users, %.1f M entries); generation %.0f s"
              % (N, parts, wall * 1e3, b / cb_, c1 - c0, cval.numel() / 1e6, 100.0 * float(cnt[cnt > 1024].sum()) / max(cval.numel(), 1),
                 a / ca, m_blk, own[2].numel() / 1e6, t_gen), flush=True)
        for which in ("B", "A"):
            line = []
            for bin_ in range(6):
                ms, c, r_, z = s.bin_stats(which, bin_)
                if c:
                    line.append("bin %d %.3f ms (%d rows, %.1f M nnz)" % (bin_, ms / c, r_, z / 1e6))
            print("     %s: %s" % (which, "; ".join(line)), flush=True)
        s.close()
    del own, crow, ccol, cval
    torch.cuda.empty_cache()
