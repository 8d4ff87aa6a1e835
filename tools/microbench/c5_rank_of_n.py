"""What ONE rank of `bench.py --gpus 8 --workload c5` holds and computes per iteration of BASELINE config 5 (CMF_explicit ALS-Chol
k = 256 fp32, 100 M x 1 M, 2e9 entries, 512 dense side-information columns on both sides, biases), built and measured on ONE GPU
without collectives (the config-4 analogue is c4_rank_of_n.py): rank 0's shard exactly as
GpuEngine.from_collective_block builds it --
    A replica 100 M x 257 (102.8 GB), B replica 1 M x 257, its user block of X (12.5 M users, 250 M entries) as CSR over all items,
    its nnz-balanced item block as CSC over the users of ALL 8 blocks (the other ranks' blocks are drawn here one after the other and
    only the entries of rank 0's items are kept), its rows of U (12.5 M x 512 = 25.6 GB, drawn on the device) and of I --
then times the four updates of an iteration (C and D as partial sums + finish, B, A), reports the memory high-water mark, and checks
what the size-independent properties of the path offer: sampled rows satisfy their normal equations in float64, the factors are
finite, and a second pass over the same half-steps from the same state is bit-identical.
    python tools/microbench/c5_rank_of_n.py [N=8] [scale=1.0]        (scale < 1 shrinks users and items alike: a smoke run)
`run(N, scale, ...)` is what tests/test_gpu_fullsize.py::test_c5_rank_true_share calls."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def run(N=8, scale=1.0, timed_iters=2, n_check=24, verbose=True):
    import torch
    import bench
    from cmfrec_amd.session import AlsSession
    from cmfrec_amd.distributed import GpuEngine, balanced_boundaries
    dev = torch.device("cuda", 0)
    k, p, q = bench.C5_K, bench.C5_P, bench.C5_P
    free0, total = torch.cuda.mem_get_info()
    low_water = [free0]

    def mark():
        torch.cuda.synchronize()
        low_water[0] = min(low_water[0], torch.cuda.mem_get_info()[0])

    t0 = time.time()
    d = bench.c5_shard_data(scale, 0, N, dev)
    m_blk, m, n = d["m_blk"], d["m"], d["n"]
    # pass 1: item counts of the whole matrix -> nnz-balanced item blocks (every rank of the real run gets them by an all-reduce)
    counts = torch.bincount(d["col"].long(), minlength=n)
    for r in range(1, N):
        o = bench.c5_shard_data(scale, r, N, dev)
        counts += torch.bincount(o["col"].long(), minlength=n)
        del o
    cb = balanced_boundaries(counts.cpu().numpy(), N)
    c0, c1 = int(cb[0]), int(cb[1])
    # pass 2: the entries of rank 0's items from all N user blocks, in source-rank order (what the all-to-all delivers)
    rows, cols, vals = [], [], []
    for r in range(N):
        o = d if r == 0 else bench.c5_shard_data(scale, r, N, dev)
        keep = (o["col"] >= c0) & (o["col"] < c1)
        rows.append((o["row"][keep].long() + r * m_blk).to(torch.int32)); cols.append((o["col"][keep] - c0).to(torch.int32))
        vals.append(o["val"][keep])
        if r:
            del o
    crow, ccol, cval = torch.cat(rows), torch.cat(cols), torch.cat(vals)
    del rows, cols, vals, counts
    torch.cuda.synchronize()
    torch.cuda.empty_cache()              # the generator's temporaries are not part of a rank's footprint
    t_gen = time.time() - t0
    mark()
    t0 = time.time()
    row_ranges = d["row_ranges"]
    col_ranges = [(int(cb[r]), int(cb[r + 1])) for r in range(N)]
    sess = AlsSession(m, n, k, implicit=False, dtype=np.float32, lam=0.05, use_cg=False, user_bias=True, item_bias=True, scale_lam=True,
                      p=p, q=q, m_u=m, n_i=n, w_user=1.0, w_item=1.0, row_range=row_ranges[0], col_range=col_ranges[0], device=0)
    mark()
    sess.set_X_coo_device("r", d["row"].to(torch.int32), d["col"].to(torch.int32), d["val"])
    sess.set_X_coo_device("c", ccol, crow, cval)
    mark()
    sess.set_sideinfo_local(d["U"], d["item_rows"](c0, c1))
    mark()
    eng = GpuEngine(sess, row_ranges, col_ranges)
    fA, fB = eng.full("A"), eng.full("B")
    vC, vD = _view(sess, "C", (p, k)), _view(sess, "D", (q, k))

    def init_state():
        g = torch.Generator(device=dev); g.manual_seed(900)
        fA.normal_(generator=g); fA.mul_(2.0 ** -7)                  # every block's rows: the B-step gathers rows of all users
        fB.normal_(generator=g); fB.mul_(2.0 ** -7)
        fA[:, k:].zero_(); fB[:, k:].zero_()                          # bias columns start at zero
        torch.cuda.synchronize()                                      # torch fills on its stream, the session launches on its own
        eng.after_gather("A"); eng.after_gather("B")
        sess.sync()
        vC.normal_(generator=g); vC.mul_(0.05); vD.normal_(generator=g); vD.mul_(0.05)
        sess.sync(); torch.cuda.synchronize()

    init_state()
    sess.sync(); torch.cuda.synchronize()
    t_setup = time.time() - t0
    mark()
    nnz_own, nnz_items = int(d["val"].numel()), int(cval.numel())
    keepU = d["U"]; own = (d["row"], d["col"], d["val"])
    del d

    ev = lambda: torch.cuda.Event(enable_timing=True)
    st = torch.cuda.ExternalStream(int(sess.stream()))

    def timed(fn):
        a, b = ev(), ev()
        a.record(st); fn(); b.record(st)
        return a, b

    def iteration(record):
        out = {}
        for which in ("C", "D"):
            e = timed(lambda: (sess.sideinfo_partial(which), sess.sideinfo_finish(which)))
            out[which] = e
        out["B"] = timed(lambda: (sess.update("B"), sess.after_gather("B")))
        out["A"] = timed(lambda: (sess.update("A"), sess.after_gather("A")))
        if record is not None:
            record.append(out)

    def state_hash():
        r0, r1 = row_ranges[0]
        blkA = fA[r0:r1].view(torch.int32); blkB = fB[c0:c1].view(torch.int32)
        return (int(blkA.sum(dtype=torch.int64).item()), int(blkB.sum(dtype=torch.int64).item()), int(blkA[::997].abs().sum(dtype=torch.int64).item()))

    # one iteration twice from the same state (bit reproducibility; also the warm-up), then the timed ones
    iteration(None); sess.sync(); mark()
    h1 = state_hash()
    dbg = None
    if os.environ.get("C5_RANK_DEBUG"):
        dbg = (fA[row_ranges[0][0]:row_ranges[0][1]].clone(), fB[c0:c1].clone(), vC.clone(), vD.clone(), _view(sess, "a", (m,)).clone(), _view(sess, "b", (n,)).clone())
    init_state()
    iteration(None); sess.sync()
    h2 = state_hash()
    if dbg is not None:
        cur = (fA[row_ranges[0][0]:row_ranges[0][1]], fB[c0:c1], vC, vD, _view(sess, "a", (m,)), _view(sess, "b", (n,)))
        for name, x, y in zip(("A block", "B block", "C", "D", "biasA", "biasB"), dbg, cur):
            diff = (x != y)
            nd = int(diff.sum().item())
            msg = "%s: %d of %d elements differ" % (name, nd, x.numel())
            if nd and x.dim() == 2:
                rr = diff.any(1).nonzero().flatten(); cc = diff.any(0).nonzero().flatten()
                msg += "; rows %d (first %s), columns %d (first %s), max abs diff %.3e" % (rr.numel(), rr[:5].tolist(), cc.numel(), cc[:5].tolist(), float((x - y).abs().max().item()))
            print(msg, flush=True)
        del dbg
    rec = []
    for _ in range(timed_iters):
        iteration(rec)
    sess.sync(); torch.cuda.synchronize(); mark()
    ms = {w: float(np.mean([r[w][0].elapsed_time(r[w][1]) for r in rec])) for w in ("C", "D", "B", "A")}
    lr_rows, lr_eig = sess.lowrank_info()
    kt = k + 1
    flA = nnz_own * kt * (kt + 1) + m_blk * (kt ** 3 / 3 + 2 * kt * kt)
    flB = nnz_items * kt * (kt + 1) + (c1 - c0) * (kt ** 3 / 3 + 2 * kt * kt)
    res = {"N": N, "scale": scale, "users_in_block": m_blk, "items_in_block": c1 - c0, "entries_user_block": nnz_own,
           "entries_item_block": nnz_items, "replica_rows": {"A": m, "B": n},
           "ms": {w: round(v, 2) for w, v in ms.items()}, "ms_per_iteration": round(sum(ms.values()), 2),
           "TFLOPs": {"A": round(flA / (ms["A"] * 1e-3) / 1e12, 1), "B": round(flB / (ms["B"] * 1e-3) / 1e12, 1)},
           "lowrank_rows": lr_rows, "lowrank_eig": {0: "not taken", 2: "one-workgroup Jacobi", 3: "tridiagonalisation + QL (own)"}.get(lr_eig, lr_eig),
           "memory_GB": {"device_total": round(total / 1e9, 1), "free_before": round(free0 / 1e9, 1),
                         "high_water_used": round((free0 - low_water[0]) / 1e9, 1)},
           "generation_s": round(t_gen, 1), "setup_s": round(t_setup, 1)}
    if verbose:
        print(json.dumps(res), flush=True)

    # ---- properties ----
    checks = {}
    r0, r1 = row_ranges[0]
    checks["finite"] = bool(torch.isfinite(fA[r0:r1]).all().item() and torch.isfinite(fB[c0:c1]).all().item())
    # sampled user rows of the block after the A-step: (sum_j b_j b_j^T + w C^T C (+) 0 + lam max(n_row, 1) I) a = w C^T u (+) 0 + sum_j (x_j - biasB_j) b_j
    Cd, Dd = vC.double(), vD.double()                                 # (C, D as the A-step of the last iteration saw them)
    lam, w = 0.05, 1.0
    rowo, colo, valo = own
    ucnt = torch.bincount(rowo.long(), minlength=m_blk)
    gs = torch.Generator(device="cpu"); gs.manual_seed(3)
    users = torch.cat([torch.argsort(ucnt, descending=True)[:3].cpu(), torch.randint(0, m_blk, (n_check,), generator=gs)])
    biasB, biasA = _view(sess, "b", (n,)).double(), _view(sess, "a", (m,)).double()
    worst = 0.0
    CtC = w * Cd.T @ Cd
    for u in users.tolist():
        e = (rowo == u).nonzero().flatten()
        Bj = torch.cat([fB[colo[e].long(), :k].double(), torch.ones((e.numel(), 1), device=dev, dtype=torch.float64)], 1)
        x = valo[e].double() - biasB[colo[e].long()]
        M = Bj.T @ Bj + lam * max(int(e.numel()), 1) * torch.eye(kt, device=dev, dtype=torch.float64)
        M[:k, :k] += CtC
        rhs = Bj.T @ x
        rhs[:k] += w * (keepU[u].double() @ Cd)
        sol = torch.cat([fA[r0 + u, :k].double(), biasA[r0 + u].reshape(1)])
        resid = float((M @ sol - rhs).abs().max().item())
        scale_ = max(1.0, float(rhs.abs().max().item()), float(M.abs().max().item()) * float(sol.abs().max().item()))
        worst = max(worst, resid / scale_)
    checks["user_rows_normal_equations_relres"] = worst
    checks["user_rows_checked"] = int(users.numel())
    checks["reproducible"] = (h1 == h2)
    mark()
    res["checks"] = checks
    res["memory_GB"]["high_water_used"] = round((free0 - low_water[0]) / 1e9, 1)
    if verbose:
        print(json.dumps({"checks": checks, "memory_GB": res["memory_GB"]}), flush=True)
    sess.close()
    return res


def _view(sess, which, shape):
    """torch view of a session-owned device array ('C', 'D': [p, k_user + k]; 'a', 'b': the bias vectors)."""
    import torch
    from cmfrec_amd.distributed import _DevArray
    ptr, _, _ = sess.device_ptr(which)
    return torch.as_tensor(_DevArray(ptr, shape, sess.dtype), device="cuda")


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    sc = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    run(N, sc)
