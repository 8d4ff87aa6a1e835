#!/usr/bin/env python3
"""bench.py -- ALS iteration throughput of the MI355X-native cmfrec hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ALS iteration (B half-step + A half-step, src/collective.c:9827-10045) over a
synthetic implicit-feedback matrix with the shape/nnz of LastFM-360K (BASELINE.json configs[1]:
CMF_implicit, ALS-CG, k=50, fp64, lambda=5, max_cg_steps=3), factors and CSR/CSC resident in HBM
when the timed region starts.  value = rows/s = (users + items) / iteration time.

N > 1: BASELINE.json configs[3] -- CMF_implicit ALS-CG k=64 fp32 on a synthetic 10M users x 1M items
matrix with 5e8 entries, STRONG scaling: the problem is fixed, every rank generates its user block
(10M / N users, 5e8 / N entries) on its own GPU, item blocks are cut nnz-balanced, the entries of a
rank's items arrive through one all-to-all, and every half-step updates the local block and is
followed by an all-gather of the updated factor rows over RCCL (torch.distributed, backend nccl),
enqueued on the session's stream (no host synchronisation inside the loop; the A-step's all-gather
runs part by part beside the kernels of the following parts).

Prints ONE JSON line (rank 0) with the contract fields + "roofline" + "cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs[1] / SURVEY.md 8d "C2"
M_USERS, N_ITEMS, NNZ, K = 358_858, 160_112, 17_309_518, 50
LAM, MAX_CG_STEPS = 5.0, int(os.environ.get("BENCH_CG_STEPS", "3"))   # env override: experiments only
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GATHER_CEILING_GBS = 6000.0    # measured: random 400-byte rows into register tiles, matrix resident in the Infinity Cache


def synth_block(m, n, nnz, seed):
    """SURVEY.md 8d generator: lognormal row weights, 1/rank^0.8 column weights (permuted),
    1.35*nnz draws, unique, permute, keep nnz; counts = ceil(lognormal(3, 1.5))."""
    rng = np.random.default_rng(seed)
    rw = rng.lognormal(0, 1, m); rw /= rw.sum()
    cw = 1.0 / np.arange(1, n + 1) ** 0.8; cw = rng.permutation(cw); cw /= cw.sum()
    draws = int(1.35 * nnz)
    rcdf = np.cumsum(rw); ccdf = np.cumsum(cw)
    r = np.minimum(np.searchsorted(rcdf, rng.random(draws)), m - 1).astype(np.int64)
    c = np.minimum(np.searchsorted(ccdf, rng.random(draws)), n - 1).astype(np.int64)
    lin = np.unique(r * n + c)
    lin = rng.permutation(lin)[:nnz]
    if len(lin) < nnz:
        raise RuntimeError("generator produced too few unique pairs")
    row = (lin // n).astype(np.int32); col = (lin % n).astype(np.int32)
    val = np.ceil(rng.lognormal(3, 1.5, nnz))
    return row, col, val


def to_csr(row, col, val, nrows):
    """Stable COO -> CSR (entries keep COO order inside a row, like helpers.c:1375-1491)."""
    order = np.argsort(row, kind="stable")
    indptr = np.zeros(nrows + 1, np.uint64)
    np.cumsum(np.bincount(row, minlength=nrows), out=indptr[1:])
    return indptr, col[order].astype(np.int32), val[order]


def algorithmic_bytes(nnz, rows, k, w=8):
    """bytes_half = nnz*k*w (gather, once) + nnz*(4+w) (CSR) + (rows+1)*8 (indptr)
    + 2*rows*k*w (warm start in, result out) + k*k*w (BtB)."""
    return nnz * k * w + nnz * (4 + w) + (rows + 1) * 8 + 2 * rows * k * w + k * k * w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale-point", action="store_true", help="skip the N = 1 point of the scaling series (config 4 on one GPU)")
    ap.add_argument("--no-side-points", action="store_true", help="skip the other BASELINE configurations (c1, c3, c4shard, c5shard, whole fit)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debug only; invalid as a result)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded engine + collectives even with 1 rank (test)")
    ap.add_argument("--item-blocks", default="dealt", choices=["dealt", "contiguous"],
                    help="config 4 on N ranks: item blocks equal in rows and balanced in nnz (items renumbered; B's all-gather lands in the "
                         "replica directly) or contiguous nnz-balanced blocks of unequal size (padded staging all-gather)")
    ap.add_argument("--allgather", default=None, choices=["collective", "p2p"],
                    help="N ranks: how the updated row blocks travel -- RCCL all-gathers (default; or CMFREC_ALLGATHER) or direct "
                         "placement, every rank sending its block to each peer over their own xGMI link in one group of "
                         "point-to-point transfers (cmfrec_amd/distributed.py, ShardedAls)")
    ap.add_argument("--implicit-features", action="store_true", help="side workloads c1 / c3: add the implicit-features matrices Ai, Bi")
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "fit", "c4shard", "c5shard", "c5"],
                    help="c2 (default, the metric's config): implicit CG LastFM shape; c1 / c3: the explicit "
                         "MovieLens10M-shaped configs of BASELINE.json (single GPU, side measurements)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from cmfrec_amd.session import AlsSession
    from cmfrec_amd.distributed import ShardedAls, GpuEngine
    if args.workload == "c5":
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        return c5_distributed(args, rank, world, local_rank)
    if use_dist and args.workload == "c2":
        return c4_distributed(args, rank, world, local_rank)
    if args.workload == "fit":
        return whole_fit(args)
    if args.workload == "c4shard":
        return c4_shard(args, local_rank)
    if args.workload == "c5shard":
        return c5_shard(args, local_rank)
    if args.workload != "c2":
        return side_workload(args, local_rank)

    m_blk = int(M_USERS * args.scale); n = int(N_ITEMS * args.scale); nnz_blk = int(NNZ * args.scale)
    m = m_blk * world
    t0 = time.time()
    row, col, val = synth_block(m_blk, n, nnz_blk, seed=2 + rank)       # C2 uses seed 2
    t_gen = time.time() - t0

    rng = np.random.default_rng(100 + rank)
    A0_blk = rng.random((m_blk, K)) * 2.0 ** -7        # uniform start like the reference (collective.c:9762)
    if not use_dist:
        csr = to_csr(row, col, val, m_blk)
        csc = to_csr(col, row, val, n)
        sess = AlsSession(m, n, K, implicit=True, dtype=np.float64, lam=LAM, use_cg=True, max_cg_steps=MAX_CG_STEPS,
                          device=local_rank)
        sess.set_X(csr, csc)
        sess.set_factors(A=A0_blk, B=np.zeros((n, K)))
        engine = None

        def step():
            sess.update("B"); sess.update("A")

        def sync():
            sess.sync()
    else:
        raise SystemExit("multi-rank runs take the c4_distributed path")

    for _ in range(args.warmup):
        step()
    sync()
    sess.reset_timers()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    rows_per_s = (m + n) / (elapsed / args.steps)

    # ---- roofline of the dominant kernel (per nnz-bin launch of the CG row kernel) ----
    names = {0: "vh_pass+vh_update x4 (rows > 1024 nnz, split rows)", 1: "cg_rows_kernel<W=8> (257..1024 nnz)",
             2: "cg_rows_kernel<W=4> (129..256 nnz)", 3: "cg_rows_kernel<W=2> (65..128 nnz)",
             4: "cg_rows_kernel<W=1> (33..64 nnz)", 5: "cg_rows_tiny_kernel (<= 32 nnz; 16-slot tile for rows of <= 16)"}
    kernels = []
    for which in ("B", "A"):
        for b in range(6):
            ms, cnt, rows_b, nnz_b = sess.bin_stats(which, b)
            if cnt:
                name_b = names[b]
                if b == 0 and sess.vh_mode(which) == 2:
                    name_b = "gram_wave+gram_cg (rows > 1024 nnz, single gather, CG on the row's Gramian)"
                kernels.append(dict(step=which, kernel=name_b, ms_total=ms, launches=cnt, rows=rows_b, nnz=nnz_b,
                                    avg_ms=ms / cnt, alg_bytes=algorithmic_bytes(nnz_b, rows_b, K),
                                    overlapped=sess.bin_overlaps(which, b)))
    msA, cntA = sess.kernel_time("A"); msB, cntB = sess.kernel_time("B")
    halfstep_ms = {"A": msA / max(cntA, 1), "B": msB / max(cntB, 1)}
    iter_bytes = sum(d["alg_bytes"] for d in kernels)
    side_by_side = all(d["overlapped"] for d in kernels)
    if side_by_side:
        # Round 3: the nnz bins of a half-step run side by side on several streams (cmfrec_amd/csrc/device.hpp, launch_cg_S), so
        # a bin's event pair spans a time in which it shares the chip -- it has no duration of its own.  The unit that does is the
        # half-step: ONE group of launches forked from and joined on the session's stream, timed by the event pair around it on
        # that stream.  The dominant launch group is the longer half-step; its algorithmic bytes are the sum over its bins.
        w_dom = max(("A", "B"), key=lambda w: halfstep_ms[w])
        grp = [d for d in kernels if d["step"] == w_dom]
        dom = dict(step=w_dom, kernel="%s half-step: %d nnz-bin launches side by side on %s streams (%s)" % (
            w_dom, len(grp), os.environ.get("CMFREC_HIP_BINS_PAR", "2"), " | ".join(d["kernel"].split(" (")[0] for d in grp)),
                   avg_ms=halfstep_ms[w_dom], alg_bytes=sum(d["alg_bytes"] for d in grp), nnz=sum(d["nnz"] for d in grp),
                   overlapped=False, group=grp)
        tr = [pmc_traffic(d) for d in grp] if args.scale == 1.0 and world == 1 else [None]
        traffic = round(sum(tr)) if all(t is not None for t in tr) else None
    else:
        # launches that run beside other kernels (few split rows on the second stream) have no duration of their own
        dom = max([d for d in kernels if not d["overlapped"]], key=lambda d: d["ms_total"])
        traffic = pmc_traffic(dom) if args.scale == 1.0 and world == 1 else None
    achieved = dom["alg_bytes"] / (dom["avg_ms"] * 1e-3) / 1e9
    # how a per-bin figure maps onto `rocprofv3 --kernel-trace --stats` rows (those average over BOTH half-steps)
    prof_names = {0: ["vh_pass_kernel<..., 0> x1 + vh_pass_kernel<..., 1> x%d" % MAX_CG_STEPS,
                      "vh_update_kernel<..., 0> x1 + vh_update_kernel<..., 1> x%d" % MAX_CG_STEPS],
                  1: ["cg_rows_kernel<double, 7, true, 8, 1, false, 0>"], 2: ["cg_rows_kernel<double, 7, true, 4, 1, false, 0>"],
                  3: ["cg_rows_kernel<double, 7, true, 2, 1, false, 0>"], 4: ["cg_rows_kernel<double, 7, true, 1, 4, false, 0>"],
                  5: ["cg_rows_tiny_kernel<double, 7, true, false, 0>"]}
    inv = {v: b for b, v in names.items()}
    inv["gram_wave+gram_cg (rows > 1024 nnz, single gather, CG on the row's Gramian)"] = 6
    prof_names[6] = ["gram_wave_kernel<double, true, 2, false>", "gram_cg_kernel<double, true>"]
    for d in kernels:
        other = [e for e in kernels if e["kernel"] == d["kernel"] and e["step"] != d["step"]]
        # (a launch that runs beside other kernels has no duration of its own, and rocprof's average mixes it in:
        #  CMFREC_HIP_BINS_PAR=1 keeps every launch in line for a clean cross-check, profiles/README.md)
        mixed = d["overlapped"] or (other and other[0]["overlapped"])
        d["rocprof"] = {"kernels": prof_names[inv[d["kernel"]]],
                        "avg_ms_over_both_halfsteps": None if mixed else
                        round((d["avg_ms"] + (other[0]["avg_ms"] if other else 0.0)) / (2 if other else 1), 4)}
    if side_by_side:
        dom["rocprof"] = {"kernels": sorted({n for d in dom["group"] for n in d["rocprof"]["kernels"]}),
                          "avg_ms_over_both_halfsteps": None,
                          "note": "kernels of one half-step overlap: rocprofv3's per-kernel durations are spans, their sum exceeds the "
                                  "half-step; CMFREC_HIP_BINS_PAR=1 puts the bins in line again for a per-kernel cross-check (profiles/README.md)"}
    roofline = dict(bound="hbm", kernel=dom["kernel"] if side_by_side else "%s, %s-step" % (dom["kernel"], dom["step"]), achieved=round(achieved, 1),
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_source=pmc_meta() if traffic is not None else None,
                    alg_bytes_per_launch=dom["alg_bytes"],
                    avg_launch_ms=round(dom["avg_ms"], 4), rocprof=dom["rocprof"],
                    # what a gather of 400-byte rows at random positions reaches on this part with nothing else to do
                    # (tools/microbench/gather_rate.hip: 6.0 TB/s out of the 256 MiB Infinity Cache, 5.1 TB/s from HBM)
                    gather_ceiling={"GBps": GATHER_CEILING_GBS, "frac": round(achieved / GATHER_CEILING_GBS, 4),
                                    "source": "profiles/r02_q_gather_rate.txt (opposing matrices of C2: 144 MB / 64 MB, cache resident)"},
                    iteration={"alg_GB": round(iter_bytes / 1e9, 3), "halfstep_ms": halfstep_ms,
                               "frac_of_hbm_peak": round(iter_bytes / 1e9 / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * world), 4)},
                    per_kernel=[dict(step=d["step"], kernel=d["kernel"], avg_ms=round(d["avg_ms"], 4),
                                     GBps=round(d["alg_bytes"] / (d["avg_ms"] * 1e-3) / 1e9, 1), rocprof=d["rocprof"],
                                     **({"runs_beside_other_kernels": True} if d["overlapped"] else {}))
                                for d in kernels])

    # ---- the same kernels with the bins in line: every bin's own duration (what rocprofv3's per-kernel averages show) ----
    if side_by_side and os.environ.get("CMFREC_HIP_BINS_PAR") is None:
        def _name(which, b):
            return ("gram_wave+gram_cg (rows > 1024 nnz, single gather, CG on the row's Gramian)"
                    if b == 0 and sess.vh_mode(which) == 2 else names[b])
        tab, hs_inline = inline_bin_leg(sess, step, sync, 5, _name, K, 8)
        by_key = {(r["step"], r["kernel"]): r for r in tab}
        for e in roofline["per_kernel"]:
            r = by_key.get((e["step"], e["kernel"]))
            if r is None:
                continue
            o = by_key.get(("A" if e["step"] == "B" else "B", e["kernel"]))
            e["inline_ms"] = r["inline_ms"]; e["inline_GBps"] = r["GBps"]; e["inline_frac"] = r["frac"]
            e["rocprof"]["avg_ms_over_both_halfsteps"] = round((r["inline_ms"] + (o["inline_ms"] if o else 0.0)) / (2 if o else 1), 4)
            e["rocprof"]["mode"] = "bins in line (CMFREC_HIP_BINS_PAR=1): the run profiles/<round>/*_kernel_stats_inline.csv is taken from"
        worst = min(tab, key=lambda r: r["frac"])
        roofline["inline"] = {"halfstep_ms": hs_inline, "iteration_ms": round(hs_inline["A"] + hs_inline["B"], 4),
                              "worst_bin": {k2: worst[k2] for k2 in ("step", "kernel", "inline_ms", "GBps", "frac")},
                              "note": "5 iterations after the timed region with the nnz bins one after the other; not part of `value`"}
        # ONE kernel with a duration of its own: the launch that takes the most time when the bins run in line (HIP events around it
        # on its own stream; rocprofv3's per-kernel average of the in-line run under profiles/ must agree)
        big = max(tab, key=lambda r: r["inline_ms"])
        roofline["dominant_inline"] = {"step": big["step"], "kernel": big["kernel"], "rows": big["rows"], "nnz": big["nnz"],
                                       "avg_launch_ms": big["inline_ms"], "alg_bytes_per_launch": algorithmic_bytes(big["nnz"], big["rows"], K),
                                       "achieved": big["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": big["frac"],
                                       "share_of_inline_iteration": round(big["inline_ms"] / (hs_inline["A"] + hs_inline["B"]), 3)}

    if not side_by_side and dom["kernel"].startswith("gram_wave"):
        # the Gramian path reads its rows once; what limits it in double precision is v_mfma_f64_16x16x4 (docs/DESIGN_HISTORY.md 3.1):
        # 10 tiles of the upper triangle per 4 entries, 2 x 16 x 16 x 4 flops each
        mf = 10 * 2 * 16 * 16 * dom["nnz"] / (dom["avg_ms"] * 1e-3) / 1e12
        roofline["matrix_pipe"] = {"executed_TFLOPs": round(mf, 1), "peak_fp64_TFLOPs": 78.6, "frac": round(mf / 78.6, 3),
                                   "note": "flops the kernel issues (16-wide tiles over k = 50 and full diagonal tiles included); "
                                           "tools/microbench/mfma_rate.hip measures 47 TFLOP/s for this instruction on the part"}
    # ---- CPU baseline: the reference itself (oracle/_ref) on this host, rank 0 / N=1 only ----
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def compare(ref):
            # the same iterations from the same start on the GPU, against the CPU run's factors (whole fit 1e-6 in fp64)
            sess.set_factors(A=A0_blk, B=np.zeros((n, K)))
            for _ in range(ref["iterations"]):
                step()
            sync()
            f = sess.get_factors()
            Ag, Bg = f["A"], f["B"]
            def rel(x, y):
                return float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
            def worst_row(x, y):
                d = np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-300)
                return float(d.max())
            return {"against": "cmfrec's own optimizeA_implicit (oracle/_ref, kind '%s', %d threads)" % (ref["kind"], ref["nthreads"])
                               if ref["kind"] == "reference" else "the C restatement (oracle/, kind 'port')",
                    "iterations": ref["iterations"], "rel_frob_A": rel(Ag[:m_blk], ref["A"]), "rel_frob_B": rel(Bg[:n], ref["B"]),
                    "max_row_rel_A": worst_row(Ag[:m_blk], ref["A"]), "max_row_rel_B": worst_row(Bg[:n], ref["B"]),
                    "tolerance": 1e-6, "workload": "the full C2 workload of this line, same start (A0 uniform 2^-7, B0 = 0)"}
        cpu, parity = cpu_baseline(row, col, val, m_blk, n, A0_blk, compare=compare)

    # ---- the N = 1 point of the scaling series: BASELINE.json configs[3] on this one GPU, through the distributed engine ----
    scale_point = None
    if rank == 0 and world == 1 and not use_dist and args.scale == 1.0 and not args.no_scale_point:
        try:
            del sess
            torch.cuda.empty_cache()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            sp = c4_run(args, 0, 1, local_rank, steps=5, warmup=2)
            dist.destroy_process_group()
            scale_point = {k: sp[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "config")}
            rf = sp.get("roofline") or {}
            scale_point["iteration_frac_of_hbm_peak"] = rf.get("iteration", {}).get("frac_of_hbm_peak")
            scale_point["per_bin_inline"] = rf.get("per_bin_inline")
            scale_point["inline"] = rf.get("inline")
        except Exception as e:        # the headline above must not depend on it
            scale_point = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- the other BASELINE configurations on this GPU, one child process each ----
    side = None
    if rank == 0 and world == 1 and not use_dist and args.scale == 1.0 and not args.no_scale_point and not args.no_side_points:
        side = side_points(args)

    if rank == 0:
        out = {"metric": "ALS rows/sec ((users+items)/iteration time), implicit ALS-CG k=50 fp64",
               "value": round(rows_per_s, 1), "unit": "rows/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "CMF_implicit ALS-CG k=50 fp64, LastFM-360K shape (synthetic): %d users x %d items, "
                                      "%d nnz%s; lambda=5, max_cg_steps=3" % (m, n, nnz_blk * world,
                                                                               " (1 LastFM-sized user block per GPU)" if world > 1 else ""),
                          "parallelism": "row-block x%d + all-gather" % world if world > 1 else "single GPU",
                          "gen_seconds": round(t_gen, 1)},
               "roofline": roofline, "cpu_baseline": cpu, "parity_vs_reference": parity, "scale_point": scale_point, "side_points": side}
        if args.scale != 1.0:
            out["config"]["INVALID_scaled_down"] = args.scale
        final_line = json.dumps(out)
    else:
        final_line = None
    if use_dist:
        dist.destroy_process_group()
    emit_last_line(final_line)


def inline_bin_leg(sess, step_fn, sync_fn, nsteps, names, k, itemsize):
    """A few more iterations with the nnz bins of a half-step IN LINE (CMFREC_HIP_BINS_PAR=1, read again through
    cmfrec_hip_reload_switches): only then a bin's HIP-event pair -- recorded on the stream the kernel is launched on -- brackets that kernel alone, and
    only then `rocprofv3 --kernel-trace --stats` of the same kernels has durations of their own to compare with
    (profiles/<round>/*_kernel_stats_inline.csv).  Returns one row per (half-step, bin): its own duration, algorithmic GB/s and
    fraction of the HBM peak -- the per-kernel roofline the side-by-side default cannot show."""
    keep = os.environ.get("CMFREC_HIP_BINS_PAR")
    os.environ["CMFREC_HIP_BINS_PAR"] = "1"
    sess.reload_switches()
    try:
        step_fn(); sync_fn()
        sess.reset_timers()
        for _ in range(nsteps):
            step_fn()
        sync_fn()
        rows = []
        for which in ("B", "A"):
            for b in range(6):
                ms, cnt, rows_b, nnz_b = sess.bin_stats(which, b)
                if cnt:
                    by = algorithmic_bytes(nnz_b, rows_b, k, itemsize)
                    rows.append(dict(step=which, bin=b, kernel=names[b] if not callable(names) else names(which, b), rows=rows_b, nnz=nnz_b,
                                     inline_ms=round(ms / cnt, 4), GBps=round(by / (ms / cnt * 1e-3) / 1e9, 1),
                                     frac=round(by / (ms / cnt * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)))
        msA, cntA = sess.kernel_time("A"); msB, cntB = sess.kernel_time("B")
        return rows, {"A": round(msA / max(cntA, 1), 4), "B": round(msB / max(cntB, 1), 4)}
    finally:
        if keep is None:
            os.environ.pop("CMFREC_HIP_BINS_PAR", None)
        else:
            os.environ["CMFREC_HIP_BINS_PAR"] = keep
        sess.reload_switches()
        sess.reset_timers()


C4_M, C4_N, C4_NNZ, C4_K = 10_000_000, 1_000_000, 500_000_000, 64     # BASELINE.json configs[3] / SURVEY.md 8d "C4"


def synth_block_torch(m, n, nnz, seed, item_seed, device):
    """The SURVEY.md 8d generator on the GPU (torch), for blocks too large to draw on the host in the bench's time
    budget: lognormal row weights, 1/rank^0.8 column weights in a permuted order that is COMMON to all ranks
    (item_seed), 1.35 nnz draws, unique, random subset of nnz; values ceil(lognormal(3, 1.5)).  Returns int32 row
    (local), int32 col, float32 val, all resident on `device`."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    gi = torch.Generator(device=device); gi.manual_seed(item_seed)
    rcdf = torch.cumsum(torch.exp(torch.randn(m, generator=g, device=device, dtype=torch.float64)), 0)
    rcdf /= rcdf[-1].clone()
    cw = 1.0 / torch.arange(1, n + 1, device=device, dtype=torch.float64) ** 0.8
    ccdf = torch.cumsum(cw[torch.randperm(n, generator=gi, device=device)], 0)
    ccdf /= ccdf[-1].clone()
    draws = int(1.35 * nnz)
    r = torch.searchsorted(rcdf, torch.rand(draws, generator=g, device=device, dtype=torch.float64)).clamp_(max=m - 1)
    c = torch.searchsorted(ccdf, torch.rand(draws, generator=g, device=device, dtype=torch.float64)).clamp_(max=n - 1)
    lin = torch.unique(r * n + c)
    del r, c
    if lin.numel() < nnz:
        raise RuntimeError("generator produced too few unique pairs")
    lin = lin[torch.randperm(lin.numel(), generator=g, device=device)[:nnz]]
    row = (lin // n).to(torch.int32); col = (lin % n).to(torch.int32)
    del lin
    val = torch.ceil(torch.exp(3.0 + 1.5 * torch.randn(nnz, generator=g, device=device, dtype=torch.float32)))
    return row, col, val


def c4_distributed(args, rank, world, local_rank):
    import torch.distributed as dist
    out = c4_run(args, rank, world, local_rank, args.steps, args.warmup)
    dist.destroy_process_group()
    emit_last_line(json.dumps(out) if out is not None else None)


def c4_run(args, rank, world, local_rank, steps, warmup):
    """BASELINE.json configs[3]: CMF_implicit ALS-CG k=64 fp32, synthetic 10M users x 1M items, 5e8 entries, row-partitioned
    over the ranks with an all-gather of the updated factor rows after every half-step.  Strong scaling: the problem
    does not change with N (--scale shrinks it for tests only and marks the line invalid)."""
    import torch
    import torch.distributed as dist
    from cmfrec_amd.distributed import ShardedAls, GpuEngine
    dev = torch.device("cuda", local_rank)
    m_blk = int(C4_M * args.scale) // world
    n = int(C4_N * args.scale)
    nnz_blk = int(C4_NNZ * args.scale) // world
    m = m_blk * world
    t0 = time.time()
    row, col, val = synth_block_torch(m_blk, n, nnz_blk, seed=40 + rank, item_seed=4, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    row_ranges = [(r * m_blk, (r + 1) * m_blk) for r in range(world)]
    a_parts = int(os.environ.get("CMFREC_HIP_AG_PARTS", "4")) if world > 1 else 1
    t0 = time.time()
    eng = GpuEngine.from_device_coo(m, n, C4_K, row, col, val, row_ranges, rank, world, local_rank, dtype=np.float32, lam=LAM,
                                    max_cg_steps=MAX_CG_STEPS, a_parts=a_parts, item_blocks=args.item_blocks)
    b_sizes = sorted({e - b for b, e in eng.ranges("B")})
    del row, col, val
    t_setup = time.time() - t0
    g = torch.Generator(device=dev); g.manual_seed(100 + rank)
    fullA = eng.full("A")
    fullA[rank * m_blk:(rank + 1) * m_blk].copy_(torch.rand((m_blk, fullA.shape[1]), generator=g, device=dev, dtype=torch.float32) * 2.0 ** -7)
    eng.full("B").zero_()
    torch.cuda.synchronize()
    engine = ShardedAls(eng, rank, world, exchange=args.allgather)
    engine.allgather("A")
    sess = eng.session

    def sync():
        sess.sync(); torch.cuda.synchronize()

    for _ in range(warmup):
        engine.iteration()
    sync()
    sess.reset_timers()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        engine.iteration()
    sync()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / steps * 1e3
    rows_per_s = (m + n) / (elapsed / steps)
    # roofline of rank 0's dominant launch + the whole job's algorithmic bytes per iteration
    names = {0: "split rows (> 1024 nnz)", 1: "cg_rows_kernel<W=8> (257..1024 nnz)", 2: "cg_rows_kernel<W=4> (129..256 nnz)",
             3: "cg_rows_kernel<W=2> (65..128 nnz)", 4: "cg_rows_kernel<W=1> (33..64 nnz)", 5: "cg_rows_tiny_kernel (<= 32 nnz)"}
    kernels = []
    for which in ("B", "A"):
        for b in range(6):
            ms, cnt, rows_b, nnz_b = sess.bin_stats(which, b)
            if cnt:
                kernels.append(dict(step=which, kernel=names[b], avg_ms=ms / cnt, alg_bytes=algorithmic_bytes(nnz_b, rows_b, C4_K, 4),
                                    overlapped=sess.bin_overlaps(which, b), rows=rows_b, nnz=nnz_b, launches=cnt))
    job_bytes = torch.tensor([float(sum(d["alg_bytes"] for d in kernels))], device="cuda", dtype=torch.float64)
    dist.all_reduce(job_bytes)
    roofline = None
    cand = [d for d in kernels if not d["overlapped"]]
    if not cand and kernels:
        # the bins of a half-step run side by side (see main()): the half-step is the unit with a duration of its own
        msA, cntA = sess.kernel_time("A"); msB, cntB = sess.kernel_time("B")
        hs = {"A": msA / max(cntA, 1), "B": msB / max(cntB, 1)}
        w_dom = max(("A", "B"), key=lambda w: hs[w])
        grp = [d for d in kernels if d["step"] == w_dom]
        cand = [dict(step=w_dom, kernel="%s half-step: %d nnz-bin launches side by side" % (w_dom, len(grp)), avg_ms=hs[w_dom],
                     alg_bytes=sum(d["alg_bytes"] for d in grp))]
    if cand:
        dom = max(cand, key=lambda d: d["avg_ms"])
        ach = dom["alg_bytes"] / (dom["avg_ms"] * 1e-3) / 1e9
        roofline = dict(bound="hbm", kernel="%s, %s-step (rank 0; with A-step parts a bin is launched once per part)" % (dom["kernel"], dom["step"]),
                        achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                        alg_bytes_per_launch=dom["alg_bytes"], avg_launch_ms=round(dom["avg_ms"], 4),
                        iteration={"alg_GB_whole_job": round(float(job_bytes.item()) / 1e9, 3),
                                   "frac_of_hbm_peak": round(float(job_bytes.item()) / 1e9 / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * world), 4)},
                        per_kernel_rank0=[dict(step=d["step"], kernel=d["kernel"], avg_ms=round(d["avg_ms"], 4), rows=d["rows"], nnz=d["nnz"],
                                               launches_timed=d["launches"], GBps=round(d["alg_bytes"] / (d["avg_ms"] * 1e-3) / 1e9, 1))
                                          for d in kernels])
    if world == 1 and roofline is not None and os.environ.get("CMFREC_HIP_BINS_PAR") is None:
        tab, hs_inline = inline_bin_leg(sess, engine.iteration, sync, 2, names, C4_K, 4)
        worst = min(tab, key=lambda r: r["frac"])
        roofline["per_bin_inline"] = tab
        roofline["inline"] = {"halfstep_ms": hs_inline, "iteration_ms": round(hs_inline["A"] + hs_inline["B"], 4),
                              "worst_bin": {k2: worst[k2] for k2 in ("step", "kernel", "inline_ms", "GBps", "frac")},
                              "note": "2 iterations after the timed region with the nnz bins one after the other; not part of `value`"}
    out = None
    if rank == 0:
        out = {"metric": "ALS rows/sec ((users+items)/iteration time), implicit ALS-CG k=64 fp32",
               "value": round(rows_per_s, 1), "unit": "rows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "CMF_implicit ALS-CG k=64 fp32, synthetic %d users x %d items, %d nnz (BASELINE.json configs[3]); "
                                      "lambda=5, max_cg_steps=3" % (m, n, nnz_blk * world),
                          "parallelism": "user / item row blocks x%d (items nnz-balanced, %s: %s rows per block), %s of the "
                                         "updated rows after every half-step, A-step in %d parts" % (
                                             world, args.item_blocks if world > 1 else "one block", "/".join(map(str, b_sizes)),
                                             "direct placement (point-to-point group over RCCL)" if engine.exchange == "p2p" else "RCCL all-gather",
                                             a_parts),
                          "gen_seconds": round(t_gen, 1), "setup_seconds": round(t_setup, 1)},
               "roofline": roofline, "cpu_baseline": None}
        out["config"]["scaling_series"] = ("BASELINE.json configs[3] at N = 1, 2, 4, 8: every --gpus N > 1 line runs THIS workload; its N = 1 point "
                                           "is the \"scale_point\" object of the --gpus 1 line (the same engine and collectives on one rank), which is "
                                           "what a per-N value is to be divided by -- not the headline of --gpus 1 (C2, double precision)")
        if args.scale != 1.0:
            out["config"]["INVALID_scaled_down"] = args.scale
    del engine, eng, sess
    torch.cuda.empty_cache()
    return out


C5_M, C5_N, C5_NNZ, C5_K, C5_P = 100_000_000, 1_000_000, 2_000_000_000, 256, 512     # BASELINE.json configs[4] / SURVEY.md 8d "C5"


def c5_shard_data(scale, rank, world, dev, seed=50):
    """This rank's share of the data of BASELINE.json configs[4], drawn on `dev` (a GPU in the bench, the CPU in the gloo test of
    the set-up path): its user block of X as COO (rows local to the block, global item ids; ratings 0.5 .. 5 minus their mean),
    its rows of U, and a function that returns the rows [c0, c1) of I -- the item side information is a function of the item id
    alone, so every rank can draw the rows of whatever item block the nnz-balanced cut assigns to it."""
    import torch
    m_blk = max(int(C5_M * scale) // world, 64)
    n = max(int(C5_N * scale), 1024)                   # (tiny test scales: enough items for 20 distinct ones per user)
    nnz_blk = (C5_NNZ // C5_M) * m_blk                 # 20 entries per user at every scale
    p = q = C5_P
    row, col, _ = synth_block_torch(m_blk, n, nnz_blk, seed=seed + rank, item_seed=5, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(500 + rank)
    val = 0.5 * torch.randint(1, 11, (len(row),), generator=g, device=dev).to(torch.float32) - 2.75
    U = torch.randn((m_blk, p), generator=g, device=dev, dtype=torch.float32)

    def item_rows(c0, c1):
        gi = torch.Generator(device=dev); gi.manual_seed(7)
        return torch.randn((n, q), generator=gi, device=dev, dtype=torch.float32)[c0:c1].contiguous()

    return dict(m_blk=m_blk, m=m_blk * world, n=n, nnz=nnz_blk * world, row=row, col=col, val=val, U=U, item_rows=item_rows,
                row_ranges=[(r * m_blk, (r + 1) * m_blk) for r in range(world)])


def c5_setup(scale, rank, world, local_rank):
    """The sharded engine of BASELINE.json configs[4] (CMF_explicit ALS-Chol k=256 fp32, 100M x 1M, 2e9 entries, 512 dense
    side-information columns on both sides, user and item biases) for this rank: GpuEngine.from_collective_block on the
    device-generated shard (U / I local, C / D by partial sums + all-reduce), start values set."""
    import torch
    from cmfrec_amd.distributed import GpuEngine
    dev = torch.device("cuda", local_rank)
    d = c5_shard_data(scale, rank, world, dev)
    m, n, k = d["m"], d["n"], C5_K
    eng = GpuEngine.from_collective_block(m, n, k, d["row"], d["col"], d["val"], d["row_ranges"], rank, world, local_rank,
                                          U_local=d["U"], I_local=d["item_rows"], p=C5_P, q=C5_P, m_u=m, n_i=n, dtype=np.float32,
                                          lam=0.05, w_user=1.0, w_item=1.0, user_bias=True, item_bias=True, scale_lam=True)
    fA, fB = eng.full("A"), eng.full("B")
    fA.zero_(); fB.zero_()
    r0, r1 = d["row_ranges"][rank]
    c0, c1 = eng.ranges("B")[rank]
    g = torch.Generator(device=dev); g.manual_seed(900 + rank)
    fA[r0:r1, :k].copy_(torch.randn((r1 - r0, k), generator=g, device=dev, dtype=torch.float32) * 2.0 ** -7)
    gb = torch.Generator(device=dev); gb.manual_seed(9)
    fB[c0:c1, :k].copy_((torch.randn((n, k), generator=gb, device=dev, dtype=torch.float32) * 2.0 ** -7)[c0:c1])
    torch.cuda.synchronize()
    nnz = d["nnz"]
    del d
    return eng, m, n, nnz


def c5_distributed(args, rank, world, local_rank):
    """`--workload c5 --gpus N`: BASELINE.json configs[4] on N ranks through ShardedAls.iteration_collective (C, D by partial sums
    and one all-reduce each; B- and A-step with an all-gather of the updated rows incl. their bias columns).  The whole problem
    needs N >= 8 (A alone is 103 GB per replica, U 205 GB in total); --scale shrinks it for tests and marks the line invalid."""
    import torch
    import torch.distributed as dist
    from cmfrec_amd.distributed import ShardedAls
    t0 = time.time()
    eng, m, n, nnz = c5_setup(args.scale, rank, world, local_rank)
    t_setup = time.time() - t0
    als = ShardedAls(eng, rank, world, exchange=args.allgather)
    als.allgather("A"); als.allgather("B")
    sess = eng.session
    steps, warmup = max(1, min(args.steps, 3)), min(args.warmup, 1)

    def sync():
        sess.sync(); torch.cuda.synchronize()

    for _ in range(warmup):
        als.iteration_collective()
    sync(); sess.reset_timers()
    dist.barrier(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        als.iteration_collective()
    sync()
    dist.barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms = elapsed / steps * 1e3
    msA, cA = sess.kernel_time("A"); msB, cB = sess.kernel_time("B")
    lr_rows, lr_eig = sess.lowrank_info()
    kt = C5_K + 1
    # flops per half-step: Gramians nnz kt (kt + 1) + factorisation kt^3 / 3 + 2 kt^2 per row
    flA = nnz * kt * (kt + 1) + m * (kt ** 3 / 3 + 2 * kt * kt)
    flB = nnz * kt * (kt + 1) + n * (kt ** 3 / 3 + 2 * kt * kt)
    f = sess.get_factors() if m * (kt + 1) < 2e8 else None
    out = None
    if rank == 0:
        out = {"metric": "ALS rows/sec ((users+items)/iteration time), explicit ALS-Chol k=256 fp32 + 512-dim side information",
               "value": round((m + n) / (elapsed / steps), 1), "unit": "rows/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic",
               "config": {"workload": "CMF_explicit ALS-Chol k=256 fp32, synthetic %d users x %d items, %d nnz, 512-dim dense side "
                                      "information on both sides, user + item biases (BASELINE.json configs[4])" % (m, n, nnz),
                          "parallelism": "user / item row blocks x%d (items nnz-balanced), U / I sharded with the rows, C / D by partial "
                                         "sums + one all-reduce each, %s of the updated rows after every half-step" % (
                                             world, "direct placement (point-to-point group)" if als.exchange == "p2p" else "all-gather"),
                          "setup_seconds": round(t_setup, 1)},
               "halfstep_ms_rank0": {"A": msA / max(cA, 1), "B": msB / max(cB, 1)},
               "halfstep_TFLOPs_rank0_share": {"A": round(flA / world / (msA / max(cA, 1) * 1e-3) / 1e12, 1) if cA else None,
                                               "B": round(flB / world / (msB / max(cB, 1) * 1e-3) / 1e12, 1) if cB else None},
               "lowrank_rows_rank0": lr_rows, "lowrank_eig": {0: "not taken", 2: "one-workgroup Jacobi", 3: "tridiagonalisation + QL (own)"}.get(lr_eig, lr_eig),
               "finite": None if f is None else bool(np.isfinite(f["A"]).all() and np.isfinite(f["B"]).all() and np.isfinite(f["C"]).all()),
               "roofline": None, "cpu_baseline": None}
        if args.scale != 1.0:
            out["config"]["INVALID_scaled_down"] = args.scale
    dist.destroy_process_group()
    emit_last_line(json.dumps(out) if out is not None else None)


def emit_last_line(line):
    """The result must be the LAST line on stdout.  RCCL writes a version banner through C stdio, which
    sits in libc's buffer until exit and would land after a Python print(); so: flush Python's buffer,
    flush every C stream (fflush(NULL)), then write the JSON line and flush again."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        print(line, flush=True)


def whole_fit(args):
    """Whole CMF_implicit.fit() of C2 through the drop-in C ABI (fit_collective_implicit_als): host COO in,
    host factors out, 15 iterations, seeded start values -- the PCIe- and preprocessing-inclusive number
    DESIGN.md quotes beside the resident-data metric.  CMFREC_HIP_TIMING=1 prints the host phases."""
    from cmfrec_amd.models import CMF_implicit
    import scipy.sparse as sp
    m, n, nnz = int(M_USERS * args.scale), int(N_ITEMS * args.scale), int(NNZ * args.scale)
    row, col, val = synth_block(m, n, nnz, seed=2)
    X = sp.coo_matrix((val, (row, col)), shape=(m, n))
    niter = 15
    model = CMF_implicit(k=K, lambda_=LAM, niter=niter, use_float=False, use_cg=True, finalize_chol=False,
                         max_cg_steps=MAX_CG_STEPS, precompute_for_predictions=False)
    times = []
    for _ in range(max(1, args.steps // 5)):
        t0 = time.perf_counter()
        model.fit(X)
        times.append(time.perf_counter() - t0)
    print(json.dumps({"workload": "whole fit(), C2, %d iterations, host COO in / host factors out" % niter,
                      "seconds": [round(t, 3) for t in times], "best_s": round(min(times), 3),
                      "rows_per_s_whole_fit": round((m + n) * niter / min(times), 1),
                      "finite": bool(np.isfinite(model.A_).all() and np.isfinite(model.B_).all()),
                      "note": "side measurement, not the headline metric"}))


def c4_shard(args, device):
    """One GPU's share of BASELINE config 4 (10M x 1M, nnz 5e8, k=64 fp32 on 8 GPUs): 1.25M users x 1M items,
    62.5M nnz, implicit ALS-CG in single precision, as a single-GPU problem (no collectives): exercises the
    fp32 kernels and the 64-bit offsets at production size.  Side measurement."""
    from cmfrec_amd.session import AlsSession
    m, n, nnz = int(1_250_000 * args.scale), int(1_000_000 * args.scale), int(62_500_000 * args.scale)
    t0 = time.time()
    row, col, val = synth_block(m, n, nnz, seed=4)
    t_gen = time.time() - t0
    k = 64
    sess = AlsSession(m, n, k, implicit=True, dtype=np.float32, lam=5.0, use_cg=True, max_cg_steps=3, device=device)
    t0 = time.time()
    sess.set_X_coo(row, col, val.astype(np.float32))
    t_setx = time.time() - t0
    rng = np.random.default_rng(7)
    sess.set_factors(A=(rng.random((m, k), dtype=np.float32) * 2.0 ** -7), B=np.zeros((n, k), np.float32))
    for _ in range(args.warmup):
        sess.iterate(1)
    sess.sync(); sess.reset_timers()
    t0 = time.perf_counter()
    sess.iterate(args.steps)
    sess.sync()
    dt = (time.perf_counter() - t0) / args.steps
    msA, cA = sess.kernel_time("A"); msB, cB = sess.kernel_time("B")
    f = sess.get_factors()
    alg = algorithmic_bytes(nnz, m, k, 4) + algorithmic_bytes(nnz, n, k, 4)
    bins = {}
    for which in ("B", "A"):
        for b in range(6):
            ms, cnt, rows_b, nnz_b = sess.bin_stats(which, b)
            if cnt:
                bins["%s%d" % (which, b)] = {"ms": round(ms / cnt, 3), "rows": rows_b, "nnz": nnz_b,
                                             "GBps": round(algorithmic_bytes(nnz_b, rows_b, k, 4) / (ms / cnt * 1e-3) / 1e9, 1)}
    print(json.dumps({"workload": "c4 shard (1/8 of 10M x 1M, nnz 5e8), implicit ALS-CG k=64 fp32, one GPU",
                      "ms_per_iteration": round(dt * 1e3, 3), "rows_per_s": round((m + n) / dt, 1),
                      "halfstep_ms": {"A": msA / max(cA, 1), "B": msB / max(cB, 1)}, "alg_GB": round(alg / 1e9, 2),
                      "frac_of_hbm_peak": round(alg / dt / 8e12, 3), "m": m, "n": n, "nnz": nnz,
                      "bins (0: split rows > 1024 nnz, 1: 257..1024, 2: 129..256, 3: 65..128, 4: 33..64, 5: <= 32)": bins,
                      "gen_seconds": round(t_gen, 1), "set_X_coo_seconds": round(t_setx, 2),
                      "finite": bool(np.isfinite(f["A"]).all() and np.isfinite(f["B"]).all()),
                      "note": "side measurement, not the headline metric"}))


def c5_shard(args, device):
    """An eighth of one GPU's share of BASELINE config 5 (1e8 x 1e6, nnz 2e9, k=256 fp32, 512-dim dense side information
    on both sides, Cholesky, 8 GPUs): 1.5625M users x 125k items, 31.25M nnz (20 per user as in the full problem),
    U[m, 512], I[n, 512], user and item biases, as a single-GPU problem.  Exercises the 17-tile single-precision
    Cholesky kernel in COLLECTIVE mode and the side-information GEMMs at production width.  Side measurement."""
    from cmfrec_amd.session import AlsSession
    sc = 0.125 * args.scale
    m, n, nnz = int(12_500_000 * sc), int(1_000_000 * sc), int(250_000_000 * sc)
    k, p, q = 256, 512, 512
    t0 = time.time()
    row, col, _ = synth_block(m, n, nnz, seed=5)
    rng = np.random.default_rng(5)
    val = (0.5 * rng.integers(1, 11, nnz)).astype(np.float32)
    val -= val.mean()
    U = rng.standard_normal((m, p), dtype=np.float32)
    II = rng.standard_normal((n, q), dtype=np.float32)
    t_gen = time.time() - t0
    sess = AlsSession(m, n, k, implicit=False, dtype=np.float32, lam=0.05, use_cg=False, user_bias=True, item_bias=True,
                      scale_lam=True, p=p, m_u=m, q=q, n_i=n, device=device)
    sess.set_X_coo(row, col, val)
    sess.set_sideinfo(U=U, II=II)
    sess.set_factors(A=rng.standard_normal((m, k), dtype=np.float32) * np.float32(2.0 ** -7),
                     B=rng.standard_normal((n, k), dtype=np.float32) * np.float32(2.0 ** -7),
                     biasA=np.zeros(m, np.float32), biasB=np.zeros(n, np.float32),
                     Cm=np.zeros((p, k), np.float32), Dm=np.zeros((q, k), np.float32))
    steps = max(1, min(args.steps, 3)); warm = min(args.warmup, 1)
    for _ in range(warm):
        sess.iterate(1)
    sess.sync(); sess.reset_timers()
    t0 = time.perf_counter()
    sess.iterate(steps)
    sess.sync()
    dt = (time.perf_counter() - t0) / steps
    msA, cA = sess.kernel_time("A"); msB, cB = sess.kernel_time("B")
    f = sess.get_factors()
    kt = k + 1
    # Gramians nnz*kt*(kt+1) + Cholesky kt^3/3 + 2 kt^2 per row, both half-steps; side-information GEMMs 2*rows*p*k (x3:
    # U C for the right-hand sides, U^T A and the C update)
    flops = 2 * nnz * kt * (kt + 1) + (m + n) * (kt ** 3 / 3 + 2 * kt * kt) + 3 * 2 * (m * p + n * q) * k
    print(json.dumps({"workload": "1/8 of a c5 GPU share (1e8 x 1e6, nnz 2e9, k=256 fp32, p=q=512, Cholesky): %d x %d, %d nnz"
                                  % (m, n, nnz),
                      "ms_per_iteration": round(dt * 1e3, 2), "rows_per_s": round((m + n) / dt, 1),
                      "halfstep_ms": {"A": msA / max(cA, 1), "B": msB / max(cB, 1)},
                      # (the full-Cholesky operation count of SURVEY 8d, NOT executed flops: the user step runs the low-rank path and skips
                      #  most of them -- the figure says how long the same half-steps would take at a given rate, not how busy the pipes are;
                      #  the item step below is counted on the flops it executes)
                      "full_cholesky_TFLOP_surveyed": round(flops / 1e12, 2), "surveyed_TFLOP_per_s_equivalent": round(flops / dt / 1e12, 1),
                      # the item step on its own flops (every item row is a full k_t x k_t system: rank-k update + factorisation);
                      # the user step runs the low-rank path and does far fewer flops than the formula above charges it
                      "item_step": (lambda fl, ms: {"TFLOP": round(fl / 1e12, 3), "ms": round(ms, 2), "TFLOPs": round(fl / ms / 1e9, 1),
                                                    "frac_of_fp32_matrix_peak_157": round(fl / ms / 1e9 / 157.0, 3)})(
                          nnz * kt * (kt + 1) + n * (kt ** 3 / 3.0), msB / max(cB, 1)),
                      "user_step_ms": round(msA / max(cA, 1), 2),
                      # how the item rows split between the low-rank kernels (<= 128 entries) and the full factorisation
                      "item_rows": (lambda cnt: {"le128": int((cnt <= 128).sum()), "full": int((cnt > 128).sum()),
                                                 "nnz_full": int(cnt[cnt > 128].sum()), "max": int(cnt.max())})(np.bincount(col, minlength=n)),
                      "gen_seconds": round(t_gen, 1), "steps": steps,
                      "finite": bool(np.isfinite(f["A"]).all() and np.isfinite(f["B"]).all() and np.isfinite(f["C"]).all()),
                      "note": "side measurement, not the headline metric"}))


def side_workload(args, device):
    """C1: CMF explicit ALS-CG k=50 fp64, biases + centering, lambda=0.05 scale_lam (MovieLens10M shape).
    C3: CMF explicit ALS-Cholesky k=128 fp64 + dense item side info (q=64).  Synthetic ratings with
    the SURVEY.md 8d generator (seed 1).  Prints per-iteration time and the half-step breakdown."""
    from cmfrec_amd.session import AlsSession
    m, n, nnz = int(69_878 * args.scale), int(10_677 * args.scale), int(10_000_054 * args.scale)
    row, col, _ = synth_block(m, n, nnz, seed=1)
    rng = np.random.default_rng(1)
    val = 0.5 * rng.integers(1, 11, nnz)
    val = val - val.mean()
    chol = args.workload == "c3"
    k = 128 if chol else 50
    q = 64 if chol else 0
    sess = AlsSession(m, n, k, implicit=False, dtype=np.float64, lam=0.05, use_cg=not chol, max_cg_steps=3,
                      user_bias=True, item_bias=True, scale_lam=True, q=q, n_i=n if q else 0, device=device)
    sess.set_X(to_csr(row, col, val, m), to_csr(col, row, val, n))
    if q:
        II = rng.standard_normal((n, q)); II -= II.mean(0)
        sess.set_sideinfo(II=II)
    sess.set_factors(A=rng.standard_normal((m, k)) * 2.0 ** -7, B=rng.standard_normal((n, k)) * 2.0 ** -7 if chol else np.zeros((n, k)),
                     biasA=np.zeros(m), biasB=np.zeros(n), Dm=np.zeros((q, k)) if q else None)
    if args.implicit_features:           # the reference's "ALS-CG / Chol + implicit features" benchmark lines (benchmark/README.md:28-29)
        sess.set_implicit_features(0.5)
    import torch
    for _ in range(args.warmup):
        sess.iterate(1)
    sess.sync(); sess.reset_timers()
    t0 = time.perf_counter()
    sess.iterate(args.steps)
    sess.sync()
    dt = (time.perf_counter() - t0) / args.steps
    msA, cA = sess.kernel_time("A"); msB, cB = sess.kernel_time("B")
    f = sess.get_factors()
    kt = k + 1
    extra = {}
    if chol:
        # Gramians nnz*kt*(kt+1) + Cholesky kt^3/3 + 2 kt^2 per row, both half-steps (+ the small side-information GEMMs)
        flops = 2 * nnz * kt * (kt + 1) + (m + n) * (kt ** 3 / 3 + 2 * kt * kt) + 3 * 2 * n * q * k
        gath = 2 * (nnz * kt * 8 + nnz * 12)
        # flops the half-steps EXECUTE: item rows and the user rows beyond 96 entries take the rank-k update + the k_t^3 / 3
        # factorisation; user rows of at most 96 entries (no side information on that side) take the low-rank kernel
        # (session.hip launch_plain_lowrank, CMF_LR_MAX_F64): an s x s Gramian over k_t, its factorisation and two s x k_t products
        ucnt = np.bincount(row, minlength=m).astype(np.float64)
        icnt = np.bincount(col, minlength=n).astype(np.float64)
        full = lambda c: float((c * kt * (kt + 1) + kt ** 3 / 3.0 + 2.0 * kt * kt).sum())
        light = ucnt[(ucnt <= 96) & (ucnt > 0)]
        executed = full(icnt) + full(ucnt[ucnt > 96]) + float((light * (light + 1) * kt + light ** 3 / 3.0 + 4.0 * light * kt).sum()) + 3 * 2 * n * q * k
        extra = {"alg_TFLOP": round(flops / 1e12, 3), "TFLOPs": round(flops / dt / 1e12, 1),
                 "executed_TFLOP": round(executed / 1e12, 3), "executed_frac_of_fp64_peak_78.6": round(executed / dt / 78.6e12, 3),
                 "frac_of_fp64_vector_peak_78.6": round(flops / dt / 78.6e12, 3),
                 "gather_GB": round(gath / 1e9, 2), "gather_frac_of_hbm_peak": round(gath / dt / 8e12, 3)}
    else:
        alg = algorithmic_bytes(nnz, m, kt) + algorithmic_bytes(nnz, n, kt)
        extra = {"alg_GB": round(alg / 1e9, 2), "frac_of_hbm_peak": round(alg / dt / 8e12, 3)}
    print(json.dumps({"workload": args.workload + ("+implicit_features" if args.implicit_features else ""),
                      "ms_per_iteration": round(dt * 1e3, 3), "rows_per_s": round((m + n) / dt, 1),
                      "halfstep_ms": {"A": msA / max(cA, 1), "B": msB / max(cB, 1)}, "k": k, "m": m, "n": n, "nnz": nnz, **extra,
                      "finite": bool(np.isfinite(f["A"]).all() and np.isfinite(f["B"]).all()),
                      "note": "side measurement, not the headline metric"}))


def pmc_meta():
    """Where roofline.traffic comes from: the committed counter file, the round / step it was taken in, the FETCH_SIZE
    correction it used, and whether the CG kernel sources have changed since (stale)."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    h = hashlib.sha1()
    for f in ("cmfrec_amd/csrc/cg_kernels.hpp", "cmfrec_amd/csrc/gram_cg_kernels.hpp", "cmfrec_amd/csrc/lanes.hpp"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    cur = h.hexdigest()[:16]
    return {"file": "profiles/pmc_latest.json", "round": d.get("round", "r01_f"), "fetch_factor": d.get("fetch_factor", 2.0),
            "fetch_factor_source": d.get("fetch_factor_source", "x2 (guide; uncalibrated for this pattern)"),
            "kernel_source_hash": d.get("kernel_source_hash"), "current_source_hash": cur,
            "stale": d.get("kernel_source_hash") != cur}


def pmc_traffic(dom):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json, produced by tools/pmc_summary.py from separate --pmc runs of this
    same command; counters cannot be read from inside the process).  None if not available."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    # (the row kernel's template arguments are <T, slots, implicit, waves per row, rows per workgroup, weighted>)
    tags = {"cg_rows_kernel<W=8>": [", 8, 1, "], "cg_rows_kernel<W=4>": [", 4, 1, "], "cg_rows_kernel<W=2>": [", 2, 1, "],
            "cg_rows_kernel<W=1>": [", 1, 4, "], "cg_rows_tiny_kernel": ["cg_rows_tiny_kernel"],
            "vh_pass": ["vh_pass_kernel", "vh_update_kernel"], "gram_wave": ["gram_wave_kernel", "gram_cg_kernel"]}
    want = next(v for k, v in tags.items() if dom["kernel"].startswith(k))
    tot, found = 0.0, False
    for name, ent in ks.items():
        head = name.split("(")[0]
        if "<double" not in head:                     # the counter passes also see the single-precision scale point
            continue
        if any(t in head for t in want) and ("cg_rows_kernel<" in head) == dom["kernel"].startswith("cg_rows_kernel"):
            r = ent.get("hbm_read_bytes_" + dom["step"]); w = ent.get("hbm_write_bytes_" + dom["step"])
            if r is not None and w is not None:
                tot += r + w; found = True
    return round(tot) if found else None


def cpu_baseline(row, col, val, m, n, A0, compare=None):
    """Times the CPU path beside the GPU number in a child process (a BLAS thread-pool failure must
    not take the GPU result down): the real reference (oracle/_ref, kind 'reference') if it
    travelled with the repo, else our C restatement (kind 'port').  Sample: full B+A half-steps of
    the same workload at a few OpenMP thread counts (the reference's row loop does not scale to
    hundreds of threads; the SciPy OpenBLAS it links supports at most 128 callers); the best one is
    reported with the thread count it used."""
    import subprocess
    import tempfile
    cores = os.cpu_count() or 1
    best = None
    factors = None       # the first successful worker's factors (the reference is bit-reproducible across thread counts)
    t_start = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        path = os.path.join(td, "w.npz")
        np.savez(path, row=row, col=col, val=val, A0=A0, m=m, n=n)
        tried = []
        # pinned threads (one per core, neighbours first: the row loop shares the opposing matrix through the last-level cache,
        # and unpinned teams of 32+ threads on this host ran SLOWER than 16: 6.6 s / 14 s / 26 s at 32 / 64 / 128 in round 2);
        # 8, 4, 16 threads first (round 3, pinned: 0.82 s / iteration at 8, 1.76 at 16, 3.97 at 32, 8.1 at 64), then others while
        # the time budget lasts
        for nthreads in [t for t in (8, 4, 16, 2, 32) if t <= cores] or [cores]:
            if time.time() - t_start > 60 and best is not None:
                break
            env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS=str(nthreads), OMP_PROC_BIND="close", OMP_PLACES="cores",
                       OMP_DYNAMIC="false")
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(nthreads)],
                                   env=env, capture_output=True, text=True, timeout=300)
            except subprocess.TimeoutExpired:
                continue
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                res = json.loads(lines[-1])
                tried.append((nthreads, res["s_per_iteration"]))
                if best is None or res["value"] > best["value"]:
                    best = res
                fpath = path + ".factors_%d.npz" % nthreads
                if os.path.exists(fpath):
                    if factors is None and compare is not None:
                        f = np.load(fpath)
                        factors = dict(A=f["A"], B=f["B"], iterations=int(f["iterations"]), kind=res["kind"], nthreads=nthreads)
                    os.remove(fpath)
    if best is None:
        return {"value": None, "unit": "rows/s", "cores": 0, "kind": "failed", "sample": "CPU baseline child failed"}, None
    best["threads_tried"] = tried
    best["host"] = host_cpu_info()
    parity = None
    if factors is not None:
        try:
            parity = compare(factors)
        except Exception as e:            # the baseline number must not depend on it
            parity = {"error": "%s: %s" % (type(e).__name__, e)}
    return best, parity


def host_cpu_info():
    """What the CPU baseline had to run on: logical cpus, the affinity mask of this process and the cgroup's cpu quota (a quota below
    the thread count, or an affinity mask narrower than cpu_count, explains a baseline that gets slower with more threads)."""
    info = {"cpu_count": os.cpu_count()}
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["sched_affinity"] = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup:" + f.rsplit("/", 1)[1]] = open(f).read().strip()
            break
        except Exception:
            pass
    try:
        info["loadavg"] = os.getloadavg()[0]
    except Exception:
        pass
    return info


def side_points(args):
    """BASELINE.json's other configurations (and the whole fit) on this GPU, driver-timed like the headline: each runs as a child
    process of this script (`--workload X`), so that none of them can take the headline down; one line per workload with its time
    per step, the roofline that bounds it and the fraction reached -- HBM for the CG configurations (algorithmic bytes),
    EXECUTED flops against the fp64 / fp32 peak for the Cholesky configurations (not the surveyed full-factorisation count)."""
    import subprocess
    here = os.path.abspath(__file__)
    plan = [("c1", 20, 3, []), ("c3", 10, 3, []), ("c4shard", 10, 3, []), ("c5shard", 3, 1, []), ("fit", 5, 0, [])]
    out = {}
    for w, st, wu, extra in plan:
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, here, "--no-cpu-baseline", "--workload", w, "--steps", str(st), "--warmup", str(wu)] + extra,
                               capture_output=True, text=True, timeout=420)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[w] = {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
                continue
            d = json.loads(lines[-1])
            if w == "fit":
                e = {"what": d["workload"], "seconds_best": d["best_s"], "rows_per_s_whole_fit": d["rows_per_s_whole_fit"],
                     "bound": "host + PCIe + 15 iterations", "frac": None}
            elif w in ("c1", "c4shard"):
                e = {"what": ("BASELINE configs[0] on the GPU: CMF explicit ALS-CG k=50 fp64, MovieLens10M shape, biases, scale_lam" if w == "c1"
                              else "one rank's share of BASELINE configs[3] at N = 8: 1.25 M users x 1 M-item replica, 62.5 M nnz, k=64 fp32"),
                     "ms_per_step": d["ms_per_iteration"], "bound": "hbm", "frac": d["frac_of_hbm_peak"], "alg_GB": d.get("alg_GB")}
            elif w == "c3":
                e = {"what": "BASELINE configs[2]: CMF explicit ALS-Chol k=128 fp64 + 64-dim item side info, MovieLens10M shape",
                     "ms_per_step": d["ms_per_iteration"], "bound": "fp64 peak 78.6 TFLOP/s (vector = matrix in double precision)",
                     "frac": d.get("executed_frac_of_fp64_peak_78.6"), "executed_TFLOP": d.get("executed_TFLOP"),
                     "surveyed_full_cholesky_frac": d.get("frac_of_fp64_vector_peak_78.6")}
            else:
                it = d.get("item_step", {})
                e = {"what": "an eighth of one rank's share of BASELINE configs[4]: " + d["workload"], "ms_per_step": d["ms_per_iteration"],
                     "bound": "fp32 matrix peak 157.3 TFLOP/s, item step on its executed flops (the user step runs the low-rank path)",
                     "frac": it.get("frac_of_fp32_matrix_peak_157"), "item_step_ms": it.get("ms"), "user_step_ms": d.get("user_step_ms")}
            e["halfstep_ms"] = d.get("halfstep_ms")
            e["finite"] = d.get("finite")
            e["steps"], e["warmup"], e["wall_s"] = st, wu, round(time.time() - t0, 1)
            out[w] = e
        except Exception as ex:            # the headline above must not depend on it
            out[w] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


def cpu_worker(path, nthreads):
    from oracle.bindings import Oracle, Reference, ref_available
    d = np.load(path)
    row, col, val, A0, m, n = d["row"], d["col"], d["val"], d["A0"], int(d["m"]), int(d["n"])
    O = Oracle(np.float64)
    csr, csc = O.coo_to_csr_and_csc(row, col, val, m, n)
    if ref_available(np.float64):
        eng, kind = Reference(np.float64), "reference"
    else:
        eng, kind = O, "port"
    A = A0.copy(); B = np.zeros((n, K))
    # one untimed warm-up iteration (page faults, thread start-up, and the first iteration's rows take the early exits less
    # often than later ones), then up to two timed ones
    t0 = time.perf_counter()
    eng.optimizeA_implicit(B, A, csc, LAM, nthreads=nthreads, use_cg=True, max_cg_steps=MAX_CG_STEPS)
    eng.optimizeA_implicit(A, B, csr, LAM, nthreads=nthreads, use_cg=True, max_cg_steps=MAX_CG_STEPS)
    t_warm = time.perf_counter() - t0
    iters, t_tot = 0, 0.0
    while iters < 2 and t_tot + t_warm < 14.0:
        t0 = time.perf_counter()
        eng.optimizeA_implicit(B, A, csc, LAM, nthreads=nthreads, use_cg=True, max_cg_steps=MAX_CG_STEPS)
        eng.optimizeA_implicit(A, B, csr, LAM, nthreads=nthreads, use_cg=True, max_cg_steps=MAX_CG_STEPS)
        t_tot += time.perf_counter() - t0
        iters += 1
    timed = iters
    if iters == 0:
        iters, t_tot = 1, t_warm       # a team too slow for a second iteration: its warm-up is its sample
    s_per_iter = t_tot / iters
    # the factors after `iters_run` iterations from the shared start: what the GPU run is compared with (parity_vs_reference)
    iters_run = 1 + timed
    np.savez(path + ".factors_%d.npz" % nthreads, A=A, B=B, iterations=iters_run)
    print(json.dumps({"value": round((m + n) / s_per_iter, 1), "unit": "rows/s", "cores": nthreads, "kind": kind,
                      "s_per_iteration": round(s_per_iter, 3),
                      "sample": "%d full ALS iteration(s) after one warm-up iteration (optimizeA_implicit B-step + A-step, the "
                                "reference's OpenMP row loop) of the same workload, nthreads=%d of %d host cpus pinned "
                                "(OMP_PROC_BIND=close, OMP_PLACES=cores; affinity mask of the worker: %d cpus), BLAS = SciPy OpenBLAS"
                                % (iters, nthreads, os.cpu_count() or 1, len(os.sched_getaffinity(0)))}))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], int(sys.argv[3]))
    else:
        main()
