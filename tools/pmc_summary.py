#!/usr/bin/env python3
"""Condenses rocprofv3 counter-collection CSVs (one directory per --pmc pass, written under
gpurun_out/ on the GPU box) into a small JSON that is committed under profiles/ and read by bench.py
for the roofline "traffic" field.

    python tools/pmc_summary.py gpurun_out/pmc_r01_* -o profiles/r01_pmc_summary.json

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so reads are doubled.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha1 over the sources of the CG row kernels: bench.py compares it with the tree it runs from, so that counter
    files taken from older kernels are visible as stale in the bench line."""
    h = hashlib.sha1()
    for f in ("cmfrec_amd/csrc/cg_kernels.hpp", "cmfrec_amd/csrc/gram_cg_kernels.hpp", "cmfrec_amd/csrc/lanes.hpp"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--round", default="", help="round / step tag of the passes (e.g. r02_m)")
    ap.add_argument("--calibration", default=os.path.join(ROOT, "profiles", "fetch_calibration.json"),
                    help="output of tools/fetch_calibration.py: FETCH_SIZE correction measured on the gather's access pattern")
    args = ap.parse_args()
    # FETCH_SIZE -> bytes: the guide's x2 holds for 16 B / lane coalesced reads; the CG kernels gather 8 B / lane in 64 B
    # segments, for which the factor is measured (tools/microbench/fetch_calib.hip)
    fetch_factor, fetch_note = 2.0, "x2 (guide, 16 B / lane coalesced; uncalibrated for this pattern)"
    factor128 = None
    if os.path.exists(args.calibration):
        cal = json.load(open(args.calibration))
        g = cal.get("gather_8B_per_lane_64B_segments")
        if g:
            fetch_factor = float(g["factor_vs_64B_sectors"])
            fetch_note = "x%.3f measured on the gather pattern (known 64 B sectors / FETCH_SIZE, %s)" % (fetch_factor, os.path.basename(args.calibration))
        g = cal.get("gather_8B_per_lane_128B_segments")
        if g:       # gram_wave_kernel: 16 lanes x 8 B per gathered row and instruction
            factor128 = float(g["factor_vs_64B_sectors"])
            fetch_note += "; gram_wave_kernel x%.3f (16 lanes x 8 B segments)" % factor128
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in args.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (int(r["Dispatch_Id"]), r["Counter_Name"])
                per_dispatch[key] += float(r["Counter_Value"])
                names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
            for (disp, cname), v in sorted(per_dispatch.items()):
                agg[names[disp]][cname].append(v)
    out = {}
    for kname, counters in agg.items():
        if "cmfhip" not in kname:
            continue
        # bench.py launches B-step then A-step; the MODE=1 split-row kernels run max_cg_steps=3 times per half-step
        per_half = 3 if (("vh_pass_kernel" in kname or "vh_update_kernel" in kname) and kname.split("(")[0].rstrip().endswith(", 1>")) else 1
        ent = {}
        for c, v in counters.items():
            halves = [sum(v[i:i + per_half]) for i in range(0, len(v) - per_half + 1, per_half)]
            ent[c] = {"mean": sum(v) / len(v), "launches": len(v),
                      "per_halfstep_B": sum(halves[0::2]) / max(len(halves[0::2]), 1),
                      "per_halfstep_A": sum(halves[1::2]) / max(len(halves[1::2]), 1)}
        ff = factor128 if (factor128 is not None and "gram_wave_kernel" in kname) else fetch_factor
        for step in ("A", "B"):
            if "FETCH_SIZE" in ent:
                ent["hbm_read_bytes_%s" % step] = ent["FETCH_SIZE"]["per_halfstep_" + step] * 1024 * ff
            if "WRITE_SIZE" in ent:
                ent["hbm_write_bytes_%s" % step] = ent["WRITE_SIZE"]["per_halfstep_" + step] * 1024
        out[kname] = ent
    json.dump({"note": "counter sums per half-step (bench.py order: B-step then A-step); FETCH_SIZE/WRITE_SIZE in KiB; "
                       "hbm_read_bytes_* = FETCH_SIZE x 1024 x fetch_factor", "fetch_factor": fetch_factor, "fetch_factor_source": fetch_note,
               "round": args.round, "kernel_source_hash": kernel_source_hash(), "kernels": out},
              open(args.out, "w"), indent=1, sort_keys=True)
    print("wrote", args.out, len(out), "kernels")


if __name__ == "__main__":
    main()
