#!/usr/bin/env python3
"""Turns a `rocprofv3 --pmc FETCH_SIZE` pass over tools/microbench/fetch_calib into the correction factors that
tools/pmc_summary.py applies (bytes that really moved / bytes the counter reports), per access pattern:

    python tools/fetch_calibration.py <rocprof output dir> <stdout of fetch_calib> -o profiles/r02_fetch_calibration.json
"""
import argparse, csv, glob, json, os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("known"); ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    known = json.loads([l for l in open(a.known) if l.startswith("{")][-1])
    vals = {"calib_stream_kernel": [], "calib_gather_kernel": [], "calib_gather128_kernel": []}
    for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
            per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
        for (_, name), v in per.items():
            for k in vals:
                if name.strip().endswith(k) or (k + "<") in name or name.strip() == k:
                    vals[k].append(v * 1024.0)                      # FETCH_SIZE is in KiB
    out = {"known": known, "note": "factor = known bytes / FETCH_SIZE bytes; the last launch of three is used (warm TLBs, cold data: "
                                   "both buffers exceed the 256 MiB Infinity Cache)"}
    if vals["calib_stream_kernel"]:
        c = vals["calib_stream_kernel"][-1]
        out["stream_16B_per_lane"] = {"fetch_size_bytes": c, "factor": known["stream_bytes"] / c}
    if vals["calib_gather_kernel"]:
        c = vals["calib_gather_kernel"][-1]
        out["gather_8B_per_lane_64B_segments"] = {"fetch_size_bytes": c, "factor_vs_logical": known["gather_logical_bytes"] / c,
                                                  "factor_vs_64B_sectors": known["gather_sector64_bytes"] / c,
                                                  "factor_vs_128B_lines": known["gather_line128_bytes"] / c,
                                                  "all_launches_bytes": vals["calib_gather_kernel"]}
    if vals["calib_gather128_kernel"]:
        c = vals["calib_gather128_kernel"][-1]
        out["gather_8B_per_lane_128B_segments"] = {"fetch_size_bytes": c, "factor_vs_logical": known["gather_logical_bytes"] / c,
                                                   "factor_vs_64B_sectors": known["gather_sector64_bytes"] / c,
                                                   "factor_vs_128B_lines": known["gather_line128_bytes"] / c,
                                                   "all_launches_bytes": vals["calib_gather128_kernel"]}
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
