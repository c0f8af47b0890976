#!/usr/bin/env python3
"""Per-(kernel, grid size) launch statistics from a `rocprofv3 --kernel-trace` CSV.

    python tools/kernel_trace_summary.py <dir with *_kernel_trace.csv> [kernel-name filter] > profiles/rNN_..._trace_summary.json

bench.py runs the forward SpMM on the CSR and the backward on the CSC through the SAME kernel template, so the
`--stats` CSV averages both; the two differ in grid size (the CSR of a zipf graph has long-row workgroups, the CSC of
uniform tails has none), so grouping the raw trace by grid size separates them.  For every group: launches, mean /
min / max duration, and (for the SpMM, given --bytes) the fraction of the 8 TB/s HBM roofline the mean stands for."""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    nbytes = float(sys.argv[3]) if len(sys.argv) > 3 else None
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    groups = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if flt and flt not in name:
                continue
            short = name.replace("void ", "").replace("(anonymous namespace)::", "")
            short = short.split("(")[0].strip() or name[:60]
            groups[(short, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for (name, grid, wg), ns in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        rec = {"kernel": name, "grid_threads": grid, "workgroups": grid // wg, "launches": len(ns),
               "mean_ms": sum(ns) / len(ns) / 1e6, "min_ms": min(ns) / 1e6, "max_ms": max(ns) / 1e6,
               "total_ms": sum(ns) / 1e6}
        if nbytes:
            rec["frac_of_8TBs_at_mean"] = nbytes / (rec["mean_ms"] * 1e-3) / 8e12
        rows.append(rec)
    print(json.dumps({"source": files, "filter": flt, "algorithmic_bytes_per_launch": nbytes, "groups": rows}, indent=1))


if __name__ == "__main__":
    main()
