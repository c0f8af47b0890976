#!/usr/bin/env python3
"""Summary of tools/pmc_c4.sh: per-dispatch counter means of spmm_csr_kernel (forward launches = those with the CSR's
long-row workgroups, i.e. the larger grid; transpose launches the smaller), with the guide's gfx950 corrections."""
import collections, csv, glob, json, os, re, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmm_csr_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
res = {"workload": "spmm_csr_kernel, 5 M entities / ~100 M stored entries / D = 256 (two 128-column slabs in one launch), zipf heads drawn on the "
                   "device; per-launch means; forward = the launch with the CSR's long-row workgroups (larger grid)"}
timing = open(f"{out}/timing.log").read() if os.path.exists(f"{out}/timing.log") else ""
res["timing_lines"] = [l.strip() for l in timing.splitlines() if " median " in l]
for name, by_grid in sorted(acc.items()):
    grids = sorted(by_grid)
    res[name] = {("forward" if g == max(grids) else "transpose") + f"(grid {g})": sum(v) / len(v) for g, v in by_grid.items()}
def pick(name, which):
    d = res.get(name, {})
    for k, v in d.items():
        if k.startswith(which):
            return v
    return None
for which in ("forward", "transpose"):
    f, w = pick("FETCH_SIZE", which), pick("WRITE_SIZE", which)
    if f is not None and w is not None:
        res[f"traffic_bytes_{which}"] = (2 * f + w) * 1024          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    h, m = pick("TCC_HIT_sum", which), pick("TCC_MISS_sum", which)
    if h is not None and m is not None:
        res[f"l2_hit_rate_{which}"] = h / (h + m)
    wc, wa = pick("SQ_WAVE_CYCLES", which), pick("SQ_WAIT_ANY", which)
    if wc and wa is not None:
        res[f"wave_cycles_waiting_share_{which}"] = wa / wc
m_ = re.search(r"stored_entries|", timing)
print(json.dumps(res, indent=1))
