#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into the per-launch HBM traffic of
the forward SpMM.  Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) read -> x2;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Separate --pmc passes (TCC slots).
   python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [gpurun_out/pmc_tcc] > profiles/rNN_pmc_traffic.json"""
import csv, glob, json, sys, collections

def per_dispatch(d, kernel="spmm_csr_kernel"):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    return {k: sorted(v) for k, v in acc.items()}

fetch = per_dispatch(sys.argv[1])["FETCH_SIZE"]
write = per_dispatch(sys.argv[2])["WRITE_SIZE"]
# bench.py alternates forward / backward launches of the same kernel: even positions are forwards
fwd_f = [v for i, (_, v, _) in enumerate(fetch) if i % 2 == 0]
fwd_w = [v for i, (_, v, _) in enumerate(write) if i % 2 == 0]
bwd_f = [v for i, (_, v, _) in enumerate(fetch) if i % 2 == 1]
bwd_w = [v for i, (_, v, _) in enumerate(write) if i % 2 == 1]
mean = lambda x: sum(x) / len(x)
out = {
    "kernel": "spmm_csr_kernel<float4,64,1,4,true>", "launches_sampled": len(fwd_f),
    "FETCH_SIZE_KiB_fwd": mean(fwd_f), "WRITE_SIZE_KiB_fwd": mean(fwd_w),
    "FETCH_SIZE_KiB_bwd": mean(bwd_f), "WRITE_SIZE_KiB_bwd": mean(bwd_w),
    "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
    "traffic_bytes_fwd": (2 * mean(fwd_f) + mean(fwd_w)) * 1024,
    "traffic_bytes_bwd": (2 * mean(bwd_f) + mean(bwd_w)) * 1024,
}
if len(sys.argv) > 3:
    t = per_dispatch(sys.argv[3])
    hit, miss = mean([v for _, v, _ in t["TCC_HIT_sum"]]), mean([v for _, v, _ in t["TCC_MISS_sum"]])
    out["l2_hit_rate"] = hit / (hit + miss)
print(json.dumps(out, indent=1))
