#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py into the per-launch HBM traffic of
the forward SpMM.  Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) read -> x2;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Separate --pmc passes (TCC slots).
   python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [gpurun_out/pmc_tcc] > profiles/rNN_pmc_traffic.json"""
import csv, glob, hashlib, json, os, sys, collections

def per_dispatch(d, kernel="spmm_csr_kernel"):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    return {k: sorted(v) for k, v in acc.items()}

# one lkg_spmm_csr_f32 call = SLABS kernel launches (1 since the equal 128-column slabs share one launch; 2 at D=256
# for the r01_v5 / v6 library); bench.py alternates forward and backward calls: [fwd x SLABS, bwd x SLABS, ...]
SLABS = int(os.environ.get("LKG_SLABS", "1"))


def per_call(rows):
    full = max(g for _, _, g in rows)                 # bench.py's spot check launches a few one-row grids: skip them
    vals = [v for _, v, g in rows if g > full // 2]
    calls = [sum(vals[i:i + SLABS]) for i in range(0, len(vals) - SLABS + 1, SLABS)]
    return calls[0::2], calls[1::2]


fetch = per_dispatch(sys.argv[1])["FETCH_SIZE"]
write = per_dispatch(sys.argv[2])["WRITE_SIZE"]
fwd_f, bwd_f = per_call(fetch)
fwd_w, bwd_w = per_call(write)
mean = lambda x: sum(x) / len(x)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {
    # bench.py reports `traffic` only while lkg_spmm.hip still has this hash (the counters describe THAT kernel)
    "spmm_source_sha16": hashlib.sha256(open(os.path.join(ROOT, "literalkg_amd", "csrc", "lkg_spmm.hip"), "rb").read()).hexdigest()[:16],
    "kernel": f"spmm_csr_kernel<float4,32,1,4,true>, {SLABS} launch(es) per call (128-column slabs, slab-major)", "calls_sampled": len(fwd_f),
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
