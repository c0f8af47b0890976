#!/bin/bash
# HBM traffic of the fused gate launch (gemm_tall_kernel<256, 1>): separate FETCH_SIZE / WRITE_SIZE passes, corrected as
# MI355X_MICROARCH.md prescribes, written as a profiles/-shaped JSON.   usage: tools/pmc_gate.sh <outdir>
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 tools/gate_only.py > /dev/null 2> $out/fetch.err || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 tools/gate_only.py > /dev/null 2> $out/write.err || exit 2
# request sizes at the fabric side of the L2: FETCH_SIZE tallies EVERY read request at 64 B, the guide's x 2 is exact only when all of
# them are 128-B requests; the gate's 16-k windows of a 1 KB row are 64-B halves of a line
timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum --kernel-trace --output-format csv -d $out/rdreq -- python3 tools/gate_only.py > /dev/null 2> $out/rdreq.err || echo "request-size pass failed" > $out/rdreq.failed
python3 - <<PY
import csv, glob, hashlib, json
def vals(d, name):
    f = glob.glob(f"$out/{d}/**/*counter_collection.csv", recursive=True)[0]
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_tall_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
fe, wr = vals("fetch", "FETCH_SIZE"), vals("write", "WRITE_SIZE")
fe, wr = fe[1:], wr[1:]                       # first launch: cold
n, d = 1_000_000, 256
alg = 4.0 * n * (d + 302 + d)
mean = lambda x: sum(x) / len(x)
rec = {"kernel": "gemm_tall_kernel<256, gate epilogue>: GateMul forward 1000000 x (256+2+300) -> 256, eval (no g/z kept)",
       "tall_source_sha16": hashlib.sha256(open("literalkg_amd/csrc/lkg_gemm_tall.hip", "rb").read()).hexdigest()[:16],
       "launches_sampled": len(fe), "FETCH_SIZE_KiB": mean(fe), "WRITE_SIZE_KiB": mean(wr),
       "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
       "traffic_bytes": (2 * mean(fe) + mean(wr)) * 1024, "algorithmic_bytes": alg}
rec["traffic_over_algorithmic"] = rec["traffic_bytes"] / alg
try:
    tot, r32, r64 = (vals("rdreq", k)[1:] for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum"))
    t_, a_, b_ = mean(tot), mean(r32), mean(r64)
    rec["read_requests"] = {"all": t_, "32B": a_, "64B": b_, "128B (the rest)": t_ - a_ - b_}
    rec["read_bytes_by_request_size"] = 32 * a_ + 64 * b_ + 128 * (t_ - a_ - b_)
    rec["traffic_bytes_by_request_size"] = rec["read_bytes_by_request_size"] + mean(wr) * 1024
    rec["traffic_over_algorithmic_by_request_size"] = rec["traffic_bytes_by_request_size"] / alg
except Exception as exc:
    rec["read_requests"] = f"not collected ({type(exc).__name__}: {exc})"
print(json.dumps(rec, indent=1))
PY
