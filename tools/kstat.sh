#!/bin/bash
# Kernel statistics of one python script.   usage: tools/kstat.sh <outdir> <script> [args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 "$@" > $out/run.log 2> $out/run.err || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(open("$out/run.log").read().strip().splitlines()[-1])
for r in rows[:12]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  {r["Name"][:100]}')
PY
