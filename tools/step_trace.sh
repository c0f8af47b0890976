#!/bin/bash
# Kernel statistics of tools/step_profile.py for one configuration.   usage: tools/step_trace.sh <config> <outdir>
cfg=$1; out=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 tools/step_profile.py --config $cfg --iters ${ITERS:-10} $EXTRA > $out/step.log 2> $out/step.err || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(open("$out/step.log").read().strip().splitlines()[-1])
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
