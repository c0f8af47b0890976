#!/bin/bash
# One gpurun call: the rocprofv3 evidence of bench.py at N = 1 (kernel stats of the timed launches only + the three
# separate PMC passes of /opt/skills/guides/MI355X_MICROARCH.md's HBM section), summarised into profiles/-shaped files.
#   usage: tools/profile_session.sh <tag>        (outputs under gpurun_out/<tag>_*)
set -o pipefail
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o b -- python3 bench.py --no-cpu-baseline --no-extra > $out/bench_profiled.json 2> $out/stats.err || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2> $out/pmc_fetch.err || exit 2
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2> $out/pmc_write.err || exit 3
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/pmc_tcc -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2> $out/pmc_tcc.err || exit 4
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_tcc > $out/pmc_traffic.json || exit 5
python3 tools/kernel_trace_summary.py $out/stats spmm_csr_kernel 11345828676 > $out/spmm_trace_summary.json || exit 6
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err
cat $out/pmc_traffic.json; head -40 $out/spmm_trace_summary.json; head -c 1500 $out/bench_profiled.json
