#!/bin/bash
# PMC passes over the C4-regime SpMM (5 M entities / 100 M entries / D = 256 on ONE GPU: nothing fits a cache -- what a rank of the
# row-range scheme runs per 1/N of the rows): fabric traffic, L2 hit rate, how long waves wait, fabric credit stalls.  Separate
# passes (TCC slots), --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
#   usage: tools/pmc_c4.sh <tag>     (outputs under gpurun_out/<tag>/; summary.json is what profiles/ keeps)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}
mkdir -p $out
run="python3 tools/spmm_micro.py --n 5000000 --e 100000000 --device-graph --iters 3 --only-main"
timeout -k 10 200 $run > $out/timing.log 2>&1 || exit 1
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pmc_$i -- $run > /dev/null 2> $out/pmc_$i.err || echo "pass $i ($pass) failed" >> $out/failed.txt
done
python3 tools/pmc_c4_summary.py $out > $out/summary.json
cat $out/summary.json
