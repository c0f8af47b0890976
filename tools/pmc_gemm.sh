#!/bin/bash
# PMC attribution of one kernel (default: the tall GEMM through tools/gemm_tall_debug.py): several small counter groups,
# one rocprofv3 pass each, summarised per kernel name.  usage: tools/pmc_gemm.sh <outdir> <kernel substring> <python script>
out=$1; kern=$2; script=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -- python3 $script > /dev/null 2> $out/g$i.err || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
