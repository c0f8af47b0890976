#!/bin/bash
# Residual-only sweep of tests/test_gpu_fuzz.py with the arbiter's exits tallied: every drawn configuration forced to
# use_residual = true, once with the fp32 weight fold of rounds 2-3 (LKG_FOLD_F32=1: "before") and once with the float64 fold.
#   tools/residual_sweep.sh <cases> <first seed> <out dir>
set -u
CASES=${1:-250}; SEED=${2:-40000}; OUT=${3:-gpurun_out/r04}
mkdir -p "$OUT"
for mode in before after; do
    rm -f "$OUT/residual_exits_$mode.json"
    if [ "$mode" = before ]; then export LKG_FOLD_F32=1; else unset LKG_FOLD_F32; fi
    LKG_FUZZ_CASES=$CASES LKG_FUZZ_SEED=$SEED LKG_FUZZ_OVERRIDE='{"residual": true}' \
        LKG_FUZZ_EXITS="$OUT/residual_exits_$mode.json" \
        timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k drawn_configuration -p no:cacheprovider \
        > "$OUT/residual_sweep_$mode.log" 2>&1
    echo "$mode rc=$? $(tail -1 "$OUT/residual_sweep_$mode.log")"
    cat "$OUT/residual_exits_$mode.json"
done
