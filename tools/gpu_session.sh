#!/bin/bash
# One gpurun call: GPU tests (not -x: all failures at once), then bench + profiles unless the tests were killed.
# usage: tools/gpu_session.sh <tag> [pytest args...]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=12 "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${tag}_tests.log
tail -n 30 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?"; cat gpurun_out/${tag}_bench.json | head -c 3000
