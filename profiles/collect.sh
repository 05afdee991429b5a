#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r01
# 1. kernel trace + stats of the exact bench.py command  -> gpurun_out/<tag>_stats/
# 2. HBM read traffic  (FETCH_SIZE, own pass)            -> gpurun_out/<tag>_fetch/
# 3. HBM write traffic (WRITE_SIZE, own pass)            -> gpurun_out/<tag>_write/
# Counters are collected in separate passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
# profiles/summarize.py turns the CSVs into the committed summary (profiles/<tag>_*.{csv,json,md}).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
ARGS="--steps 5 --warmup 1 --cpu-sample 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o ${TAG} -- python bench.py $ARGS > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_fetch -o ${TAG} -- python bench.py $ARGS > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${TAG}_write -o ${TAG} -- python bench.py $ARGS > gpurun_out/${TAG}_write.log 2>&1
tail -1 gpurun_out/${TAG}_stats.log | cut -c1-400
