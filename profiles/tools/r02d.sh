#!/bin/bash
# One-box record of the r02d build: GPU test tier, default workload at 4 and 16 replicas + nodejs shape (torch-free
# sweep), then -- only if the tests pass -- kernel stats and the two HBM-traffic passes of the default bench command.
set -u
mkdir -p gpurun_out
timeout 150 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_gpu_tests.log 2>&1
rc=$?
echo "gpu tests rc $rc"; tail -3 gpurun_out/r02d_gpu_tests.log
rm -f gpurun_out/r02d_sweep.jsonl
timeout 60 python profiles/tools/batch_sweep.py --tag r02d --replicas 4,16 --steps 3 --out gpurun_out/r02d_sweep.jsonl | cut -c1-420
timeout 60 python profiles/tools/batch_sweep.py --tag r02d --workload nodejs --concurrency 4 --n-in 20000 --replicas 4 --steps 2 --out gpurun_out/r02d_sweep.jsonl | cut -c1-420
if [ $rc -eq 0 ]; then
    cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
    PASSES="fetch write" PASS_TIMEOUT=60 bash profiles/collect.sh r02d
fi
