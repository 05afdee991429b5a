#!/bin/bash
# Everything the round's record needs from one box, each step under its own timeout:
#   bash profiles/tools/final.sh <tag>
# GPU tier, profile collection (profiles/collect.sh), the default bench line (with the host legs), the load regimes, the
# sharded modes with several gloo ranks sharing the one GPU (+ verification against the single-GPU result), the RCCL
# smoke, end to end at a larger size.
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -1 gpurun_out/${TAG}_gpu_tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/${TAG}_smoke.log
timeout 400 bash profiles/collect.sh ${TAG} > gpurun_out/${TAG}_collect.log 2>&1; echo "collect rc $?"
timeout 240 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; cut -c1-400 gpurun_out/${TAG}_bench.json
# (the load regimes are a separate call: bash profiles/tools/regimes.sh <tag>)
timeout 150 python bench.py --gpus 4 --backend gloo --workload alibaba --verify 1 --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_alibaba_4rank.json 2> gpurun_out/${TAG}_bench_alibaba_4rank.err; echo "alibaba 4 ranks rc $?"
timeout 150 python bench.py --gpus 2 --backend gloo --workload media-split --verify 1 --n-in 20000 --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_split_2rank.json 2> gpurun_out/${TAG}_bench_split_2rank.err; echo "media-split 2 ranks rc $?"
timeout 60 python profiles/tools/rccl_smoke.py > gpurun_out/${TAG}_rccl_smoke.json 2> gpurun_out/${TAG}_rccl_smoke.err; echo "rccl smoke rc $?"; tail -1 gpurun_out/${TAG}_rccl_smoke.json
timeout 150 python profiles/tools/e2e_scale.py alibaba:50000 > gpurun_out/${TAG}_e2e_scale.json 2> gpurun_out/${TAG}_e2e_scale.err; echo "e2e scale rc $?"; cut -c1-600 gpurun_out/${TAG}_e2e_scale.json
for f in alibaba_4rank split_2rank; do tail -1 gpurun_out/${TAG}_bench_$f.json | python -c "
import json,sys
try:
    r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$f', r['n_gpus'], '%.3g'%r['value'], r.get('sharded_equals_single_gpu'), r['accuracy'])
except Exception as e: print('$f no result', e)"; done
