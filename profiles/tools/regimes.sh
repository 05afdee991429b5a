# bench lines of the load regimes (+ rocprofv3 kernel stats where asked): bash profiles/tools/regimes.sh <tag> [stats]
export TMPDIR=/tmp
TAG=${1:-r02}
for W in "alibaba:--workload alibaba" "c4:--concurrency 4 --n-in 20000 --replicas 4" "nodejs:--workload nodejs --n-in 20000" "c8:--concurrency 8 --n-in 5000 --replicas 4"; do
  tag=${W%%:*}; args=${W#*:}
  if [ "${2:-}" = "stats" ]; then
    mkdir -p gpurun_out/${TAG}_$tag
    timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_$tag -o p -- python bench.py $args --cpu-sample 0 --regimes 0 --sync engine --steps 2 --warmup 1 > gpurun_out/${TAG}_$tag.log 2>&1
    echo $tag rc $?
    rm -f gpurun_out/${TAG}_$tag/p_kernel_trace.csv gpurun_out/${TAG}_$tag/p_agent_info.csv
    head -8 gpurun_out/${TAG}_$tag/p_kernel_stats.csv | cut -c1-150
    tail -1 gpurun_out/${TAG}_$tag.log | cut -c1-200
  else
    timeout 150 python bench.py $args --cpu-sample 0 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_$tag.json 2> gpurun_out/${TAG}_bench_$tag.err
    echo $tag rc $?; python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/${TAG}_bench_$tag.json').read().strip().split('\n')[-1])
    print('$tag', '%.3g'%r['value'], round(r['ms_per_step'],2), r['accuracy'], 'budget',r['budget_windows'], 'repaired',r['repaired_windows'], {k:round(v,2) for k,v in r['roofline']['group_ms_per_launch'].items()})
except Exception as e: print('$tag', 'no result', e)
PY
  fi
done
