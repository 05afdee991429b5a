export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_gpu_tests.log 2>&1; echo tests rc $?; tail -3 gpurun_out/r02c_gpu_tests.log
for W in "alibaba:--workload alibaba" "c4:--concurrency 4 --n-in 20000" "nodejs:--workload nodejs --n-in 20000" "c8:--concurrency 8 --n-in 5000"; do
  tag=${W%%:*}; args=${W#*:}
  mkdir -p gpurun_out/r02c_$tag
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c_$tag -o p -- python bench.py $args --cpu-sample 0 --steps 2 --warmup 1 > gpurun_out/r02c_$tag.log 2>&1
  echo $tag rc $?
  rm -f gpurun_out/r02c_$tag/p_kernel_trace.csv gpurun_out/r02c_$tag/p_agent_info.csv
  head -12 gpurun_out/r02c_$tag/p_kernel_stats.csv | cut -c1-150
done
