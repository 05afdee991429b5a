#!/bin/bash
# On the GPU box (through gpurun, from the repo root; ~4 minutes): for every build of build_stack.py -- the GPU test tier with that
# library in place of the product's (in the box's copy of the tree only), then its timings on the default workload at 6.4 M and
# 25.6 M spans and on the nodejs shape (torch-free sweep).  Rows: gpurun_out/stack.jsonl, test tails: gpurun_out/stack_tests_<name>.log
set -u
mkdir -p gpurun_out
cp traceweaver_amd/lib/libtwgpu.so /tmp/libtwgpu_head.so
rm -f gpurun_out/stack.jsonl
for so in scratch/variants/s[0-9]_*.so; do
    tag=$(basename $so .so)
    cp $so traceweaver_amd/lib/libtwgpu.so
    timeout 150 python -m pytest tests -m gpu -x -q > gpurun_out/stack_tests_$tag.log 2>&1
    echo "$tag: gpu tests rc $? $(tail -1 gpurun_out/stack_tests_$tag.log)"
    timeout 60 python profiles/tools/batch_sweep.py --lib $so --tag $tag --replicas 4,16 --steps 3 --out gpurun_out/stack.jsonl > /dev/null 2> gpurun_out/stack_err_$tag.txt || echo "$tag media failed"
    timeout 60 python profiles/tools/batch_sweep.py --lib $so --tag $tag --workload nodejs --concurrency 4 --n-in 20000 --replicas 4 --steps 2 --out gpurun_out/stack.jsonl > /dev/null 2>> gpurun_out/stack_err_$tag.txt || echo "$tag nodejs failed"
done
cp /tmp/libtwgpu_head.so traceweaver_amd/lib/libtwgpu.so
python - <<'PY'
import json
for l in open("gpurun_out/stack.jsonl"):
    r = json.loads(l)
    print("%-22s %-7s x%-2d %7.2f ms/step  enumerate %6.2f  select %6.2f  fit %5.2f  %.3g spans/s  acc %.4f  unproven %d" % (
        r["tag"], r["workload"], r["replicas"], r["ms_per_step"], r["enumerate_ms"], r["select_ms"], r["fit_ms"], r["spans_per_s"], r["accuracy"], r["budget_windows"]))
PY
