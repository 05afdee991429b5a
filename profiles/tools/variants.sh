#!/bin/bash
# Times every build under scratch/variants/ on the default workload (4 and 16 replicas) and, for the selection
# variants, on the nodejs shape; rows appended to gpurun_out/variants.jsonl.  Run through gpurun from the repo root.
mkdir -p gpurun_out
OUT=gpurun_out/${1:-variants}.jsonl
shift
for so in "$@"; do
    tag=$(basename $so .so)
    timeout 60 python profiles/tools/batch_sweep.py --lib $so --tag $tag --replicas 4,16 --steps 3 --out $OUT > /dev/null 2>gpurun_out/err_$tag.txt || echo "$tag failed"
    case $tag in base|memo*) timeout 60 python profiles/tools/batch_sweep.py --lib $so --tag $tag --workload nodejs --concurrency 4 --n-in 20000 --replicas 4 --steps 2 --out $OUT > /dev/null 2>>gpurun_out/err_$tag.txt || echo "$tag nodejs failed";; esac
done
cut -c1-330 $OUT
