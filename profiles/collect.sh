#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r02a [bench args...]
# 1. kernel trace + stats of the exact bench.py command   -> gpurun_out/<tag>_stats/
# 2. HBM read traffic  (FETCH_SIZE, own pass)             -> gpurun_out/<tag>_fetch/
# 3. HBM write traffic (WRITE_SIZE, own pass)             -> gpurun_out/<tag>_write/
# 4. SQ passes (occupancy, VALU utilisation, wait states) -> gpurun_out/<tag>_sq1/
# 5. SQ passes (LDS bank conflicts, instruction mix)      -> gpurun_out/<tag>_sq2/
# Counters are collected in separate passes with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots:
# 8 SQ slots, FETCH_SIZE and WRITE_SIZE do not fit one pass).  Every command runs under its own timeout.
# profiles/summarize.py turns the CSVs into the committed summary (profiles/<tag>_*.{csv,json,md}).
set -u
TAG=${1:-r02}
shift || true
# --sync engine: bench.py without torch (N = 1; every C-ABI call returns synchronised) -- a fresh box spends a minute or
# two on its first `import torch`, once per pass
ARGS="--steps 5 --warmup 1 --cpu-sample 0 --regimes 0 --sync engine $*"
PASSES=${PASSES:-"fetch write sq1 sq2 grbm"}
export TMPDIR=/tmp
T="timeout ${PASS_TIMEOUT:-240}"
mkdir -p gpurun_out
python -c "from traceweaver_amd import build; print(build.source_digest())" > gpurun_out/${TAG}_digest.txt 2>/dev/null || true
cp .git_head gpurun_out/${TAG}_commit.txt 2>/dev/null || true
run_pmc() {  # name, counters...
    local name=$1; shift
    mkdir -p gpurun_out/${TAG}_${name}
    $T rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${TAG}_${name} -o ${TAG} -- python bench.py $ARGS > gpurun_out/${TAG}_${name}.log 2>&1
    echo "${name}: rc $?"
}
if [ -z "${SKIP_STATS:-}" ]; then   # (a later call that only adds counter passes to a set keeps the first call's stats)
mkdir -p gpurun_out/${TAG}_stats
$T rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -o ${TAG} -- python bench.py $ARGS > gpurun_out/${TAG}_stats.log 2>&1
echo "stats: rc $?"
fi
for p in $PASSES; do
    case $p in
    fetch) run_pmc fetch FETCH_SIZE;;
    write) run_pmc write WRITE_SIZE;;
    sq1) run_pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU;;
    sq2) run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA;;
    grbm) run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT;;
    esac
done
# only the summaries travel back (the per-dispatch CSVs of five passes stay under the 64 MiB pull limit)
find gpurun_out -name "*_agent_info.csv" -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_stats.log | cut -c1-600
