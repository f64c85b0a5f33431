#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box.
#   usage (from the repo root, under gpurun):  bash profiles/collect.sh <tag> [bench args...]
# Writes gpurun_out/<tag>/{trace,pmc_sq,pmc_fetch,pmc_write}/...  (copy summaries to profiles/).
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache.txt
BENCH="timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --forward-steps 2 --no-cpu-baseline --no-extra-modes $*"
$BENCH > "$OUT/bench_plain.log" 2>&1   # warms the tune cache so the traces hold no autotuning launches
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -o p -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_lds" -o p -- $BENCH > "$OUT/pmc_lds.log" 2>&1
find "$OUT" -name "*.csv" | head -20
du -sh "$OUT"
