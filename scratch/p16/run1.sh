#!/bin/bash
# prototype run: checks, timings (random and zero data), SQ counters of the timing cases
cd $GRAFT_REPO_ROOT/scratch/p16
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/p16
timeout 200 ./bench_p16 > $GRAFT_REPO_ROOT/gpurun_out/p16/rand.txt 2>&1
P16_ZERO=1 timeout 200 ./bench_p16 time > $GRAFT_REPO_ROOT/gpurun_out/p16/zero.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/p16/pmc -o p -- $GRAFT_REPO_ROOT/scratch/p16/bench_p16 time > $GRAFT_REPO_ROOT/gpurun_out/p16/pmc.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/p16/rand.txt
echo ---- zero; cat $GRAFT_REPO_ROOT/gpurun_out/p16/zero.txt
