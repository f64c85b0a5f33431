#!/bin/bash
# phase-removal timings of the fp16-split weight-gradient kernel + SQ counters of the full variant
cd $GRAFT_REPO_ROOT/scratch/wg16
O=$GRAFT_REPO_ROOT/gpurun_out/wg16
mkdir -p $O
for v in "" _nomfma _nostore _nofetch; do
  echo "---- bench_wg$v" >> $O/time.txt
  timeout 200 ./bench_wg$v >> $O/time.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $O/pmc -o p -- $GRAFT_REPO_ROOT/scratch/wg16/bench_wg 3 > $O/pmc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d $O/pmc2 -o p -- $GRAFT_REPO_ROOT/scratch/wg16/bench_wg 3 > $O/pmc2.log 2>&1
cat $O/time.txt
