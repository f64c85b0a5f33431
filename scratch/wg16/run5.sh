#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/wg16
W=$GRAFT_REPO_ROOT/scratch/wg16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
: > $O/t5.txt
run() { # tag, env..., binary
  tag=$1; shift
  rm -rf $O/p5_$tag
  env "$@" 3 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/p5_$tag -o p -- env "$@" 3 > $O/p5_$tag.log 2>&1
  echo "---- $tag" >> $O/t5.txt
  python $W/pmc_summary.py $O/p5_$tag/p_counter_collection.csv >> $O/t5.txt 2>&1
}
run old $W/bench_wg
run mfmaonly $W/bench_wg_nostore_nofetch
run fetchonly $W/bench_wg_fetchonly
run pipe WG_PIPE=1 $W/bench_wg
cat $O/t5.txt
