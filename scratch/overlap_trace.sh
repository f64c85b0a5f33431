#!/bin/bash
# kernel trace of the B=32 f16x2 train loop + overlap analysis between the main and the weight-gradient stream
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/overlap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MONOCON_HIP_TUNE_CACHE=/tmp/monocon_tune_cache_ov.txt
PREC=f16x2 python $ROOT/scratch/train_prof.py > $OUT/plain.log 2>&1
rm -rf $OUT/trace
PREC=f16x2 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/scratch/train_prof.py > $OUT/trace.log 2>&1
python $ROOT/scratch/overlap_report.py $(find $OUT/trace -name t_kernel_trace.csv | head -1) | tee $OUT/report.txt
python $ROOT/scratch/stream_exclusive.py $(find $OUT/trace -name t_kernel_trace.csv | head -1) | tee -a $OUT/report.txt
