#!/bin/bash
# per-closure profile dump of the train step for two builds of the library + the new one without conv_wres
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=$ROOT/monocon-pytorch_amd/hipmonocon/libmonocon_hip.so
mkdir -p $ROOT/gpurun_out/libdump
cp $L /tmp/lib_orig.so
run() { # tag lib env
  cp $ROOT/scratch/ab/lib_$2.so $L
  env $3 MONOCON_HIP_PROFILE_DUMP=1 python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes 2> $ROOT/gpurun_out/libdump/$1.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'conv', r['conv_ms'], 'wgrad', r['wgrad_ms'], 'other', r['other_ms'], 'fwd', d['forward_only']['ms'])"
  grep "^prof " $ROOT/gpurun_out/libdump/$1.err > $ROOT/gpurun_out/libdump/$1.txt; rm $ROOT/gpurun_out/libdump/$1.err
}
run old old X=1
run new new X=1
run new_nowres new MONOCON_HIP_WRES=0
run old2 old X=1
run new2 new X=1
cp /tmp/lib_orig.so $L
