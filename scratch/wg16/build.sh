#!/bin/bash
# builds the harness variants (run here, the binaries travel with the snapshot)
cd $(dirname $0)
F="-O3 --offload-arch=gfx950 -std=c++17 -I../../monocon-pytorch_amd/csrc"
hipcc $F bench_wg.hip -o bench_wg &
hipcc $F -DMC_EXP_NO_MFMA bench_wg.hip -o bench_wg_nomfma &
hipcc $F -DMC_EXP_NO_STORE bench_wg.hip -o bench_wg_nostore &
hipcc $F -DMC_EXP_NO_FETCH bench_wg.hip -o bench_wg_nofetch &
wait
ls -la bench_wg*
