#!/bin/bash
# usage: scratch/regs.sh [pattern]  -- register / occupancy summary of the conv kernels
cd /root/repo/monocon-pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c conv_mfma.hip -o conv_mfma.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" | paste - - - - - | grep -E "${1:-.}" | grep -o "error.*\|kernelI[A-Za-z0-9]*\|VGPRs: [0-9]*\|AGPRs: [0-9]*\|ScratchSize \[bytes/lane\]: [0-9]*\|Occupancy \[waves/SIMD\]: [0-9]*" | paste - - - - - | sed 's/ScratchSize \[bytes\/lane\]/scratch/; s/Occupancy \[waves\/SIMD\]/occ/; s/kernelI//; s/Li//g; s/EEvNS//'
