#!/bin/bash
cd /root/repo/monocon-pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ${1:-wgrad_mfma.hip} -o /tmp/wregs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy" | paste - - - - | grep -o "error.*\|_kernelI[A-Za-z0-9]*\|VGPRs: [0-9]*\|lane\]: [0-9]*\|SIMD\]: [0-9]*" | paste - - - -
