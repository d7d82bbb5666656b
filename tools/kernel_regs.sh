#!/bin/bash
# Usage: tools/kernel_regs.sh <file.hip> [name-filter]   -- VGPR/AGPR/occupancy/spills per kernel (compile only)
F=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$(dirname "$0")/../sylph-few-shot-detection_amd/csrc" -c "$F" -o /tmp/kernel_regs.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|Occupancy|VGPRs Spill|LDS Size" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass-analysis=kernel-resource-usage\]//' \
  | paste - - - - - - | sed 's/Function Name: //; s/_ZN5sylph//; s/EEEvNS_8ConvArgsE//' | grep -E "$PAT" | tr -s ' \t' ' '
