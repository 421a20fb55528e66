#!/bin/bash
# usage: scripts/kernel_resources.sh <file.hip> [name filter] [extra hipcc flags]
# one line per kernel: demangled name | VGPRs | scratch bytes/lane | waves/SIMD | spilled VGPRs   (hipcc remarks)
f=$1; filt=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result \
  -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /dev/null 2>&1 |
  grep -E "Function Name|  VGPRs:|Occupancy|VGPRs Spill|ScratchSize" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  paste - - - - - | sed -E 's/Function Name: //; s/VGPRs: /v=/; s/ScratchSize \[bytes\/lane\]: /scratch=/; s/Occupancy \[waves\/SIMD\]: /occ=/; s/VGPRs Spill: /spill=/' |
  while IFS=$'\t' read -r name rest; do
    echo "$(echo "$name" | c++filt | sed -E 's/\(.*//; s/void nrhip:://') | $rest"
  done | grep -E "$filt"
