#!/bin/sh
# VGPRs / spills / LDS / occupancy of every kernel of one translation unit:  tools/regs.sh prmf_amd/csrc/residual.hip [-DFOO=1 ...]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include "$@" -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /VGPRs Spill/ {sp=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {printf "%-110s vgpr %4s spill %4s occ %s lds %s\n", name, v, sp, oc, $(NF-1)}' |
  sed 's/\[-Rpass-analysis=kernel-resource-usage\]//'
