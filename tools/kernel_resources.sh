#!/bin/bash
# Prints the register / LDS / spill summary of every kernel of one csrc/*.hip translation unit (device code only).
#   tools/kernel_resources.sh net.hip [filter-regex on the demangled name]
src=${1:-net.hip}; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I include $AO_EXTRA_FLAGS \
  --cuda-device-only -c alpha_omok_amd/csrc/$src -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep "remark:" | sed -e 's/^.*remark: *//' -e 's/ \[-Rpass-analysis.*$//' |
  awk '/Function Name:/{name=$3; next} /VGPRs:|AGPRs:|SGPRs:|Spill|LDS Size|Occupancy/{printf "%s | %s\n", name, $0}' |
  c++filt | grep -E "$filt"
