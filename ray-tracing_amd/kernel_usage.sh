#!/bin/bash
# Prints VGPR / scratch / occupancy / LDS per kernel of the device library.
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -I../include -Icsrc/device -ffp-contract=off \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/v4r_dev_usage.o csrc/device/v4r_device.hip 2>&1 \
  | grep -E "Function Name|  VGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*remark: [^ ]+ +//; s/\[-Rpass.*//' | paste - - - - - \
  | sed -E 's/Function Name: _ZN12_GLOBAL__N_1//; s/EvN3v4r8SceneDev[^ \t]*//; s/EN3v4r8SceneDev[^ \t]*//'
