#!/bin/bash
# Prints "kernel VGPRs scratch-bytes waves/SIMD LDS-bytes" per kernel of the device library; extra arguments go to hipcc (-D...).
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -I../include -Icsrc/device -ffp-contract=off "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/v4r_dev_usage.o csrc/device/v4r_device.hip 2>&1 \
  | grep -E "Function Name|  VGPRs:|ScratchSize|Occupancy|LDS Size" \
  | sed -E 's/.*remark: +//; s/ *\[-Rpass.*//' | paste - - - - - \
  | sed -E 's/Function Name: _ZN12_GLOBAL__N_1[0-9]+//; s/N3v4r8SceneDev[^ \t]*//; s/ENS_6Queues[^ \t]*//; s/VGPRs: //; s/ScratchSize \[bytes\/lane\]: //; s/Occupancy \[waves\/SIMD\]: //; s/LDS Size \[bytes\/block\]: //' \
  | grep -v "ILb1E" | sort
