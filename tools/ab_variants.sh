#!/bin/bash
# A/B of device-library builds on the GPU box: tools/ab_variants.sh "<workloads>" <spp> <variant.so>...
# ("base" = the in-tree library).  Prints Mpaths/s, ms per step and the exclusive k_trace and k_shade times of 64 passes.
set -u
WL="$1"; SPP="$2"; shift 2
for v in "$@"; do
  for w in $WL; do
    if [ "$v" = base ]; then unset V4R_DEVICE_LIB; else export V4R_DEVICE_LIB="$PWD/ray-tracing_amd/lib/variants/$v.so"; fi
    timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --spp $SPP 2>&1 | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', '$w', d['value'], d['ms_per_step'], d['roofline']['groups']['trace']['ms_per_64_passes'], d['roofline']['groups']['shade']['ms_per_64_passes'])" || exit 1
  done
done
