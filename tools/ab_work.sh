#!/bin/bash
# A/B of device-library builds with the work counters: tools/ab_work.sh "<workloads>" <spp> <variant>... ("base" = the in-tree library)
set -u
WL="$1"; SPP="$2"; shift 2
for v in "$@"; do
  for w in $WL; do
    if [ "$v" = base ]; then unset V4R_DEVICE_LIB; else export V4R_DEVICE_LIB="$PWD/ray-tracing_amd/lib/variants/$v.so"; fi
    timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-side --spp $SPP 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); w=d['roofline']['work']; h=d['roofline']['groups']
print('$v', '$w', d['value'], 'trace', h['trace']['ms_per_64_passes'], 'shade', h['shade']['ms_per_64_passes'], {k:w[k] for k in ('nodes_per_segment','prims_per_segment','lane_utilisation_node_step','lane_utilisation_prim_test','wave_node_steps_per_wave_ray','wave_prim_tests_per_wave_ray')})" || exit 1
  done
done
