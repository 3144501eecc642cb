#!/bin/bash
# A/B of environment knobs on the GPU box: tools/ab_env.sh "<workloads>" <spp> "VAR=a VAR2=b" "VAR=c" ...
# (each quoted argument is one setting; "-" = defaults).  Prints Mpaths/s, ms per step and the exclusive k_trace and k_shade times of 64 passes.
set -u
WL="$1"; SPP="$2"; shift 2
for setting in "$@"; do
  for w in $WL; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-side --spp $SPP 2>&1 | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$setting]', '$w', d['value'], d['ms_per_step'], d['roofline']['groups']['trace']['ms_per_64_passes'], d['roofline']['groups']['shade']['ms_per_64_passes'])" || exit 1
  done
done
