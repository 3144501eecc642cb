#!/bin/bash
# Careful A/B of environment settings: tools/ab_steps.sh <workload> <steps> <repeats> "VAR=a" "VAR=b" ...   ("-" = defaults)
# Settings alternate within each repeat; prints Mpaths/s per run.
W="$1"; STEPS="$2"; REP="$3"; shift 3
for k in $(seq 1 $REP); do
  for setting in "$@"; do
    if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
    env $envs timeout -k 10 300 python bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline --no-side --no-profile 2>/dev/null | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$setting]', '$W', d['value'])" || exit 1
  done
done
