for setting in "-" "V4R_GIVE_UP_LANES=0" "V4R_GIVE_UP_LANES=16" "V4R_GIVE_UP_LANES=24"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout -k 10 300 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline --no-side --spp 128 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); w=d['roofline']['work']; print('[$setting]', d['value'], d['roofline']['hbm']['trace']['ms_per_64_passes'], {k:w[k] for k in ('nodes_per_segment','prims_per_segment','lane_utilisation_node_step','lane_utilisation_prim_test','wave_node_steps_per_wave_ray','wave_prim_tests_per_wave_ray')})" || exit 1
done
