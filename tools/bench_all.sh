#!/bin/bash
# Bench lines of every workload at its quoted size and sample count (GPU box) -> gpurun_out/bench_all/<name>.json
set -u
OUT=gpurun_out/bench_all
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || exit 1; echo $name $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'].split(':')[0], r['bound'], r['frac'], 'hbm', r['hbm_traffic_frac'], 'issue', r['issue_frac'], 'cpu', d['cpu_baseline']['value'], d.get('gpu_over_cpu'))"); }
run c4 --workload c4 &&
run c2 --workload c2 --no-side &&
run c1 --workload c1 --no-side &&
run c3 --workload c3 --no-side &&
run c5 --workload c5 --steps 2 --no-side &&
run c1_128spp --workload c1 --spp 128 --no-side &&
run c4_4968tris --workload c4 --c4-subdivide 0 --no-side
