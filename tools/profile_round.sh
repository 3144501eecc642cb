#!/bin/bash
# Profiles of one round on the GPU box: kernel trace + stats of the default bench line, single-stream timelines,
# and the five PMC passes (one counter set per run) that give HBM traffic per kernel.  Everything goes to
# gpurun_out/prof_round/; the summaries are copied into profiles/ by hand.
#   tools/profile_round.sh trace|pmc
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_round
mkdir -p $OUT
case "$1" in
trace)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -- python3 bench.py --no-bunny --no-cpu-baseline > $OUT/c2_bench_under_rocprof.json 2> $OUT/c2_stats.err || exit 1
  python3 profiles/summarize_trace.py $(ls $OUT/c2_stats/*/*kernel_trace.csv | head -1) > $OUT/c2_trace_summary.txt || exit 1
  for w in c1 c2 c4; do
    V4R_PIPES=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/${w}_single -- python3 bench.py --workload $w --steps 1 --warmup 1 --spp 64 --no-bunny --no-cpu-baseline > $OUT/${w}_single.log 2>&1 || exit 1
    python3 profiles/summarize_trace.py $(ls $OUT/${w}_single/*/*kernel_trace.csv | head -1) > $OUT/${w}_single_pipe_timeline.txt || exit 1
  done
  ;;
pmc)
  run() {  # name, counters
    timeout -s KILL 240 rocprofv3 --pmc $2 --output-format csv -d $OUT/pmc2_$1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-bunny --spp 64 > $OUT/pmc2_$1.log 2>&1 || exit 1
    cp $OUT/pmc2_$1/*/*_counter_collection.csv $OUT/pmc2_$1/ || exit 1
  }
  run fetch "FETCH_SIZE" && run write "WRITE_SIZE" && run l2 "TCC_HIT_sum TCC_MISS_sum" &&
  run sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" &&
  run ta "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_ANY SQ_BUSY_CYCLES" || exit 1
  python3 profiles/summarize_pmc_passes.py $OUT 132710400 > $OUT/c2_pmc_traffic.json || exit 1
  ;;
esac
echo done $1
