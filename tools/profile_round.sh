#!/bin/bash
# Profiles of one workload on the GPU box.  Everything goes to gpurun_out/prof_<tag>/; the summaries are copied
# into profiles/ by hand.
#   tools/profile_round.sh trace <workload> <tag>   kernel trace + stats of the bench line (three pipelines) and a
#                                                   single-stream timeline (V4R_PIPES=1: exclusive kernel times)
#   tools/profile_round.sh pmc <workload> <tag>     the five PMC passes (one counter set per run) over a 64-spp
#                                                   single-stream step: HBM bytes, cache hit rates, issue statistics
set -u
MODE="$1"; W="$2"; TAG="$3"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
case "$MODE" in
trace)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -- python3 bench.py --workload $W --no-cpu-baseline --no-side > $OUT/${W}_bench_under_rocprof.json 2> $OUT/${W}_stats.err || exit 1
  python3 profiles/summarize_trace.py $(ls $OUT/${W}_stats/*/*kernel_trace.csv | head -1) > $OUT/${W}_trace_summary.txt || exit 1
  cp $(ls $OUT/${W}_stats/*/*kernel_stats.csv | head -1) $OUT/${W}_kernel_stats.csv || exit 1
  V4R_PIPES=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/${W}_single -- python3 bench.py --workload $W --steps 1 --warmup 1 --spp 64 --no-side --no-cpu-baseline > $OUT/${W}_single.log 2>&1 || exit 1
  python3 profiles/summarize_trace.py $(ls $OUT/${W}_single/*/*kernel_trace.csv | head -1) > $OUT/${W}_single_pipe_timeline.txt || exit 1
  ;;
pmc)
  run() {  # name, counters
    V4R_PIPES=1 timeout -s KILL 240 rocprofv3 --pmc $2 --output-format csv -d $OUT/pmc2_$1 -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-side --spp 64 > $OUT/pmc2_$1.log 2>&1 || exit 1
    cp $OUT/pmc2_$1/*/*_counter_collection.csv $OUT/pmc2_$1/ || exit 1
  }
  run fetch "FETCH_SIZE" && run write "WRITE_SIZE" && run l2 "TCC_HIT_sum TCC_MISS_sum" &&
  run sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" &&
  run ta "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_WAIT_ANY SQ_BUSY_CYCLES" || exit 1
  # the exclusive kernel times of the same command (no counters): the denominator of the per-kernel rates
  V4R_PIPES=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc2_time -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-side --spp 64 > $OUT/pmc2_time.log 2>&1 || exit 1
  cp $(ls $OUT/pmc2_time/*/*kernel_trace.csv | head -1) $OUT/pmc2_time/kernel_trace.csv || exit 1
  python3 profiles/summarize_pmc_passes.py $OUT $W > $OUT/${W}_pmc.json || exit 1
  ;;
esac
echo done $MODE $W
