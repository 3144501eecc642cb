#!/bin/bash
# Copy the summaries of tools/profile_all.sh (and tools/bench_all.sh) from gpurun_out/ into profiles/ under the round's names.
set -u
TAG="${1:-r03}"
for w in c1 c2 c3 c4 c5; do
  D=gpurun_out/prof_${TAG}$w
  cp $D/${w}_pmc.json profiles/${TAG}_${w}_pmc.json
  cp $D/${w}_single_pipe_timeline.txt profiles/${TAG}_${w}_single_pipe_timeline.txt
done
D=gpurun_out/prof_${TAG}c4
cp $D/c4_kernel_stats.csv profiles/${TAG}_c4_kernel_stats.csv
cp $D/c4_trace_summary.txt profiles/${TAG}_c4_trace_summary.txt
cp $D/c4_bench_under_rocprof.json profiles/${TAG}_c4_bench_under_rocprof.json
for n in c1 c2 c3 c4 c5 c1_128spp c4_4968tris; do
  [ -f gpurun_out/bench_all/$n.json ] && tail -1 gpurun_out/bench_all/$n.json > profiles/${TAG}_bench_$n.json
done
ls -la profiles/${TAG}_*
