#!/bin/bash
# Every profile the round commits, on the GPU box: per workload the five PMC passes + their timing pass and a single-stream
# kernel timeline; for the headline workload also the bench line under rocprofv3 --kernel-trace --stats.
# Results: gpurun_out/prof_<tag><workload>/ (tools/collect_profiles.sh copies the summaries into profiles/).
set -u
TAG="${1:-r03}"
bash tools/profile_round.sh trace c4 ${TAG}c4 || exit 1
for w in c1 c2 c3 c4 c5; do
  bash tools/profile_round.sh pmc $w ${TAG}$w || exit 1
  if [ $w != c4 ]; then
    OUT=gpurun_out/prof_${TAG}$w
    V4R_PIPES=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/${w}_single -- python3 bench.py --workload $w --steps 1 --warmup 1 --spp 64 --no-side --no-cpu-baseline > $OUT/${w}_single.log 2>&1 || exit 1
    python3 profiles/summarize_trace.py $(ls $OUT/${w}_single/*/*kernel_trace.csv | head -1) > $OUT/${w}_single_pipe_timeline.txt || exit 1
  fi
done
echo all profiles done
