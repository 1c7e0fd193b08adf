#!/bin/bash
# The round's profile suite on the GPU box: tools/profile.sh once per workload, each summarised into profiles/.
# Usage (through gpurun): tools/profile_suite.sh <round tag> [workload ...]
set -u
R=${1:-r2}; shift || true
WLS=${*:-stress_4k teapot_4k cover_1080p cover_4k reflective_world_1080p}
for wl in $WLS; do
  echo "#### $wl"
  bash tools/profile.sh ${R}_$wl --workload $wl --steps 2 --warmup 1 > gpurun_out/prof_${R}_$wl.out 2>&1 || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_${R}_$wl.out; exit 1; }
  python3 tools/summarize_prof.py gpurun_out/prof_${R}_$wl gpurun_out/${R}_$wl.txt || exit 1
  tail -3 gpurun_out/${R}_$wl.txt
done
