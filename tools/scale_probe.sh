#!/bin/bash
# Frame time of the stress scene at three resolutions (same scene, same rays per pixel on average): what does not scale
# with the pixel count is the frame's tail (the dependent chains of its heaviest pixels).
for wl in stress_1080p stress_4k stress_8k; do
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours --steps 2 --warmup 1 --workload $wl "$@" > gpurun_out/scale.json 2> gpurun_out/scale.err || { echo "$wl FAILED"; tail -3 gpurun_out/scale.err; exit 1; }
  python3 tools/summarize_bench.py gpurun_out/scale.json
done
