#!/bin/bash
# Sweep one environment knob of the kernel over values (on the GPU box).  Usage: tools/env_sweep.sh VAR "v1 v2 .." [bench args]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-pipelined --steps 2 --warmup 1 "$@" > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { echo "$VAR=$v FAILED"; tail -5 gpurun_out/sweep.err; exit 1; }
  echo "== $VAR=$v"; python3 tools/summarize_bench.py gpurun_out/sweep.json
done
