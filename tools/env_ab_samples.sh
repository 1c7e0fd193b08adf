#!/bin/bash
# A/B of run-time knobs on the sample scenes (GPU box): tools/env_ab_samples.sh "VAR=a VAR=b ..." ; one line per workload and setting
SETTINGS=$1; shift
for kv in $SETTINGS; do
  echo "== $kv"
  env $kv timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-pipelined --steps 10 --warmup 2 --workload teapot_4k --parity-pixels 2000 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$kv FAILED"; tail -5 gpurun_out/ab.err; exit 1; }
  python3 tools/summarize_bench.py gpurun_out/ab.json
done
