#!/bin/bash
# A/B of run-time knobs on the GPU box: tools/env_ab.sh "VAR=a VAR=b ..." [bench args]; one headline line per setting
SETTINGS=$1; shift
for kv in $SETTINGS; do
  echo "== $kv"
  env $kv timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours --steps 2 --warmup 1 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$kv FAILED"; tail -5 gpurun_out/ab.err; exit 1; }
  python3 tools/summarize_bench.py gpurun_out/ab.json
done
