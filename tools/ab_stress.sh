#!/bin/bash
# A/B on the GPU box: bench.py on the headline workload for the in-tree library and every ab_variants/*.so
for lib in "" ab_variants/*.so; do
  if [ -n "$lib" ]; then export RTC_AMD_LIB=$PWD/$lib; else unset RTC_AMD_LIB; fi
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours --steps 2 --warmup 1 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "${lib:-in-tree} FAILED"; tail -5 gpurun_out/ab.err; exit 1; }
  echo "== ${lib:-in-tree}"; python3 tools/summarize_bench.py gpurun_out/ab.json
done
