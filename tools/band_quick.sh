#!/bin/bash
# full-frame bench line + three bands (proxy for one rank of an 8-GPU run)
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --steps 2 --warmup 1 > gpurun_out/bq.json 2> gpurun_out/bq.err || exit 1
python3 tools/summarize_bench.py gpurun_out/bq.json
timeout -k 10 300 python3 tools/band_probe.py 27 2>&1 | tail -9
