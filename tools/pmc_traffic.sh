#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the headline frame for the in-tree library and every library given.  Usage: tools/pmc_traffic.sh [lib.so ...]
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --steps 2 --warmup 1"
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export RTC_AMD_LIB=$PWD/$lib; else unset RTC_AMD_LIB; fi
  name=$(basename "${lib:-in-tree}" .so)
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 170 rocprofv3 --pmc $c -d gpurun_out/pmct_$name/$c -o pmc -- $BENCH > gpurun_out/pmct_$name.$c.log 2>&1 || echo "$name $c failed"
  done
  python3 - gpurun_out/pmct_$name $name <<'PY'
import glob, sqlite3, sys
c = {}
for f in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, n, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1,2"):
        if "k_render<false" in k or "k_renderILb0" in k:
            c[n] = v
print("%-12s FETCH_SIZE %.1f GB   WRITE_SIZE %.1f GB per frame" % (sys.argv[2], c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
done
