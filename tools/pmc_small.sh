#!/bin/bash
# Counter record for the small (static-schedule) scenes: frame time and one SQ counter pass per library and workload.
# Usage: tools/pmc_small.sh <tag> "<workload ...>" [lib.so ...]   ("" = in-tree)
TAG=$1; WLS=$2; shift 2
OUT=gpurun_out/pmcs_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export RTC_AMD_LIB=$PWD/$lib; else unset RTC_AMD_LIB; fi
  name=$(basename "${lib:-in-tree}" .so)
  for wl in $WLS; do
    BENCH="python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --steps 20 --warmup 3 --workload $wl"
    timeout -k 10 120 $BENCH > $OUT/$name.$wl.json 2> $OUT/$name.$wl.err || { echo "$name $wl bench failed"; tail -3 $OUT/$name.$wl.err; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/$name.$wl.pmc -o pmc -- $BENCH > $OUT/$name.$wl.log 2>&1 || echo "$name $wl pmc pass failed"
    python3 - $OUT/$name.$wl.json $OUT/$name.$wl.pmc "$name" $wl <<'PY'
import glob, json, sqlite3, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = {}
for f in glob.glob(sys.argv[2] + "/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, n, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1,2"):
        if "k_render<false" in k or "k_renderILb0" in k:
            c[n] = v
g = lambda n: c.get(n, float("nan"))
print("%-10s %-24s %8.3f ms  SMEM %.3gM  LDS %.3gM  VMEM_RD %.3gM  VALU %.3gM  waiting %.3f  issue-wait %.3f  active %.3f" % (
    sys.argv[3], sys.argv[4], d["roofline"]["kernel_ms"], g("SQ_INSTS_SMEM") / 1e6, g("SQ_INSTS_LDS") / 1e6, g("SQ_INSTS_VMEM_RD") / 1e6,
    g("SQ_INSTS_VALU") / 1e6, g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
PY
  done
done
