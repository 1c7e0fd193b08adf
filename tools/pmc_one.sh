#!/bin/bash
# One counter pass of the headline bench.  Usage: tools/pmc_one.sh "<counters>" [path/to/lib.so] [bench args]
CNT=$1; LIB=${2:-}; shift 2 || shift $#
[ -n "$LIB" ] && export RTC_AMD_LIB=$PWD/$LIB
OUT=gpurun_out/pmc1_$$
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 150 rocprofv3 --pmc $CNT -d $OUT -o pmc -- python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours --steps 2 --warmup 1 "$@" > $OUT/log.txt 2>&1 || { echo "pass failed"; tail -3 $OUT/log.txt; }
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
for d in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    db = sqlite3.connect(d)
    for k, n, v, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by 1,2"):
        if "k_render<false" in k or "k_renderILb0" in k:
            print("    %-28s %18.0f  (n=%d)" % (n, v, cnt))
PY
rm -rf $OUT
