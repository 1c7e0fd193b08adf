#!/bin/bash
# Counter passes for one command on the GPU box.  Usage: tools/pmc_probe.sh <outdir> <kernel substring> -- <program> [args]
OUT=$1; KSUB=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p $OUT
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $pass -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$KSUB" <<'PY'
import sqlite3, glob, sys
for d in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    c = sqlite3.connect(d)
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by 1,2"):
        if sys.argv[2] in r[0]:
            print("%-36s %20.0f  (n=%d)" % (r[1], r[2], r[3]))
PY
