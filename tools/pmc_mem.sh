#!/bin/bash
# Memory-side counter passes for one command on the GPU box.  Usage: tools/pmc_mem.sh <outdir> <kernel substring> -- <program> [args]
OUT=$1; KSUB=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p $OUT
i=0
for pass in "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
            "TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_REQ_sum TCC_EA0_RDREQ_LEVEL_sum" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 170 rocprofv3 --pmc $pass -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed: $pass"
done
python3 - "$OUT" "$KSUB" <<'PY'
import sqlite3, glob, sys
for d in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    c = sqlite3.connect(d)
    for r in c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by 1,2"):
        if sys.argv[2] in r[0]:
            print("%-40s %20.0f  (n=%d)" % (r[1], r[2], r[3]))
PY
