#!/bin/bash
# Profiling passes for bench.py on the GPU box (run through gpurun).  Usage: tools/profile.sh <tag> [bench args]
# Writes under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours $*"
echo "== kernel trace + stats" | tee $OUT/log.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH >> $OUT/log.txt 2>&1 || exit 1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass" | tee -a $OUT/log.txt
  timeout -s KILL 170 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- $BENCH >> $OUT/log.txt 2>&1 || echo "pass failed: $pass" | tee -a $OUT/log.txt
done
find $OUT -name '*.csv' | head -50
