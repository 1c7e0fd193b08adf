#!/bin/bash
# Four counter passes + one plain run of the headline bench for one library.  Usage: tools/pmc_quick.sh <tag> [path/to/lib.so] [bench args]
TAG=$1; LIB=${2:-}; shift 2 || shift $#
[ -n "$LIB" ] && export RTC_AMD_LIB=$PWD/$LIB
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --no-cpu-baseline --no-pipelined --no-extra --no-flavours --steps 2 --warmup 1 $*"
timeout -k 10 120 $BENCH > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -3 $OUT/bench.err; exit 1; }
python3 tools/summarize_bench.py $OUT/bench.json
python3 - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
w, r = d["work_per_frame"], d["rays_per_frame"]
passes = r["shadow"] + r["secondary"] + r["primary"] + w["container_passes"]
print("  passes %.3fG  box tests %.1fG (%.1f per pass)  prim tests %.2fG  shadow skipped %.2fG" % (
    passes / 1e9, w["box_tests"] / 1e9, w["box_tests"] / passes, sum(w["prim_tests"]) / 1e9,
    r.get("shadow_skipped_light_behind_surface", 0) / 1e9))
PY
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT" \
            "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $pass -d $OUT/p$i -o pmc -- $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" <<'PY'
import sqlite3, glob, sys
c = {}
for d in sorted(glob.glob(sys.argv[1] + "/p*/**/*.db", recursive=True)):
    db = sqlite3.connect(d)
    for k, n, v, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by 1,2"):
        if "k_render<false" in k or "k_renderILb0" in k:
            c[n] = v
for n in sorted(c):
    print("    %-28s %18.0f" % (n, c[n]))
g = lambda n: c.get(n, float("nan"))
print("  lane utilisation   %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))))
print("  waves waiting      %.3f   issue stalls %.3f   active %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
print("  VALU busy (x4/GUI) %.3f   SALU/VALU %.3f" % (g("SQ_ACTIVE_INST_VALU") * 4 / (g("GRBM_GUI_ACTIVE") / 8 * 1024), g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU")))
print("  L2 hit rate        %.3f   VALU instr per wave-cycle-quad %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), g("SQ_INSTS_VALU") / g("SQ_WAVE_CYCLES")))
PY
