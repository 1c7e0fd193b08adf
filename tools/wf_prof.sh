#!/bin/bash
# Per-kernel times of one wavefront frame of the stress scene (rocprofv3 kernel trace).  Usage: tools/wf_prof.sh W H [tag]
W=${1:-1920}; H=${2:-1080}; TAG=${3:-wf}
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof_$TAG
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o wf -- python3 tools/wf_trace.py $W $H > gpurun_out/prof_$TAG.log 2>&1
grep "max|delta" gpurun_out/prof_$TAG.log
python3 - "$TAG" <<'PY'
import sqlite3, glob, sys
db = glob.glob("gpurun_out/prof_%s/**/*.db" % sys.argv[1], recursive=True)[0]
c = sqlite3.connect(db)
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    if "rocclr" in name: continue
    print("  %-64s %6d launches %10.1f ms in all" % (name[:64], calls, total / 1e3))
PY
