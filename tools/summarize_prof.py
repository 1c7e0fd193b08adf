#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory (rocprofv3 rocpd sqlite DBs) into a text report.
Usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>.txt"""
import glob
import json
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
lines = []
tr = glob.glob(os.path.join(src, "trace", "*.db"))
if tr:
    c = sqlite3.connect(tr[0])
    lines.append("== rocprofv3 --kernel-trace --stats (top_kernels; durations in ns -> us)")
    lines.append("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
        lines.append("%-90s %8d %14.3f %12.3f %7.2f" % (name[:90], calls, total, avg, pct))
counters = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
    c = sqlite3.connect(d)
    for k, n, v, cnt in c.execute(
            "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by 1, 2"):
        if "k_render" in k:
            counters.setdefault(k.split("(")[0], {})[n] = (v, cnt)
lines.append("")
lines.append("== rocprofv3 --pmc (separate passes), average per dispatch")
for k, cs in counters.items():
    lines.append(k)
    for n, (v, cnt) in sorted(cs.items()):
        lines.append("    %-32s %20.1f   (dispatches %d)" % (n, v, cnt))
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"][0], cs["WRITE_SIZE"][0]
        lines.append("    -> HBM traffic per launch: FETCH_SIZE %.0f KiB (x2 gfx950 correction for 16 B/lane streams = "
                     "%.1f MB upper bound), WRITE_SIZE %.0f KiB = %.1f MB" % (f, 2 * f * 1024 / 1e6, w, w * 1024 / 1e6))
    if "GRBM_GUI_ACTIVE" in cs:
        lines.append("    -> GRBM_GUI_ACTIVE/8 = %.0f cycles per launch" % (cs["GRBM_GUI_ACTIVE"][0] / 8))
log = os.path.join(src, "log.txt")
if os.path.exists(log):
    for ln in open(log):
        if ln.startswith('{"metric"'):
            lines.append("")
            lines.append("== bench line of the traced run")
            lines.append(ln.strip())
            break
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
