#!/usr/bin/env python3
"""profiles/hbm_traffic.json (read by bench.py -> roofline.traffic) from the round's profile summaries.
Usage: tools/hbm_from_profiles.py r2   (reads profiles/<round>_<workload>.txt written by tools/summarize_prof.py)"""
import glob
import json
import os
import re
import sys

import subprocess

rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
try:  # the tree the profiles were taken of: pass it as the second argument, else the current HEAD
    commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "?"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = {}
for path in sorted(glob.glob(os.path.join(root, "profiles", rnd + "_*.txt"))):
    wl = os.path.basename(path)[len(rnd) + 1:-4]
    kernel, best = None, None
    vals = {}
    for ln in open(path):
        m = re.match(r"^(void )?(rtc::\S*k_render<([a-z, ]+)>)", ln)
        if m and "(dispatches" not in ln and not ln.startswith(" "):
            kernel = m.group(3)
            continue
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(dispatches (\d+)\)", ln)
        if m and kernel is not None and not kernel.startswith("true"):  # (the STATS variant is the counting frame)
            vals[m.group(1)] = float(m.group(2)) * 1024.0
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out[wl] = {
            "fetch_bytes": vals["FETCH_SIZE"], "write_bytes": vals["WRITE_SIZE"],
            "bytes_per_launch": vals["FETCH_SIZE"] + vals["WRITE_SIZE"],
            "source": "profiles/" + os.path.basename(path),
            "commit": commit,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB), separate passes, per k_render launch of the timed "
                    "variant; FETCH_SIZE is NOT doubled: the guide's x2 applies to 16 B/lane streaming reads, these are "
                    "scattered 16-64 B record reads (uncalibrated width); Infinity-Cache hits are included in both",
        }
json.dump(out, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-26s fetch %.3e B  write %.3e B" % (k, v["fetch_bytes"], v["write_bytes"]))
