#!/usr/bin/env python3
"""A/B harness: runs bench.py --extra for every library in ab_variants/ (plus the in-tree one) and prints frame
times per workload.  Usage (on the GPU box): python tools/ab.py [steps]"""
import glob
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "10"
libs = [("in-tree", "")] + [(os.path.basename(p), p) for p in sorted(glob.glob(os.path.join(root, "ab_variants", "*.so")))]
rows = {}
for name, path in libs:
    env = dict(os.environ)
    if path:
        env["RTC_AMD_LIB"] = path
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", steps, "--warmup", "2", "--extra",
                        "--no-cpu-baseline", "--no-pipelined", "--no-flavours", "--workload", "teapot_4k"], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        print(name, "FAILED", r.stderr[-500:])
        continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    row = {"teapot_4k": d["roofline"]["kernel_ms"]}
    for k, v in d.get("extra_workloads", {}).items():
        row[k] = v["frame_ms"]
    rows[name] = row
    print(name, " ".join("%s=%.3f" % kv for kv in row.items()), flush=True)
