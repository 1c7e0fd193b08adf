#!/usr/bin/env python3
"""Prints the frame times of one bench.py JSON line (headline + --extra workloads)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %s: %.3f ms/step kernel %.3f ms  %.0f Mrays/s  %.2f TFLOP/s (frac %.4f)" % (
    d["config"]["workload"].split(":")[0], d["ms_per_step"], r["kernel_ms"], d["value"], r["achieved"], r["frac"]))
for k, v in d.get("extra_workloads", {}).items():
    print("  %-24s %.3f ms  %.0f Mrays/s  %.2f TFLOP/s" % (k, v["frame_ms"], v["Mrays_per_s"], v["tflops"]))
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"])
