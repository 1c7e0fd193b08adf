#!/usr/bin/env python3
"""Static schedule on mid-size hierarchies (GPU box): N random spheres of the stress scene, forced schedule 1.
Usage: python3 tools/mid_probe.py [n ...]   (RTC_AMD_LIB selects the library)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtc = importlib.import_module("ray-tracer-challenge_amd")
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
out = []
for n in [int(a) for a in sys.argv[1:]] or [300, 3000, 30000]:
    world, cam = rtc.stress_scene.build(n, 960, 540, with_teapot=False)
    scene = cam.compile(world, device=0)
    buf = torch.empty((540, 960, 3), dtype=torch.float64, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scene.render_device(buf.data_ptr(), stream.cuda_stream, schedule=1, depth=2)
    torch.cuda.synchronize(dev)
    a.record(stream)
    for _ in range(5):
        scene.render_device(buf.data_ptr(), stream.cuda_stream, schedule=1, depth=2)
    b.record(stream)
    torch.cuda.synchronize(dev)
    out.append("n=%d depth %d: %.3f ms" % (n, scene.info()["bvh_max_depth"], a.elapsed_time(b) / 5))
    scene.close()
print(os.path.basename(os.environ.get("RTC_AMD_LIB", "in-tree")), " ".join(out))
