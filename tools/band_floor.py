#!/usr/bin/env python3
"""Where the floor of a small frame comes from: every band of 8 rows of the frame rendered by itself (static schedule).
Usage: python3 tools/band_floor.py cover 1920 1080"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
sc = rtc.parse_scene_file(os.path.join(ROOT, "scenes", name + ".yaml"), w, h)
scene = sc.camera.compile(rtc.World(sc.objects, sc.lights), device=0)
buf = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
bands = (h + 7) // 8
times = []
for i in range(-1, bands):
    kw = dict(schedule=1) if i < 0 else dict(schedule=1, tile_rows=8, tile_index=i, tile_count=bands)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scene.render_device(buf.data_ptr(), stream.cuda_stream, **kw)
    torch.cuda.synchronize(dev)
    a.record(stream)
    for _ in range(5):
        scene.render_device(buf.data_ptr(), stream.cuda_stream, **kw)
    b.record(stream)
    torch.cuda.synchronize(dev)
    times.append(a.elapsed_time(b) / 5)
full, per = times[0], times[1:]
order = sorted(range(bands), key=lambda i: -per[i])
print("%s %dx%d: whole frame %.3f ms; slowest bands of 8 rows alone: %s; median band %.3f ms; sum of bands %.1f ms" % (
    name, w, h, full, ", ".join("rows %d-%d %.3f ms" % (8 * i, 8 * i + 7, per[i]) for i in order[:4]),
    sorted(per)[bands // 2], sum(per)))
