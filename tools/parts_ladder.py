#!/usr/bin/env python3
"""Heavy tiles in 1 / 2 / 4 parts over a ladder of resolutions (static schedule, GPU box): calibrates the launcher's choice.
Usage: python3 tools/parts_ladder.py [scene ...]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtc = importlib.import_module("ray-tracer-challenge_amd")
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
for name in sys.argv[1:] or ["cover", "reflective_world"]:
    for w, h in ((320, 180), (480, 270), (640, 360), (960, 540), (1280, 720), (1600, 900), (1920, 1080), (2560, 1440), (3200, 1800), (3840, 2160)):
        sc = rtc.parse_scene_file(os.path.join(ROOT, "scenes", name + ".yaml"), w, h)
        scene = sc.camera.compile(rtc.World(sc.objects, sc.lights), device=0)
        buf = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
        row = []
        for parts in ("1", "2", "4", "0"):
            os.environ["RTC_TILE_SPLIT"] = parts
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            scene.render_device(buf.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            a.record(stream)
            for _ in range(10):
                scene.render_device(buf.data_ptr(), stream.cuda_stream)
            b.record(stream)
            torch.cuda.synchronize(dev)
            row.append(a.elapsed_time(b) / 10)
        print("%-16s %5dx%-5d heavy tiles %6d of %6d: whole %7.3f ms  halves %7.3f  quarters %7.3f  auto %7.3f" % (
            name, w, h, scene.info()["n_heavy_tiles"], ((w + 7) // 8) * ((h + 7) // 8), row[0], row[1], row[2], row[3]), flush=True)
        scene.close()
