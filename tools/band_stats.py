#!/usr/bin/env python3
"""Stage clocks and rounds of the slowest 8-row band of a sample frame (counting kernel, static schedule).
Usage: python3 tools/band_stats.py cover 1920 1080 [exchange]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
exchange = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
sc = rtc.parse_scene_file(os.path.join(ROOT, "scenes", name + ".yaml"), w, h)
scene = sc.camera.compile(rtc.World(sc.objects, sc.lights), device=0)
buf = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
bands = (h + 7) // 8


def timed(**kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scene.render_device(buf.data_ptr(), stream.cuda_stream, **kw)
    torch.cuda.synchronize(dev)
    a.record(stream)
    for _ in range(5):
        scene.render_device(buf.data_ptr(), stream.cuda_stream, **kw)
    b.record(stream)
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / 5


per = [timed(schedule=1, exchange=exchange, tile_rows=8, tile_index=i, tile_count=bands) for i in range(0, bands, 4)]
i = 4 * max(range(len(per)), key=lambda k: per[k])
kw = dict(schedule=1, exchange=exchange, tile_rows=8, tile_index=i, tile_count=bands)
print("%s %dx%d exchange=%d: whole frame %.3f ms, band %d alone %.3f ms" % (name, w, h, exchange, timed(schedule=1, exchange=exchange), i, timed(**kw)))
scene.render_device(buf.data_ptr(), stream.cuda_stream, collect_stats=True, **kw)
st = scene.stats()
d = st["diag"]
tot = float(sum(d[:4])) or 1.0
waves = (w + 7) // 8
print("  stage cycles (share): refill %.3f start %.3f traverse %.3f shade %.3f; mean cycles per wave %.0f" % (tuple(x / tot for x in d[:4]) + (tot / waves,)))
for nm, r, l, cyc in (("start", d[4], d[5], d[1]), ("traverse", d[6], d[7], d[2]), ("shade", d[8], d[9], d[3])):
    print("  %-9s rounds %d (%.1f per wave) lanes/round %.1f  cycles/round %.0f" % (nm, r, r / waves, l / max(r, 1), cyc / max(r, 1)))
print("  rays: primary %d shadow %d secondary %d containers %d box tests %d prim tests %s" % (
    st["primary_rays"], st["shadow_rays"], st["secondary_rays"], st["container_passes"], st["box_tests"], st["prim_tests"]))
