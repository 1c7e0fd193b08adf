#!/usr/bin/env python3
"""Renders scenes and dumps the canvases (A/B bit-comparison of two library builds: RTC_AMD_LIB=...)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
rtc = importlib.import_module("ray-tracer-challenge_amd")
out = sys.argv[1]
imgs = {}
for name, w, h in (("teapot", 960, 540), ("cover", 640, 640), ("reflective_world", 800, 400), ("ice_cream", 400, 200)):
    sc = rtc.parse_scene_file(os.path.join("scenes", name + ".yaml"), w, h)
    imgs[name] = sc.camera.render(rtc.World(sc.objects, sc.lights)).pixels
world, cam = rtc.stress_scene.build(3000, 320, 180)
imgs["stress"] = cam.render(world).pixels
np.savez(out, **imgs)
