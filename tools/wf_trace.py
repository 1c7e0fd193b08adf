import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rtc = importlib.import_module("ray-tracer-challenge_amd")
w, h = int(sys.argv[1]), int(sys.argv[2])
world, cam = rtc.stress_scene.build(100000, w, h)
scene = cam.compile(world, device=0)
ref = scene.render(schedule=0); t_ref = scene.last_stats["kernel_ms"]
t = time.perf_counter(); got = scene.render(schedule=3); t1 = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); got = scene.render(schedule=3); t2 = (time.perf_counter() - t) * 1e3
print("max|delta| %.3e nan %d megakernel %.1f ms wavefront call %.1f / %.1f ms" % (float(np.nanmax(np.abs(got - ref))), int(np.isnan(got).sum()), t_ref, t1, t2))
