"""What each rank of an N-GPU run renders, timed one after the other on ONE GPU (interleaved 8-row tiles).
Usage: tools/rank_probe.py N"""
import importlib, sys
sys.path.insert(0, '.')
rtc = importlib.import_module("ray-tracer-challenge_amd")
n = int(sys.argv[1])
world, cam = rtc.stress_scene.build(100000, 3840, 2160)
sc = cam.compile(world, device=0)
sc.render(tile_rows=8, tile_index=0, tile_count=n)
ms = []
for r in range(n):
    sc.render(tile_rows=8, tile_index=r, tile_count=n)
    ms.append(sc.last_stats["kernel_ms"])
sc.render()
full = sc.last_stats["kernel_ms"]
print("N=%d per-rank kernel ms: %s" % (n, " ".join("%.0f" % m for m in ms)))
print("  max %.1f ms, whole frame on one GPU %.1f ms -> strong-scaling efficiency of the render %.2f" % (max(ms), full, full / (n * max(ms))))
