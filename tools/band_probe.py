"""Cost profile of the stress frame over image height: renders horizontal bands one at a time.  Usage: tools/band_probe.py [bands]"""
import importlib, sys
sys.path.insert(0, '.')
rtc = importlib.import_module("ray-tracer-challenge_amd")
bands = int(sys.argv[1]) if len(sys.argv) > 1 else 27
w, h = 3840, 2160
rows = h // bands
world, cam = rtc.stress_scene.build(100000, w, h)
sc = cam.compile(world, device=0)
sc.render(tile_rows=rows, tile_index=0, tile_count=bands)
tot_ms = tot_rays = 0.0
for i in range(bands):
    sc.render(tile_rows=rows, tile_index=i, tile_count=bands)
    ms = sc.last_stats["kernel_ms"]
    sc.render(tile_rows=rows, tile_index=i, tile_count=bands, collect_stats=True)
    st = sc.last_stats
    rays = st["primary_rays"] + st["shadow_rays"] + st["secondary_rays"]
    tot_ms += ms; tot_rays += rays
    print("band %2d rows %4d-%4d  %8.1f ms  %10.4g rays  %7.0f Mrays/s  rays/pixel %7.1f" % (i, i * rows, (i + 1) * rows - 1, ms, rays, rays / ms / 1e3, rays / (rows * w)), flush=True)
print("sum %.1f ms, %.4g rays" % (tot_ms, tot_rays))
