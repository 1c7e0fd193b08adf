"""Stage clocks of ONE horizontal band of the stress frame (timing build).  Usage: RTC_AMD_LIB=ab_variants/timing.so tools/band_stage_probe.py bands index"""
import importlib, sys
sys.path.insert(0, '.')
rtc = importlib.import_module("ray-tracer-challenge_amd")
bands, idx = int(sys.argv[1]), int(sys.argv[2])
w, h = 3840, 2160
world, cam = rtc.stress_scene.build(100000, w, h)
sc = cam.compile(world, device=0)
kw = dict(tile_rows=h // bands, tile_index=idx, tile_count=bands)
sc.render(**kw)
sc.render(**kw)
print("plain kernel_ms %.1f" % sc.last_stats["kernel_ms"])
sc.render(collect_stats=True, **kw)
st = sc.last_stats
d, pt = st["diag"], st["prim_tests"]
tot = float(sum(d[:4])) or 1.0
print("timing kernel_ms %.1f" % st["kernel_ms"])
print("  stage cycles (share): refill %.3f start %.3f traverse %.3f shade %.3f" % tuple(x / tot for x in d[:4]))
for nm, r, l in (("start", d[4], d[5]), ("traverse", d[6], d[7]), ("shade", d[8], d[9])):
    print("  %-9s rounds %d  lanes/round %.1f" % (nm, r, l / max(r, 1)))
print("  step halves: node half ran %d times with %.1f lanes, leaf half %d times with %.1f lanes" % (pt[2], pt[3] / max(pt[2], 1), pt[4], pt[5] / max(pt[4], 1)))
span = st["shaded_hits"] - ((~st["light_evals"]) & 0xFFFFFFFFFFFFFFFF)
print("  wave lifetimes (100 MHz ticks): kernel span %d, summed lifetimes %d = %.3f of span x 3072" % (span, pt[6], pt[6] / 3072.0 / max(span, 1)))
