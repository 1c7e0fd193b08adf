"""Stage shares and lane participation of one sample scene (counting kernel).  Usage: tools/scene_probe.py name w h"""
import importlib, os, sys
sys.path.insert(0, '.')
rtc = importlib.import_module("ray-tracer-challenge_amd")
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sc = rtc.parse_scene_file(os.path.join("scenes", name + ".yaml"), w, h)
world = rtc.World(sc.objects, sc.lights)
scene = sc.camera.compile(world, device=0)
print(scene.info())
for k in range(2):
    scene.render(collect_stats=True)
st = scene.last_stats
print("kernel_ms", st["kernel_ms"], {k: st.get(k) for k in ("primary_rays", "shadow_rays", "secondary_rays", "container_passes", "box_tests", "prim_tests")})
d = st.get("diag")
tot = float(sum(d[:4])) or 1.0
print("  stage cycles (share): refill %.3f start %.3f traverse %.3f shade %.3f" % tuple(x / tot for x in d[:4]))
for nm, r, l in (("start", d[4], d[5]), ("traverse", d[6], d[7]), ("shade", d[8], d[9])):
    print("  %-9s rounds %d  lanes/round %.1f" % (nm, r, l / max(r, 1)))
scene.render(); print("plain kernel_ms", scene.last_stats["kernel_ms"])
