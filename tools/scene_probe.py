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
sub = [st.get("primary_rays", 0), st.get("container_passes", 0), st.get("shadow_rays", 0), st.get("secondary_rays", 0), st.get("box_tests", 0)]
if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so") and sum(sub):
    print("  state machine sub-blocks (share of shade): closest %.3f containers %.3f lights %.3f shade_hit tail %.3f next ray %.3f" % tuple(x / float(sum(sub)) for x in sub))
if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so"):
    pt = st.get("prim_tests")
    print("  light block: look-ups %.3f lighting %.3f rest %.3f (of the light block)" % (pt[0] / float(sub[2] + pt[0] + pt[1]), pt[1] / float(sub[2] + pt[0] + pt[1]), sub[2] / float(sub[2] + pt[0] + pt[1])))
for nm, r, l in (("start", d[4], d[5]), ("traverse", d[6], d[7]), ("shade", d[8], d[9])):
    print("  %-9s rounds %d  lanes/round %.1f" % (nm, r, l / max(r, 1)))
scene.render(); print("plain kernel_ms", scene.last_stats["kernel_ms"])
