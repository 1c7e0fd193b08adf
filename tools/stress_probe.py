import importlib, sys, time
sys.path.insert(0, '.')
import os
rtc = importlib.import_module("ray-tracer-challenge_amd")
n, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = time.time(); world, cam = rtc.stress_scene.build(n, w, h); print("build %.1fs" % (time.time() - t), flush=True)
t = time.time(); sc = cam.compile(world, device=0); print("commit %.1fs" % (time.time() - t), sc.info(), flush=True)
for mode in sys.argv[4:]:
    kw = {}
    if mode.startswith("tile"):
        kw = dict(tile_rows=8, tile_index=int(mode[4:]), tile_count=(h + 7) // 8)
        for ub in (1, 2):
            t = time.time(); img = sc.render(use_bvh=ub, collect_stats=True, **kw)
            print(mode, "use_bvh", ub, "%.2fs" % (time.time() - t), "kernel_ms", sc.last_stats["kernel_ms"], "sum", float(abs(img).sum()), {k: sc.last_stats[k] for k in ("primary_rays", "shadow_rays", "secondary_rays", "container_passes", "overflow", "box_tests")}, flush=True)
    else:
        t = time.time(); img = sc.render(collect_stats=("stats" in sys.argv))
        print("full", "%.2fs" % (time.time() - t), "kernel_ms", sc.last_stats["kernel_ms"], {k: sc.last_stats.get(k) for k in ("primary_rays", "shadow_rays", "secondary_rays", "container_passes", "overflow", "box_tests", "prim_tests")}, flush=True)
        d = sc.last_stats.get("diag")
        if d and sum(d):
            tot = float(sum(d[:4])) or 1.0
            print("  stage cycles (share): refill %.3f start %.3f traverse %.3f shade %.3f" % tuple(x / tot for x in d[:4]))
            st = sc.last_stats
            sub = [st.get("primary_rays", 0), st.get("container_passes", 0), st.get("shadow_rays", 0), st.get("secondary_rays", 0), st.get("box_tests", 0)]
            if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so") and sum(sub):
                print("  state machine sub-blocks (share of shade): closest %.3f containers %.3f lights %.3f shade_hit tail %.3f next ray %.3f" % tuple(x / float(sum(sub)) for x in sub))
            pt = st.get("prim_tests")
            if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so") and pt[2]:
                print("  step halves: node half ran %d times with %.1f lanes, leaf half %d times with %.1f lanes" % (pt[2], pt[3] / max(pt[2], 1), pt[4], pt[5] / max(pt[4], 1)))
            if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so") and st.get("shaded_hits"):
                span = st["shaded_hits"] - ((~st["light_evals"]) & 0xFFFFFFFFFFFFFFFF)
                print("  wave lifetimes (100 MHz ticks): kernel span %d, mean wave lifetime %.0f = %.3f of the span" % (span, pt[6] / 3072.0, pt[6] / 3072.0 / max(span, 1)))
            if os.environ.get("RTC_AMD_LIB", "").endswith("timing.so"):
                for nm, a, b in (("closest block", "xform_levels", "schlick_evals"), ("containers block", "refract_evals", "shadow_rays_skipped"), ("lighting block", "secondary_rays_reference", "overflow")):
                    print("  %-16s ran %d times with %.1f lanes" % (nm, st.get(b, 0), st.get(a, 0) / max(st.get(b, 0), 1)))
            for name, r, l in (("start", d[4], d[5]), ("traverse", d[6], d[7]), ("shade", d[8], d[9])):
                print("  %-9s rounds %d  lanes/round %.1f" % (name, r, l / max(r, 1)))
