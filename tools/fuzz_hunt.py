"""Bug hunt beyond the committed seeds: random scenes of tests/fuzz_scenes.py, GPU (static whole tiles, default,
persistent, linear scan) against the oracle.  Prints one line per seed that differs.  Usage: tools/fuzz_hunt.py first last [objects per scene]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import scene_feed as sf
from scene_feed import rtc
from fuzz_scenes import random_scene

first, last = int(sys.argv[1]), int(sys.argv[2])
n_objects = int(sys.argv[3]) if len(sys.argv) > 3 else None
bad = 0
t0 = time.time()
last_print = t0
for seed in range(first, last + 1):
    world, cam, depth = random_scene(seed, n_objects=n_objects)
    ow = sf.to_oracle(world, cam)
    ref = ow.render(depth=depth)
    good = ~np.isnan(ref).any(axis=-1)
    scene = cam.compile(world, device=0)
    imgs = {}
    try:
        for name, kw in (("static", dict(schedule=1, exchange=1)), ("default", {}), ("persistent", dict(schedule=2)),
                         ("brute", dict(schedule=1, exchange=1, use_bvh=2))):
            try:
                imgs[name] = scene.render(depth=depth, **kw)
            except rtc.RtcError as e:
                imgs[name] = scene.last_canvas
                if good.all():
                    print(f"seed {seed}: {name} raised {e}", flush=True)
                    bad += 1
    finally:
        scene.close()
    rep = sf.compare(imgs["static"][good], ref[good])
    msgs = []
    if rep["nan"] or rep["max_8bit"] > 1 or rep["max_abs"] >= 1e-9:
        msgs.append(f"static vs oracle {rep}")
    if not np.array_equal(imgs["static"], imgs["brute"], equal_nan=True):
        msgs.append("hierarchy != linear scan")
    for name in ("default", "persistent"):
        # (relative to the frame's largest value: a TestPattern returns coordinates as colours, pixels reach 1e5)
        scale = max(1.0, float(np.nanmax(np.abs(imgs["static"][good])))) if good.any() else 1.0
        d = float(np.nanmax(np.abs(imgs[name][good] - imgs["static"][good]))) / scale if good.any() else 0.0
        if not d < 1e-12:
            msgs.append(f"{name} vs static {d:.3g}")
    if msgs:
        bad += 1
        print(f"seed {seed} ({int((~good).sum())} diverged pixels, depth {depth}, {len(world.lights)} lights): " + "; ".join(msgs), flush=True)
    if seed % 20 == 0 or time.time() - last_print > 30:
        last_print = time.time()
        print(f"... seed {seed}, {bad} findings, {time.time() - t0:.0f} s", flush=True)
print(f"seeds {first}..{last}: {bad} findings")
