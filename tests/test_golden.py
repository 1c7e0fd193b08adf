"""Committed golden canvases (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle).

CPU side: the oracle still produces them (it can not drift silently).  GPU side: the HIP path through the C ABI
matches them to the parity bar of BASELINE.json (every 8-bit channel equal within 1/255; the f64 delta is bounded by
1e-9 here, observed ~1e-15).
"""
import glob
import importlib.util
import os

import numpy as np
import pytest

import scene_feed as sf

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)

CASES = {name: (kind, args, depth) for name, kind, args, depth in mg.CASES}


def _golden(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    assert int(z["depth"]) == CASES[name][2]
    return z["canvas"]


def test_every_case_has_its_file_and_nothing_else_is_there():
    have = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    assert have == sorted(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    kind, args, depth = CASES[name]
    world, cam = mg.build(kind, args)
    canvas = sf.to_oracle(world, cam).render(depth=depth)
    want = _golden(name)
    assert canvas.shape == want.shape
    # same libm on every box of this image: the bits agree; the bound leaves room for a different host pow()
    assert float(np.max(np.abs(canvas - want))) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden(name):
    kind, args, depth = CASES[name]
    world, cam = mg.build(kind, args)
    scene = cam.compile(world, device=0)
    try:
        gpu = scene.render(depth=depth)
    finally:
        scene.close()
    rep = sf.compare(gpu, _golden(name))
    print(f"[golden] {name}: {rep}")
    assert rep["nan"] == 0 and rep["max_8bit"] <= 1 and rep["mismatched_channels"] == 0, rep
    assert rep["max_abs"] < 1e-9, rep
