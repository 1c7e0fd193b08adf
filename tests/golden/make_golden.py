#!/usr/bin/env python3
"""Writes the golden canvases under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

What they are: f64 canvases rendered by the CPU oracle (oracle/rtc_oracle.c, itself pinned to the reference's
known-answer tests by tests/test_oracle_known_answers.py) for the reference's sample scenes (scenes/*.yaml, camera
resized) and for a small instance of the stress scene.  The Rust reference can not run in this image (no toolchain),
so these are NOT outputs of the reference binary; they freeze the oracle's output so that (1) the oracle can not
drift silently and (2) the GPU parity tests also compare against committed data.
Files are .npz with one array `canvas` (vsize x hsize x 3, float64) plus the generating parameters.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import scene_feed as sf  # noqa: E402

CASES = [  # name, kind, args, depth
    ("cover_64x64", "yaml", ("cover", 64, 64), 5),
    ("reflective_world_80x40", "yaml", ("reflective_world", 80, 40), 5),
    ("ice_cream_80x40", "yaml", ("ice_cream", 80, 40), 5),
    ("teapot_48x24", "yaml", ("teapot", 48, 24), 5),
    ("stress_1500_64x36", "stress", (1500, 64, 36), 5),
]


def build(kind, args):
    if kind == "yaml":
        return sf.load(*args)
    return sf.rtc.stress_scene.build(args[0], args[1], args[2], with_teapot=False)


def main():
    for name, kind, args, depth in CASES:
        world, cam = build(kind, args)
        canvas = sf.to_oracle(world, cam).render(depth=depth)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), canvas=canvas, depth=np.int64(depth))
        print(name, canvas.shape, float(np.abs(canvas).sum()))


if __name__ == "__main__":
    main()
