#!/usr/bin/env python3
"""Writes tests/golden/sky.hdr: a 96 x 48 lat-long environment map (horizon gradient, a sun 4000 times brighter than the sky, a
dark ground with one exactly black row) in run-length-encoded RGBE, and sky_flat.hdr, the same pixels without run-length
encoding - the fixtures of the `--envmap` tests (the reference ships no .hdr file)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tauray_amd.hdr import write_hdr   # noqa: E402

W, H = 96, 48
SUN_DIR = (0.35, 0.55)        # u, v of the sun's centre


def build():
    v, u = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
    sky = np.stack([0.25 + 0.5 * v, 0.35 + 0.45 * v, 0.7 + 0.2 * v], axis=-1) * (v < 0.5)[..., None]
    ground = np.array([0.08, 0.07, 0.05]) * (v >= 0.5)[..., None]
    img = (sky + ground).astype(np.float32)
    d2 = ((u - SUN_DIR[0]) * 2.0) ** 2 + (v - SUN_DIR[1]) ** 2
    img += (np.exp(-d2 / 0.0008)[..., None] * np.array([4000.0, 3600.0, 3000.0])).astype(np.float32)
    img[H - 5] = 0.0
    here = os.path.dirname(os.path.abspath(__file__))
    for name, rle in (("sky.hdr", True), ("sky_flat.hdr", False)):
        path = os.path.join(here, "..", "tests", "golden", name)
        write_hdr(path, img, rle)
        print("wrote", os.path.normpath(path), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    build()
