#!/usr/bin/env python3
"""Decode the reference's golden EXRs (test/references/validate_*.exr) into
small committed fixtures under tests/golden/.

Runs only in the build container (needs /root/reference).  The decoder is
compiled in /tmp against the tinyexr header vendored by the reference; only the
decoded pixel data (float16, exactly the precision of the HALF EXRs) is
committed, as tests/golden/validate_<name>.npz.
"""
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
NAMES = ["distance", "world-pos", "view-pos", "world-normal", "view-normal", "albedo", "path-tracer"]
PIZ_FIXTURES = ("albedo", "view-normal", "path-tracer")


def main():
    exe = "/tmp/exr_to_raw"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-w", "-I", os.path.join(REF, "external"),
                           os.path.join(ROOT, "tools/exr_to_raw.cc"), "-o", exe, "-lpthread"])
    for n in NAMES:
        src = os.path.join(REF, "test/references", f"validate_{n}.exr")
        raw = f"/tmp/golden_{n}.raw"
        subprocess.check_call([exe, src, raw])
        d = open(raw, "rb").read()
        w, h = np.frombuffer(d[:8], dtype=np.int32)
        img = np.frombuffer(d[8:], dtype=np.float32).reshape(h, w, 3)
        img16 = img.astype(np.float16)
        assert np.array_equal(img16.astype(np.float32), img) or np.isnan(img).any(), "golden is not exactly half precision"
        np.savez_compressed(os.path.join(ROOT, "tests/golden", f"validate_{n}.npz"), rgb=img16)
        if n in PIZ_FIXTURES:     # the file itself (PIZ, half: written by Tauray through tinyexr) pins include/tauray_exr.hh's decoder
            shutil.copyfile(src, os.path.join(ROOT, "tests/golden", f"ref_piz_{n}.exr"))
        print(n, img.shape, "mean", img.reshape(-1, 3).mean(0), "min", img.min(), "max", img.max())


if __name__ == "__main__":
    sys.exit(main())
