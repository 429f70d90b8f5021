"""The C++17 host layer (include/tauray_hip.hh + tauray_amd/tauray_hip CLI): same frames as the Python mirror, the
reference's --fake-devices multi-GPU check, the headless file naming contract and the EXR channel layout."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CLI = os.path.join(ROOT, "tauray_amd", "tauray_hip")


def read_simple_exr(path):
    """Minimal reader for scanline, uncompressed EXR (what headless::write_exr emits): returns {channel: array}."""
    d = open(path, "rb").read()
    assert struct.unpack("<I", d[:4])[0] == 20000630
    pos = 8
    attrs = {}
    while d[pos] != 0:
        e = d.index(b"\0", pos); name = d[pos:e].decode(); pos = e + 1
        e = d.index(b"\0", pos); typ = d[pos:e].decode(); pos = e + 1
        size = struct.unpack("<i", d[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, d[pos:pos + size]); pos += size
    pos += 1
    chans, c = [], attrs["channels"][1]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p); n = c[p:e].decode(); p = e + 1
        ptype = struct.unpack("<i", c[p:p + 4])[0]; p += 16
        chans.append((n, ptype))
    assert attrs["compression"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = struct.unpack(f"<{h}Q", d[pos:pos + 8 * h])
    out = {n: np.zeros((h, w), dtype=np.float32) for n, _ in chans}
    for y in range(h):
        o = offsets[y]
        yy, nbytes = struct.unpack("<ii", d[o:o + 8]); o += 8
        for n, ptype in chans:
            dt, sz = (np.float16, 2) if ptype == 1 else (np.float32, 4)
            out[n][yy] = np.frombuffer(d, dtype=dt, count=w, offset=o).astype(np.float32); o += sz * w
    return [n for n, _ in chans], out


@pytest.fixture(scope="module")
def scene_dump(tmp_path_factory, test_glb_128):
    from tauray_amd.scene_io import write_scene_dump
    p = str(tmp_path_factory.mktemp("trsc") / "test.trsc")
    write_scene_dump(test_glb_128, p)
    return p


def test_cli_exists_and_links_only_the_c_abi():
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    needed = subprocess.run(["readelf", "-d", CLI], capture_output=True, text=True).stdout
    assert "libtrhip.so" in needed and "libtorch" not in needed and "libpython" not in needed


def test_cli_fails_loudly(scene_dump):
    r = subprocess.run([CLI, "/nonexistent.trsc"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to open" in r.stderr
    r = subprocess.run([CLI, scene_dump, "--bogus=1"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown option" in r.stderr


@pytest.mark.gpu
def test_cpp_renderer_matches_python_mirror_and_fake_devices(tmp_path, scene_dump, test_glb_128):
    from tauray_amd import renderer as R
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    W = H = 128
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, test_glb_128)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(test_glb_128, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    pt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    ref = disp.download((H, W, 4))
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw"]
    for tag, extra in (("one", []), ("fake3_scanline", ["--fake-devices=3", "--distribution-strategy=scanline"]),
                       ("fake4_strips", ["--fake-devices=4", "--distribution-strategy=shuffled-strips"])):
        prefix = str(tmp_path / tag)
        r = subprocess.run([CLI] + common + [f"--headless={prefix}", "-t"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "[path tracing (1 viewports)]" in r.stdout and "HOST:" in r.stdout
        got = np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4)
        assert np.array_equal(got, ref), tag


@pytest.mark.gpu
def test_headless_naming_and_exr_layout(tmp_path, scene_dump):
    prefix = str(tmp_path / "frame")
    r = subprocess.run([CLI, scene_dump, "--width=64", "--height=48", "--max-ray-depth=2", f"--headless={prefix}", "--frames=2",
                        "--format=rgb16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(tmp_path)) == ["frame0.exr", "frame1.exr"]          # prefix + frame number (src/headless.cc:305-309)
    names, ch = read_simple_exr(prefix + "0.exr")
    assert names == ["B", "G", "R"] and ch["R"].shape == (48, 64)               # B,G,R order, half (src/headless.cc:385-393)
    raw_prefix = str(tmp_path / "single")
    subprocess.check_call([CLI, scene_dump, "--width=64", "--height=48", "--max-ray-depth=2", f"--headless={raw_prefix}", "--filetype=raw"])
    raw = np.fromfile(raw_prefix + ".raw", dtype=np.float32).reshape(48, 64, 4)   # single frame: no number suffix
    assert np.array_equal(ch["R"], raw[..., 0].astype(np.float16).astype(np.float32))
    assert np.array_equal(ch["B"], raw[..., 2].astype(np.float16).astype(np.float32))
    assert not np.array_equal(read_simple_exr(prefix + "1.exr")[1]["R"], ch["R"])  # the sample counter advances between frames
    rgba = str(tmp_path / "rgba")
    subprocess.check_call([CLI, scene_dump, "--width=16", "--height=8", "--max-ray-depth=1", f"--headless={rgba}", "--format=rgba32"])
    names, ch = read_simple_exr(rgba + ".exr")
    assert names == ["A", "B", "G", "R"] and (ch["A"] == 1).all()
