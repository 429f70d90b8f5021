"""The C++17 host layer (include/tauray_hip.hh + tauray_amd/tauray_hip CLI): same frames as the Python mirror, the
reference's --fake-devices multi-GPU check, the headless file naming contract and the EXR channel layout."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CLI = os.path.join(ROOT, "tauray_amd", "tauray_hip")


@pytest.fixture(scope="module", autouse=True)
def fresh_cli(tmp_path_factory):
    """The tests run the command line built from the sources of this tree, not whatever binary travelled with it: the host layer
    is plain C++17 over the C ABI, so g++ and libtrhip.so are all it takes (the link line of tauray_amd/csrc/Makefile)."""
    global CLI
    out = str(tmp_path_factory.mktemp("cli") / "tauray_hip")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-DTAURAY_HIP_WITH_ZLIB", "-I" + os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tauray_amd", "host", "tauray_hip_cli.cc"), "-L" + os.path.join(ROOT, "tauray_amd"), "-ltrhip", "-ltrhip_comm", "-lz",
                           "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "tauray_amd"), "-Wl,-rpath-link,/opt/rocm/lib"])
    CLI = out
    yield out


def read_simple_exr(path):
    """Independent reader for the scanline EXR files headless::write_exr emits (no compression, RLE, ZIPS, ZIP; PIZ see below):
    returns ([channel names in file order], {channel: array}, compression code)."""
    import zlib
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
    comp = attrs["compression"][1][0]
    if comp == 4:     # PIZ: through include/tauray_exr.hh's decoder, which tests/test_exr.py pins to the reference's own PIZ files
        from tauray_amd.exr import decode_exr
        img = decode_exr(d)
        order = [n for n in "RGBA" if n in [c for c, _ in chans]]
        return [n for n, _ in chans], {n: img[..., i] for i, n in enumerate(order)}, comp
    lines = {0: 1, 1: 1, 2: 1, 3: 16}[comp]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    n_blocks = (h + lines - 1) // lines
    offsets = struct.unpack(f"<{n_blocks}Q", d[pos:pos + 8 * n_blocks])
    words = [1 if t == 1 else 2 for _, t in chans]
    out = {n: np.zeros((h, w), dtype=np.float32) for n, _ in chans}

    def unpredict(buf):   # inverse of OpenEXR's delta predictor + even/odd byte split (ZIP)
        t = np.frombuffer(buf, dtype=np.uint8).astype(np.int64)
        t[1:] -= 128
        t = (np.cumsum(t) & 0xFF).astype(np.uint8)
        n = len(t)
        o = np.empty(n, dtype=np.uint8)
        o[0::2] = t[:(n + 1) // 2]
        o[1::2] = t[(n + 1) // 2:]
        return o.tobytes()

    for b in range(n_blocks):
        o = offsets[b]
        yy, nbytes = struct.unpack("<ii", d[o:o + 8]); o += 8
        ny = min(lines, h - yy)
        raw_size = sum(words) * 2 * w * ny
        buf = d[o:o + nbytes]
        if nbytes < raw_size:
            assert comp in (1, 2, 3)
            if comp == 1:     # run lengths: count c < 0 -> -c literal bytes, else the next byte c + 1 times
                o2, q = bytearray(), 0
                while q < len(buf):
                    c = buf[q] - 256 if buf[q] > 127 else buf[q]; q += 1
                    if c < 0:
                        o2 += buf[q:q - c]; q -= c
                    else:
                        o2 += bytes([buf[q]]) * (c + 1); q += 1
                buf = unpredict(bytes(o2))
            else:
                buf = unpredict(zlib.decompress(buf))
        assert len(buf) == raw_size
        q = 0
        for y in range(yy, yy + ny):
            for n, ptype in chans:
                dt, sz = (np.float16, 2) if ptype == 1 else (np.float32, 4)
                out[n][y] = np.frombuffer(buf, dtype=dt, count=w, offset=q).astype(np.float32); q += sz * w
    return [n for n, _ in chans], out, comp


@pytest.fixture(scope="module")
def scene_dump(tmp_path_factory, test_glb_128):
    from tauray_amd.scene_io import write_scene_dump
    p = str(tmp_path_factory.mktemp("trsc") / "test.trsc")
    write_scene_dump(test_glb_128, p)
    return p


def test_cli_exists_and_links_only_the_c_abi():
    assert os.path.exists(os.path.join(ROOT, "tauray_amd", "tauray_hip")), "run __graft_entry__.build()"
    needed = subprocess.run(["readelf", "-d", CLI], capture_output=True, text=True).stdout
    assert "libtrhip.so" in needed and "libtorch" not in needed and "libpython" not in needed


def test_comm_id_file_is_not_taken_from_another_job(tmp_path):
    """exchange_comm_id_through_file (include/tauray_hip_comm.hh): a rank other than 0 accepts the id file only when it carries the job's
    nonce - a file of an earlier or crashed job (other nonce, no header, or, without a nonce, older than the reader) is waited out."""
    import struct
    import time
    exe = str(tmp_path / "comm_id_file_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "comm_id_file_check.cc"),
                           "-L" + os.path.join(ROOT, "tauray_amd"), "-ltrhip", "-ltrhip_comm", "-Wl,-rpath," + os.path.join(ROOT, "tauray_amd"),
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    path = str(tmp_path / "job.id")
    run = lambda nonce, timeout=0.3: subprocess.run([exe, path, str(nonce), str(timeout)], capture_output=True, text=True, check=True).stdout.strip()
    write = lambda nonce, first: open(path, "wb").write(b"TRHIPCID" + struct.pack("<Q", nonce) + bytes([first]) + bytes(127))
    assert run(7) == "timeout"                     # no file
    open(path, "wb").write(bytes([9]) * 128)       # a file of the old format (a bare id): not ours
    assert run(7) == "timeout" and run(0) == "timeout"
    write(5, 11)                                   # another job's nonce
    assert run(7) == "timeout"
    write(7, 12)
    assert run(7) == "id 12"
    write(0, 13)                                   # no nonce: a fresh file is taken ...
    assert run(0) == "id 13"
    old = time.time() - 3600
    os.utime(path, (old, old))                     # ... an hour-old one is a leftover
    assert run(0) == "timeout"
    # the writer arrives while the reader waits
    os.remove(path)
    p = subprocess.Popen([exe, path, "42", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.3)
    write(41, 1)
    time.sleep(0.2)
    tmp = path + ".tmp"
    open(tmp, "wb").write(b"TRHIPCID" + struct.pack("<Q", 42) + bytes([77]) + bytes(127))
    os.rename(tmp, path)
    assert p.communicate(timeout=30)[0].strip() == "id 77"


def test_exr_writer_all_compressions(tmp_path):
    """tr::headless writes what src/headless.cc:349-422 writes through tinyexr: scanline EXR, channels in alphabetical order,
    half or float, NONE / RLE / ZIPS / ZIP / PIZ (the default, src/headless.hh:56).  Every combination is read back by the reader above."""
    exe = str(tmp_path / "exr_writer_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-DTAURAY_HIP_WITH_ZLIB", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "exr_writer_check.cc"), "-L" + os.path.join(ROOT, "tauray_amd"), "-ltrhip", "-lz",
                           "-Wl,-rpath," + os.path.join(ROOT, "tauray_amd")])
    sizes = {}
    for (w, h) in ((64, 48), (33, 70), (5, 3), (1, 1)):
        for fmt in range(4):
            for comp in (0, 1, 2, 3, 4):
                exr, raw = str(tmp_path / "o.exr"), str(tmp_path / "o.raw")
                subprocess.check_call([exe, str(w), str(h), str(fmt), str(comp), exr, raw])
                names, ch, c = read_simple_exr(exr)
                assert c == comp and names == (["A", "B", "G", "R"] if fmt >= 2 else ["B", "G", "R"])
                src = np.fromfile(raw, dtype=np.float32).reshape(h, w, 4)
                for i, n in enumerate("RGBA"):
                    if n in ch:
                        want = src[..., i] if fmt in (1, 3) else src[..., i].astype(np.float16).astype(np.float32)
                        assert np.array_equal(ch[n], want), (w, h, fmt, comp, n)
                sizes[(w, h, fmt, comp)] = os.path.getsize(exr)
    assert sizes[(64, 48, 3, 3)] < sizes[(64, 48, 3, 0)] and sizes[(64, 48, 3, 2)] < sizes[(64, 48, 3, 0)]   # they do compress


def test_cpp_glb_loader_matches_python_loader(tmp_path):
    """tr::load_glb (include/tauray_gltf.hh: the C++ host's reader for src/gltf.cc's path-tracer subset) against the Python
    mirror's loader on the reference's own test/test.glb: every array of the flattened scene - instances with their model /
    normal matrices and materials, spans, vertices (including the tangents mesh::calculate_tangents makes up for the teapot,
    which has no uvs: NaN), indices, lights, the PNG texture, non-opaque flags - byte for byte; the camera block to 1e-12 (the
    two inverses differ in how they round a zero)."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene_io import write_scene_dump
    names = ["instances", "spans", "vertices", "indices", "point_lights", "directional_lights", "texture_infos", "texels", "envmap",
             "alias_table", "cameras", "non_opaque"]

    def sections(path):
        d = open(path, "rb").read()
        assert d[:4] == b"TRSC"
        pos, out = 8, {}
        for n in names:
            size = struct.unpack("<Q", d[pos:pos + 8])[0]
            out[n] = d[pos + 8:pos + 8 + size]
            pos += 8 + size
        out["tail"] = d[pos:]
        return out

    for (w, h) in ((128, 128), (1920, 1080)):
        cpp, py = str(tmp_path / "cpp.trsc"), str(tmp_path / "py.trsc")
        subprocess.check_call([CLI, os.path.join(GOLDEN, "test.glb"), f"--width={w}", f"--height={h}", f"--dump-scene={cpp}"])
        write_scene_dump(load_glb(os.path.join(GOLDEN, "test.glb"), w, h), py)
        a, b = sections(cpp), sections(py)
        for n in names + ["tail"]:
            if n == "cameras":
                x, y = np.frombuffer(a[n], np.float32), np.frombuffer(b[n], np.float32)
                assert x.shape == y.shape and np.abs(x.astype(np.float64) - y).max() < 1e-12
            else:
                assert a[n] == b[n], f"{n} differs at {w}x{h}"
        assert len(a["vertices"]) == 48 * 45827 or len(a["vertices"]) > 0
    # the skinned fixture: same bind-pose scene, same {joints, weights} per vertex, same rest-pose joint matrices
    cpp, py = str(tmp_path / "s.trsc"), str(tmp_path / "s_py.trsc")
    subprocess.check_call([CLI, os.path.join(GOLDEN, "skinned.glb"), "--width=64", "--height=64", f"--dump-scene={cpp}"])
    sc = load_glb(os.path.join(GOLDEN, "skinned.glb"), 64, 64)
    write_scene_dump(sc, py)
    a, b = sections(cpp), sections(py)
    for n in names + ["tail"]:
        if n != "cameras":
            assert a[n] == b[n], f"skinned.glb: {n} differs"
    d = open(cpp + ".skins", "rb").read()
    pos = 0
    assert len(sc.skinned) >= 1
    for sk in sc.skinned:
        inst, nv, nj = struct.unpack("<3I", d[pos:pos + 12]); pos += 12
        assert inst == sk.instance and nv == len(sk.skins)
        assert d[pos:pos + 32 * nv] == np.ascontiguousarray(sk.skins).tobytes(); pos += 32 * nv
        jt = np.frombuffer(d, np.float32, 16 * nj, pos).reshape(nj, 4, 4); pos += 64 * nj
        want = np.stack([m.T for m in sc.joint_transforms(sk)])      # column-major storage
        assert np.abs(jt.astype(np.float64) - want).max() < 1e-6
    assert pos == len(d)
    # the forms real assets come in (tests/golden/textured, tools/make_textured_gltf.py): text glTF with an external buffer and
    # external images - a baseline 4:2:0 JPEG, an interlaced palette PNG behind a percent-encoded name, a 16-bit PNG -, the same
    # scene as a .glb with the files embedded and as a .gltf with data: URIs: six loads, one scene, byte for byte
    dumps = {}
    for name in ("room.gltf", "room_embedded.glb", "room_datauri.gltf"):
        cpp, py = str(tmp_path / (name + ".trsc")), str(tmp_path / (name + "_py.trsc"))
        subprocess.check_call([CLI, os.path.join(GOLDEN, "textured", name), "--width=96", "--height=96", f"--dump-scene={cpp}"])
        write_scene_dump(load_glb(os.path.join(GOLDEN, "textured", name), 96, 96), py)
        a, b = sections(cpp), sections(py)
        for n in names + ["tail"]:
            if n != "cameras":
                assert a[n] == b[n], f"{name}: {n} differs"
        dumps[name] = a
    for name in ("room_embedded.glb", "room_datauri.gltf"):
        for n in names:
            assert dumps[name][n] == dumps["room.gltf"][n], f"{name} vs room.gltf: {n}"
    # two RGBA8 textures and the 16-bit PNG as RGBA16 (8 bytes per texel), what the reference keeps as R16G16B16A16Unorm (src/gltf.cc:548-556)
    assert len(dumps["room.gltf"]["texels"]) == 4 * (64 * 48 + 20 * 12) + 8 * (32 * 32)
    infos = np.frombuffer(dumps["room.gltf"]["texture_infos"], np.uint32).reshape(-1, 4)
    assert list(infos[:, 3]) == [0, 0, 1] and list(infos[:, 2]) == [0, 64 * 48, 64 * 48 + 20 * 12]
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb file at all")
    r = subprocess.run([CLI, str(bad), f"--dump-scene={tmp_path / 'x.trsc'}"], capture_output=True, text=True)
    assert r.returncode == 1 and "not a GLB" in r.stderr


def test_cpp_camera_grid_matches_python(tmp_path):
    """generate_cameras (src/tauray.cc:680-727) in both hosts: `tauray_hip --camera-grid=w,h,x,y` leaves the camera blocks
    tauray_amd.scene.generate_camera_grid packs (pan in the projection matrix and in camera_data.pan)."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import generate_camera_grid
    glb = os.path.join(GOLDEN, "test.glb")
    for (gw, gh, dx, dy, rec, roll) in ((3, 2, 0.02, 0.02, 5.0, 0.0), (9, 5, 0.02, 0.02, 5.0, 0.0), (2, 2, 0.1, 0.05, 3.0, 30.0)):
        cpp = str(tmp_path / "grid.trsc")
        subprocess.check_call([CLI, glb, "--width=160", "--height=90", f"--camera-grid={gw},{gh},{dx},{dy}", f"--camera-recentering-distance={rec}",
                               f"--camera-grid-roll={roll}", f"--dump-scene={cpp}"])
        d = open(cpp, "rb").read()
        pos, secs = 8, []
        for _ in range(12):
            n = struct.unpack("<Q", d[pos:pos + 8])[0]
            secs.append(d[pos + 8:pos + 8 + n]); pos += 8 + n
        got = np.frombuffer(secs[10], np.float32).reshape(gw * gh, 80)
        sc = load_glb(glb, 160, 90)
        want = np.concatenate([c.pack() for c in generate_camera_grid(sc.cameras[0], gw, gh, dx, dy, rec, roll)]).view(np.float32).reshape(gw * gh, 80)
        assert got.shape == want.shape and np.abs(got.astype(np.float64) - want).max() < 2e-6, (gw, gh)
        assert np.abs(got[:, 76:78]).max() > 0      # the views are panned


def test_cpp_animator_matches_python_animator(tmp_path):
    """tr::scene_animator (include/tauray_gltf.hh) against tauray_amd.animation.SceneAnimator on tests/golden/animated.glb: after
    the same number of updates at the same frame rate - default clip by fallback, the named clip "spin", a frame rate whose step
    jumps over keys - the instance records (model and model_prev bit for bit, the normal matrices to 1e-12), the cameras and the
    skin's joint matrices agree, and so does the record of the point light that rides on an animated node."""
    from tauray_amd import scene as S
    from tauray_amd.animation import SceneAnimator
    from tauray_amd.gltf import load_glb
    glb = os.path.join(GOLDEN, "animated.glb")
    dump = str(tmp_path / "an.trsc")
    for name, frames, fps in (("", 1, 24), ("", 7, 24), ("", 29, 24), ("spin", 5, 60), ("", 2, 3)):
        subprocess.check_call([CLI, glb, "--width=64", "--height=64", "--animation" + ("=" + name if name else ""), f"--framerate={fps}",
                               f"--frames={frames}", f"--dump-scene={dump}"])
        sc = load_glb(glb, 64, 64)
        an = SceneAnimator(sc)
        an.play(name)
        dt = int(np.floor(1000000.0 / fps + 0.5))
        for f in range(frames):
            inst, cams, globs = an.update(0 if f == 0 else dt)
        raw = open(dump, "rb").read()
        pos, secs = 8, []
        for _ in range(12):
            n = struct.unpack_from("<Q", raw, pos)[0]
            secs.append(raw[pos + 8:pos + 8 + n])
            pos += 8 + n
        got = np.frombuffer(secs[0], dtype=S.INSTANCE)
        what = f"clip {name or '<any>'}, {frames} updates at {fps} fps"
        assert np.array_equal(got["model"], inst["model"]) and np.array_equal(got["model_prev"], inst["model_prev"]), what
        assert np.abs(got["model_normal"].astype(np.float64) - inst["model_normal"]).max() < 1e-12, what
        assert frames == 1 or not np.array_equal(got["model"], got["model_prev"]), what
        cam = np.frombuffer(secs[10], dtype=np.float32)
        assert np.abs(cam.astype(np.float64) - np.concatenate([c.pack() for c in cams]).view(np.float32)).max() < 1e-12, what
        assert secs[4] == sc.point_lights.tobytes() and secs[5] == sc.directional_lights.tobytes(), what + ": lights"     # the lamp drifts in "move"
        d = open(dump + ".skins", "rb").read()
        _, nv, nj = struct.unpack_from("<3I", d, 0)
        joints = np.frombuffer(d[12 + nv * 32:12 + nv * 32 + nj * 64], dtype=np.float32).reshape(nj, 4, 4)
        want = np.stack([S.to_glm(m) for m in sc.joint_transforms(sc.skinned[0], globs)])
        assert np.abs(joints.astype(np.float64) - want).max() < 1e-6, what


def test_cpp_envmap_matches_python(tmp_path):
    """`tauray_hip --envmap=file.hdr` (include/tauray_envmap.hh: Radiance RGBE reader + environment_map::generate_alias_table)
    against the Python mirror (tauray_amd/hdr.py, scene.build_alias_table): the texels, every entry of the alias table and the
    environment factor, byte for byte, for the run-length-encoded and the flat fixture; a file that is not an .hdr fails loudly."""
    from tauray_amd.hdr import load_hdr
    from tauray_amd.scene import build_alias_table
    dump = str(tmp_path / "env.trsc")
    for name in ("sky.hdr", "sky_flat.hdr"):
        subprocess.check_call([CLI, os.path.join(GOLDEN, "test.glb"), "--width=64", "--height=64", f"--envmap={os.path.join(GOLDEN, name)}", f"--dump-scene={dump}"])
        raw = open(dump, "rb").read()
        pos, secs = 8, []
        for _ in range(12):
            n = struct.unpack_from("<Q", raw, pos)[0]
            secs.append(raw[pos + 8:pos + 8 + n])
            pos += 8 + n
        env = load_hdr(os.path.join(GOLDEN, name))
        assert secs[8] == env.tobytes(), name
        assert secs[9] == build_alias_table(env).tobytes(), name
        assert struct.unpack("<II4f", raw[pos:pos + 24]) == (96, 48, 1.0, 1.0, 1.0, 1.0)
    r = subprocess.run([CLI, os.path.join(GOLDEN, "test.glb"), f"--envmap={os.path.join(GOLDEN, 'test.glb')}", f"--dump-scene={dump}"], capture_output=True, text=True)
    assert r.returncode != 0 and "not a Radiance" in r.stderr
    # the same sky as an OpenEXR file (src/texture.cc:409-429): float channels, three of them -> alpha 1, no half rounding; PIZ and ZIP
    from tauray_amd import exr
    from tauray_amd.hdr import set_envmap
    sky = load_hdr(os.path.join(GOLDEN, "sky.hdr"))
    sky[..., :3] *= np.float32(1.0 / 3.0)              # not representable in half
    for comp, half in ((exr.PIZ, False), (exr.ZIP, True)):
        f = str(tmp_path / "sky.exr")
        open(f, "wb").write(exr.encode_exr(sky, alpha=False, half=half, compression=comp))
        subprocess.check_call([CLI, os.path.join(GOLDEN, "test.glb"), "--width=64", "--height=64", f"--envmap={f}", f"--dump-scene={dump}"])
        raw = open(dump, "rb").read()
        pos, secs = 8, []
        for _ in range(12):
            n = struct.unpack_from("<Q", raw, pos)[0]
            secs.append(raw[pos + 8:pos + 8 + n])
            pos += 8 + n

        class S:
            pass
        env = set_envmap(S(), f).envmap
        want = sky.astype(np.float16).astype(np.float32) if half else sky
        want[..., 3] = 1.0
        assert np.array_equal(env, want) and env.dtype == np.float32
        assert secs[8] == env.tobytes() and secs[9] == build_alias_table(env).tobytes(), (comp, half)


def test_cpp_envmap_random_images_match_python(tmp_path):
    """Seeded random environment maps - sizes that are multiples of nothing, a few bright texels over a dim sky, black texels, whole black
    rows, values beyond half precision - as .hdr (run-length encoded and flat) and as .exr: the C++ host's texels and alias table equal the
    Python mirror's byte for byte.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns."""
    from tauray_amd import exr
    from tauray_amd.hdr import set_envmap, write_hdr
    from tauray_amd.scene import build_alias_table
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "19")))
    dump = str(tmp_path / "env.trsc")
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "3"))):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 100))
        sky = (rng.uniform(0, 1, (h, w, 3)) ** 4 * rng.choice([0.01, 1.0, 50.0])).astype(np.float32)
        for _ in range(int(rng.integers(0, 4))):
            sky[int(rng.integers(0, h)), int(rng.integers(0, w))] = rng.uniform(1e3, 2e5, 3)      # suns, some past the half-precision clamp
        if rng.uniform() < 0.3:
            sky[int(rng.integers(0, h))] = 0
        if rng.uniform() < 0.3:
            sky[rng.uniform(0, 1, (h, w)) < 0.3] = 0
        kind = k % 3
        f = str(tmp_path / ("sky.exr" if kind == 2 else "sky.hdr"))
        if kind == 2:
            half = bool(rng.integers(0, 2))
            if half:
                sky = np.minimum(sky, np.float32(6e4))      # a half file cannot hold more; infinite texels are refused (below)
            rgba = np.concatenate([sky, np.ones((h, w, 1), np.float32)], axis=-1)
            open(f, "wb").write(exr.encode_exr(rgba, alpha=False, half=half, compression=int(rng.choice([exr.NONE, exr.ZIP, exr.PIZ]))))
        else:
            write_hdr(f, sky, rle=kind == 0)
        subprocess.check_call([CLI, os.path.join(GOLDEN, "test.glb"), "--width=32", "--height=32", f"--envmap={f}", f"--dump-scene={dump}"])
        raw = open(dump, "rb").read()
        pos, secs = 8, []
        for _ in range(12):
            n = struct.unpack_from("<Q", raw, pos)[0]
            secs.append(raw[pos + 8:pos + 8 + n])
            pos += 8 + n

        class S:
            pass
        env = set_envmap(S(), f).envmap
        assert env.shape == (h, w, 4) and secs[8] == env.tobytes(), f"draw {k}: {w}x{h} kind {kind}: texels differ"
        assert secs[9] == build_alias_table(env).tobytes(), f"draw {k}: {w}x{h} kind {kind}: alias tables differ"
    # a texel that overflowed to infinity on its way into a half file: both hosts refuse the map instead of rendering NaN
    f = str(tmp_path / "inf.exr")
    rgba = np.ones((4, 8, 4), np.float32); rgba[1, 2, 0] = 1e6
    open(f, "wb").write(exr.encode_exr(rgba, alpha=False, half=True, compression=exr.ZIP))
    r = subprocess.run([CLI, os.path.join(GOLDEN, "test.glb"), "--width=32", "--height=32", f"--envmap={f}", f"--dump-scene={dump}"], capture_output=True, text=True)
    assert r.returncode != 0 and "non-finite" in r.stderr

    class S2:
        pass
    with pytest.raises(ValueError, match="non-finite"):
        set_envmap(S2(), f)


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
    glb_common = [os.path.join(GOLDEN, "test.glb")] + common[1:]      # the same frame straight from the reference's .glb
    for tag, extra, args in (("one", [], common), ("fake3_scanline", ["--fake-devices=3", "--distribution-strategy=scanline"], common),
                             ("fake4_strips", ["--fake-devices=4", "--distribution-strategy=shuffled-strips"], common), ("from_glb", [], glb_common)):
        prefix = str(tmp_path / tag)
        r = subprocess.run([CLI] + args + [f"--headless={prefix}", "-t"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "[path tracing (1 viewports)]" in r.stdout and "HOST:" in r.stdout
        got = np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4)
        assert np.array_equal(got, ref), tag
    # rt_renderer<direct_stage> (tr::direct_renderer, --renderer=direct) against the Python DirectStage
    dopt = R.options_for_scene(test_glb_128, max_bounces=4, samples_per_pixel=2, samples_per_pass=2)
    dt = R.DirectStage(ctx, ss, dopt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color.zero()
    dt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    ref = disp.download((H, W, 4))
    for tag, extra in (("direct", []), ("direct_fake2", ["--fake-devices=2", "--distribution-strategy=scanline"]), ("direct_slots", ["--frames-in-flight=2"])):
        prefix = str(tmp_path / tag)
        subprocess.check_call([CLI] + common + [f"--headless={prefix}", "--renderer=direct", "--samples-per-pixel=2", "--samples-per-pass=2"] + extra)
        assert np.array_equal(np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4), ref), tag
    r = subprocess.run([CLI] + common + ["--renderer=whitted"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown renderer" in r.stderr


@pytest.mark.gpu
def test_textured_gltf_renders_the_same_through_both_hosts_and_like_the_oracle(tmp_path):
    """tests/golden/textured/room.gltf (external buffer, JPEG / interlaced palette PNG / 16-bit PNG textures): the C++ host renders
    the file the Python mirror renders, bit for bit, and the frame agrees with the CPU oracle on the same flattened scene."""
    from tauray_amd import renderer as R
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    from tauray_amd.gltf import load_glb
    from oracle import binding as B
    W = H = 96
    path = os.path.join(GOLDEN, "textured", "room.gltf")
    scene = load_glb(path, W, H)
    assert len(scene.textures) == 3 and scene.triangle_count == 16
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=4), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    pt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    img, shown = color.download((H, W, 4)), disp.download((H, W, 4))
    prefix = str(tmp_path / "room")
    subprocess.check_call([CLI, path, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", f"--headless={prefix}"])
    assert np.array_equal(np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4), shown)
    ref = B.OracleScene(scene).render_pt(B.options_for_scene(scene, max_bounces=4), W, H)[0]
    rel = np.abs(img[..., :3] - ref[..., :3]) / (np.abs(ref[..., :3]) + 1e-2)
    assert float((rel.max(-1) > 1e-2).mean()) <= 2e-3 and img[..., :3].mean() > 0.01
    assert abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) / float(ref[..., :3].mean()) < 2e-3


@pytest.mark.gpu
def test_cpp_camera_grid_and_view_shards(tmp_path):
    """BASELINE config 5's shape through the C++ host: a camera grid rendered as the layers of one launch (one file per view,
    `prefix<view>_.raw`), equal to the Python mirror's views bit for bit; and divided among processes by views (--shard=views:
    viewport v on rank v mod N, no exchange) the ranks write the same files."""
    from tauray_amd import renderer as R
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene import generate_camera_grid
    W, H, gw, gh = 96, 54, 3, 2
    glb = os.path.join(GOLDEN, "test.glb")
    scene = load_glb(glb, W, H)
    scene.cameras = generate_camera_grid(scene.cameras[0], gw, gh, 0.02, 0.02, 5.0)
    ctx = R.Context(0)
    rr = R.RtRenderer(ctx, scene, R.options_for_scene(scene, max_bounces=4), (W, H), viewports=gw * gh)
    rr.render()
    ref = rr.download("display")
    common = [glb, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", f"--camera-grid={gw},{gh},0.02,0.02"]
    prefix = str(tmp_path / "grid")
    subprocess.check_call([CLI] + common + [f"--headless={prefix}"])
    for v in range(gw * gh):
        assert np.array_equal(np.fromfile(f"{prefix}{v}_.raw", dtype=np.float32).reshape(H, W, 4), ref[v]), v
    assert not np.array_equal(ref[0], ref[gw * gh - 1])
    sprefix = str(tmp_path / "shard")
    for rank in range(4):      # four ranks for six views: 2, 2, 1, 1
        subprocess.check_call([CLI] + common + [f"--headless={sprefix}", "--shard=views", "--process-count=4", f"--process-rank={rank}", "--device=0"])
    for v in range(gw * gh):
        assert np.array_equal(np.fromfile(f"{sprefix}{v}_.raw", dtype=np.float32).reshape(H, W, 4), ref[v]), ("shard", v)


@pytest.mark.gpu
def test_cpp_one_process_per_gpu_mode_with_one_rank(tmp_path, scene_dump):
    """tr::process_rt_renderer (include/tauray_hip_comm.hh) behind `--process-count=N --process-rank=R --comm-id=file`: the RCCL
    communicator is created from the id rank 0 leaves in the file, the frame goes through trhip_gather_partials; with one rank -
    what a one-GPU box can run - the files equal the single-process renderer's, frame slots included."""
    W = H = 96
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", "--frames=3"]
    ref_prefix = str(tmp_path / "ref")
    subprocess.check_call([CLI] + common + [f"--headless={ref_prefix}"])
    for tag, extra in (("proc", []), ("proc_strips", ["--distribution-strategy=shuffled-strips", "--device-workloads=1"])):
        prefix, idf = str(tmp_path / tag), str(tmp_path / (tag + ".id"))
        r = subprocess.run([CLI] + common + [f"--headless={prefix}", "--process-count=1", "--process-rank=0", "--device=0", f"--comm-id={idf}", "-t"] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert not os.path.exists(idf) and "RANK 0" in r.stdout      # rank 0 removes the id file once the communicator exists
        for f in range(3):
            assert np.array_equal(np.fromfile(f"{prefix}{f}.raw", dtype=np.float32), np.fromfile(f"{ref_prefix}{f}.raw", dtype=np.float32)), (tag, f)
    r = subprocess.run([CLI] + common + ["--process-count=2", "--process-rank=2", f"--comm-id={tmp_path / 'x.id'}"], capture_output=True, text=True)
    assert r.returncode != 0 and "--process-rank" in r.stderr


@pytest.mark.gpu
def test_cpp_processes_exchange_frames_through_the_copy_engines(tmp_path, scene_dump):
    """tauray_hip --process-count=2 --exchange=ipc: two processes on the one GPU, the partial frames of the second written into the first
    one's IPC-mapped arena by hipMemcpyAsync (trhip_ipc_*, include/trhip_comm.h), frame slots included; the files are the single-process
    renderer's byte for byte."""
    W = H = 96
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=4", "--filetype=raw", "--frames=5"]
    ref_prefix = str(tmp_path / "ref")
    subprocess.check_call([CLI] + common + [f"--headless={ref_prefix}"])
    for tag, extra in (("ipc", []), ("ipc_slots", ["--frames-in-flight=2", "--distribution-strategy=shuffled-strips"])):
        prefix, idf = str(tmp_path / tag), str(tmp_path / (tag + ".id"))
        procs = [subprocess.Popen([CLI] + common + [f"--headless={prefix}", "--process-count=2", f"--process-rank={r}", "--device=0", f"--comm-id={idf}", "--exchange=ipc",
                                                    f"--comm-nonce={os.getpid()}"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        for p in procs:
            so, se = p.communicate(timeout=600)
            assert p.returncode == 0, se[-2000:]
        for f in range(5):
            assert np.array_equal(np.fromfile(f"{prefix}{f}.raw", dtype=np.float32), np.fromfile(f"{ref_prefix}{f}.raw", dtype=np.float32)), (tag, f)


@pytest.mark.gpu
def test_cpp_ranks_with_different_shading_programs_refuse_to_render(tmp_path, scene_dump):
    """process_rt_renderer::check_same_program: rank 1 started under TRHIP_SHADE_CLI=0 would shade its strips with a program compiled at
    run time where rank 0 runs the ahead-of-time instances - two implementations at the default arithmetic.  Both ranks stop before the first
    frame and name the two programs; the blob files of the set-up are gone afterwards in the good case."""
    W = H = 64
    common = [scene_dump, f"--width={W}", f"--height={H}", "--max-ray-depth=3", "--filetype=raw", "--frames=1"]
    for case, env1 in (("same", {}), ("differ", {"TRHIP_SHADE_CLI": "0"})):
        prefix, idf = str(tmp_path / case), str(tmp_path / (case + ".id"))
        procs = [subprocess.Popen([CLI] + common + [f"--headless={prefix}", "--process-count=2", f"--process-rank={r}", "--device=0", f"--comm-id={idf}", "--exchange=ipc",
                                                    f"--comm-nonce={os.getpid() + 7}"], env=dict(os.environ, **(env1 if r == 1 else {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        if case == "same":
            assert all(p.returncode == 0 for p in procs), [e[-1500:] for _, e in outs]
            left = [f for f in os.listdir(tmp_path) if f.startswith("same.id") and not f.endswith(".read")]
            assert left == [], left      # the ipc / prog blobs were removed once every rank had read them
        else:
            assert all(p.returncode != 0 for p in procs)
            assert all("different shading programs" in e and "command-line set, ahead of time" in e and "compiled" in e for _, e in outs), [e[-800:] for _, e in outs]
    r = subprocess.run([CLI] + common + [f"--headless={tmp_path / 'n'}", "--process-count=2", "--process-rank=0", "--device=0", f"--comm-id={tmp_path / 'n.id'}", "--exchange=ipc"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "nonce" in r.stderr      # blobs with IPC handles are never taken from a job that cannot be told apart


@pytest.mark.gpu
def test_cpp_renders_the_skinned_glb_like_the_python_mirror(tmp_path):
    """tests/golden/skinned.glb through tr::load_glb + scene_stage::set_scene (bind-pose vertices, skins, rest-pose joint
    matrices -> trhip_scene_set_skin / trhip_scene_skin before the build) against the Python mirror doing the same: the
    skinned tube is bent, not straight, and the frames agree (joint matrices may differ in the last bit: tolerance, not bits)."""
    from tauray_amd import renderer as R
    from tauray_amd.gltf import load_glb
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    W, H = 160, 120
    glb = os.path.join(GOLDEN, "skinned.glb")
    scene = load_glb(glb, W, H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=3), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    pt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    ref = disp.download((H, W, 4))
    prefix = str(tmp_path / "sk")
    subprocess.check_call([CLI, glb, f"--width={W}", f"--height={H}", "--max-ray-depth=3", "--filetype=raw", f"--headless={prefix}"])
    got = np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4)
    differing = float((np.abs(got - ref).max(-1) > 1e-3).mean())
    assert differing < 2e-3 and abs(float(got.mean()) - float(ref.mean())) < 1e-4, f"{differing:.4%} of the pixels differ"
    # and it is the posed mesh that was rendered: without the skins the (straight) bind pose gives another image
    bind = load_glb(glb, W, H)
    bind.skinned = []
    ss2 = R.SceneStage(R.Context(0), bind)
    pt2 = R.PathTracerStage(ss2.ctx, ss2, R.options_for_scene(bind, max_bounces=3), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    c2, d2 = ss2.ctx.alloc(W * H * 16).zero(), ss2.ctx.alloc(W * H * 16)
    pt2.run(c2)
    R.TonemapStage(ss2.ctx).run(c2, d2, W, H)
    assert float((np.abs(d2.download((H, W, 4)) - ref).max(-1) > 1e-3).mean()) > 0.01


@pytest.mark.gpu
def test_cpp_plays_the_animated_glb_like_the_python_mirror(tmp_path):
    """`tauray_hip animated.glb --animation --framerate=24`: frames until the clip ends (30: the timer reaches the last key in
    the 31st update), each after scene_animator::update + rt_renderer::update_scene (instances, joints, cameras, acceleration
    structure); checked frames agree with the Python mirror playing the same file (SceneAnimator + SceneStage.animate)."""
    from tauray_amd import renderer as R
    from tauray_amd.animation import SceneAnimator
    from tauray_amd.gltf import load_glb
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    W, H = 160, 120
    glb = os.path.join(GOLDEN, "animated.glb")
    prefix = str(tmp_path / "an")
    subprocess.check_call([CLI, glb, f"--width={W}", f"--height={H}", "--max-ray-depth=3", "--filetype=raw", "--animation", "--framerate=24", f"--headless={prefix}"])
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".raw"))
    assert len(files) == 30, files
    scene = load_glb(glb, W, H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    an = SceneAnimator(scene)
    an.play("")
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, max_bounces=3), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    shown = []
    for frame in range(30):
        ss.animate(an, 0 if frame == 0 else round(1000000.0 / 24.0), refit=(frame % 3 != 2))
        pt.reset_accumulated_samples()
        pt.run(color)                                        # the stage's frame counter advances like the renderer's
        if frame in (0, 9, 20, 29):
            R.TonemapStage(ctx).run(color, disp, W, H)
            ref = disp.download((H, W, 4))
            got = np.fromfile(f"{prefix}{frame}.raw", dtype=np.float32).reshape(H, W, 4)
            differing = float((np.abs(got - ref).max(-1) > 1e-3).mean())
            assert differing < 2e-3 and abs(float(got.mean()) - float(ref.mean())) < 1e-4, f"frame {frame}: {differing:.4%} of the pixels differ"
            shown.append(got)
    assert all(float((np.abs(shown[k] - shown[k + 1]).max(-1) > 1e-3).mean()) > 0.01 for k in range(3)), "the frames do not move"


@pytest.mark.gpu
def test_cpp_renders_with_an_envmap_like_the_python_mirror_and_the_oracle(tmp_path):
    """test.glb lit by tests/golden/sky.hdr (a sun 4000 times brighter than the sky: the alias table matters): the C++ host's
    frame agrees with the Python mirror's, and the mirror's with the oracle's."""
    from tauray_amd import renderer as R
    from tauray_amd.gltf import load_glb
    from tauray_amd.hdr import set_envmap
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    from oracle import binding as oracle
    W, H = 160, 120
    glb, hdr = os.path.join(GOLDEN, "test.glb"), os.path.join(GOLDEN, "sky.hdr")
    scene = set_envmap(load_glb(glb, W, H), hdr)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    opt = R.options_for_scene(scene, max_bounces=3)
    assert opt.nee_envmap > 0
    pt = R.PathTracerStage(ctx, ss, opt, DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
    pt.run(color)
    R.TonemapStage(ctx).run(color, disp, W, H)
    ref = disp.download((H, W, 4))
    prefix = str(tmp_path / "env")
    subprocess.check_call([CLI, glb, f"--width={W}", f"--height={H}", "--max-ray-depth=3", "--filetype=raw", f"--envmap={hdr}", f"--headless={prefix}"])
    got = np.fromfile(prefix + ".raw", dtype=np.float32).reshape(H, W, 4)
    differing = float((np.abs(got - ref).max(-1) > 1e-3).mean())
    assert differing < 2e-3 and abs(float(got.mean()) - float(ref.mean())) < 1e-4, f"{differing:.4%} of the pixels differ"
    plain = load_glb(glb, W, H)
    ss2 = R.SceneStage(ctx, plain)
    pt2 = R.PathTracerStage(ctx, ss2, R.options_for_scene(plain, max_bounces=3), DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True))
    c2 = ctx.alloc(W * H * 16).zero()
    pt2.run(c2)
    assert abs(float(c2.download((H, W, 4))[..., :3].mean()) - float(color.download((H, W, 4))[..., :3].mean())) > 0.01, "the environment map lit nothing"
    hip = color.download((1, H, W, 4))
    want = oracle.OracleScene(scene).render_pt(oracle.options_for_scene(scene, max_bounces=3), W, H)
    rel = np.abs(hip[..., :3] - want[..., :3]) / (np.abs(want[..., :3]) + 1e-2)
    assert float((rel.max(-1) > 1e-2).mean()) < 2e-3 and abs(float(hip[..., :3].mean()) - float(want[..., :3].mean())) < 2e-3 * float(want[..., :3].mean())


@pytest.mark.gpu
def test_headless_naming_and_exr_layout(tmp_path, scene_dump):
    prefix = str(tmp_path / "frame")
    r = subprocess.run([CLI, scene_dump, "--width=64", "--height=48", "--max-ray-depth=2", f"--headless={prefix}", "--frames=2",
                        "--format=rgb16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(tmp_path)) == ["frame0.exr", "frame1.exr"]          # prefix + frame number (src/headless.cc:305-309)
    names, ch, comp = read_simple_exr(prefix + "0.exr")
    assert comp == 4                                                                 # PIZ by default (src/headless.hh:56)
    assert names == ["B", "G", "R"] and ch["R"].shape == (48, 64)               # B,G,R order, half (src/headless.cc:385-393)
    raw_prefix = str(tmp_path / "single")
    subprocess.check_call([CLI, scene_dump, "--width=64", "--height=48", "--max-ray-depth=2", f"--headless={raw_prefix}", "--filetype=raw"])
    raw = np.fromfile(raw_prefix + ".raw", dtype=np.float32).reshape(48, 64, 4)   # single frame: no number suffix
    assert np.array_equal(ch["R"], raw[..., 0].astype(np.float16).astype(np.float32))
    assert np.array_equal(ch["B"], raw[..., 2].astype(np.float16).astype(np.float32))
    assert not np.array_equal(read_simple_exr(prefix + "1.exr")[1]["R"], ch["R"])  # the sample counter advances between frames
    rgba = str(tmp_path / "rgba")
    subprocess.check_call([CLI, scene_dump, "--width=16", "--height=8", "--max-ray-depth=1", f"--headless={rgba}", "--format=rgba32", "--compression=none"])
    names, ch, comp = read_simple_exr(rgba + ".exr")
    assert comp == 0 and names == ["A", "B", "G", "R"] and (ch["A"] == 1).all()


@pytest.mark.gpu
def test_cpp_frames_in_flight_write_the_same_files(tmp_path, scene_dump):
    """tr::rt_renderer with frame slots (options.max_frames_in_flight): frame f renders while earlier frames are read back and
    written; every file is byte-identical to the one-frame-at-a-time run's."""
    common = [CLI, scene_dump, "--width=96", "--height=64", "--max-ray-depth=3", "--frames=7", "--filetype=raw", "--warmup-frames=2"]
    a, b = str(tmp_path / "serial"), str(tmp_path / "slots")
    subprocess.check_call(common + [f"--headless={a}"])
    subprocess.check_call(common + [f"--headless={b}", "--frames-in-flight=3"])
    for f in range(7):
        x, y = open(f"{a}{f}.raw", "rb").read(), open(f"{b}{f}.raw", "rb").read()
        assert len(x) == 96 * 64 * 16 and x == y, f"frame {f} differs"
    assert open(f"{a}0.raw", "rb").read() != open(f"{a}1.raw", "rb").read()
    # frame slots across devices (src/rt_renderer.cc:84-133 never blocks the host): eight fake devices, four slots, both
    # distribution strategies - the same bytes as one device, one frame at a time
    for tag, extra in (("scan", ["--fake-devices=8", "--frames-in-flight=4", "--distribution-strategy=scanline"]),
                       ("strips", ["--fake-devices=8", "--frames-in-flight=4", "--distribution-strategy=shuffled-strips"]),
                       ("two", ["--fake-devices=2", "--frames-in-flight=2"]),
                       # rt_renderer::options::frames_per_launch: B consecutive frames per render() (trhip_pt_set_frame_batch), seven frames
                       # are two launches of three and one of which a single frame is kept; alone, with slots, across devices
                       ("batch3", ["--frames-per-launch=3"]), ("batch2_slots3", ["--frames-per-launch=2", "--frames-in-flight=3"]),
                       ("batch3_strips", ["--fake-devices=4", "--frames-per-launch=3", "--frames-in-flight=2", "--distribution-strategy=shuffled-strips"])):
        c = str(tmp_path / tag)
        subprocess.check_call(common + [f"--headless={c}"] + extra)
        for f in range(7):
            assert open(f"{a}{f}.raw", "rb").read() == open(f"{c}{f}.raw", "rb").read(), f"{tag}: frame {f} differs"


@pytest.mark.gpu
def test_cpp_random_multi_device_runs(tmp_path, scene_dump):
    """Seeded draws of what the C++ host can be asked for: frame sizes that are not multiples of anything, 1-8 fake devices, scanlines
    or shuffled strips, 1-5 frame slots, 1-4 frames per launch, uneven device workloads with zeros, 1-3 bounces, several frames: the files must be the bytes of
    the single-device, one-frame-at-a-time run.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns."""
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "17")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "5"))):
        W, H = int(rng.integers(1, 260)), int(rng.integers(1, 160))
        frames = int(rng.integers(1, 7))
        common = [CLI, scene_dump, f"--width={W}", f"--height={H}", f"--max-ray-depth={int(rng.integers(1, 4))}", f"--frames={frames}", "--filetype=raw"]
        devices = int(rng.integers(1, 9))
        extra = [f"--fake-devices={devices}", f"--frames-in-flight={int(rng.integers(1, 6))}", f"--frames-per-launch={int(rng.choice([1, 1, 2, 3, 4]))}",
                 "--distribution-strategy=" + str(rng.choice(["scanline", "shuffled-strips"]))]
        if devices > 1 and rng.uniform() < 0.5:
            w = rng.uniform(0.05, 1, devices) * (rng.uniform(0, 1, devices) > 0.15)
            if w.sum() > 0:
                # the last device is offered everything: set_device_workloads clamps a ratio to what is left (src/rt_renderer.cc:151), and
                # ratios that sum to less than one leave the rest of the frame to nobody there as here (a first version of this test
                # rounded them to four digits and found the last pixel black)
                extra.append("--device-workloads=" + ",".join([f"{x:.6f}" for x in (w / w.sum())[:-1]] + ["1"]))
        a, b = str(tmp_path / f"one{k}_"), str(tmp_path / f"many{k}_")
        subprocess.check_call(common + [f"--headless={a}"])
        r = subprocess.run(common + [f"--headless={b}"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, f"draw {k}: {' '.join(common[2:] + extra)}: {r.stderr[-400:]}"
        for f in range(frames):
            fa, fb = (f"{a}{f}.raw", f"{b}{f}.raw") if frames > 1 else (f"{a}.raw", f"{b}.raw")
            x, y = open(fa, "rb").read(), open(fb, "rb").read()
            assert len(x) == W * H * 16 and x == y, f"draw {k}: {' '.join(common[2:] + extra)}: frame {f} differs"
            os.remove(fa); os.remove(fb)


@pytest.mark.gpu
def test_cpp_set_device_workloads_resizes_the_shares(tmp_path, scene_dump):
    """rt_renderer::set_device_workloads (src/rt_renderer.cc:135-183) with shuffled strips: shares that grow past the even
    split (the non-primary targets are allocated with get_distribution_target_max_size) still give the single-device frame."""
    common = [CLI, scene_dump, "--width=96", "--height=64", "--max-ray-depth=3", "--frames=3", "--filetype=raw"]
    a = str(tmp_path / "one")
    subprocess.check_call(common + [f"--headless={a}"])
    for tag, w in (("grow_last", "0.1,0.2,0.7"), ("grow_mid", "0.05,0.9,0.05"), ("starve", "0.5,0.0,0.5")):
        b = str(tmp_path / tag)
        subprocess.check_call(common + [f"--headless={b}", "--fake-devices=3", "--distribution-strategy=shuffled-strips", f"--device-workloads={w}"])
        for f in range(3):
            assert open(f"{a}{f}.raw", "rb").read() == open(f"{b}{f}.raw", "rb").read(), f"{tag}: frame {f} differs"
    r = subprocess.run(common + [f"--headless={a}", "--fake-devices=2", "--device-workloads=1"], capture_output=True, text=True)
    assert r.returncode != 0 and "one ratio per device" in r.stderr
    # a non-display device with the whole frame of a size whose strips are padded (45 x 51 = 2 295 pixels, 16 regions of 144 = 2 304 ids):
    # its partial image is a row taller than the frame (get_distribution_target_max_size in include/tauray_hip.hh)
    small = [CLI, scene_dump, "--width=45", "--height=51", "--max-ray-depth=3", "--frames=2", "--filetype=raw"]
    c, d = str(tmp_path / "small_one"), str(tmp_path / "small_all_on_1")
    subprocess.check_call(small + [f"--headless={c}"])
    subprocess.check_call(small + [f"--headless={d}", "--fake-devices=2", "--frames-in-flight=2", "--distribution-strategy=shuffled-strips", "--device-workloads=0,1"])
    for f in range(2):
        assert open(f"{c}{f}.raw", "rb").read() == open(f"{d}{f}.raw", "rb").read(), f"45x51, workloads 0,1: frame {f} differs"
