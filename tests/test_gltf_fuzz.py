"""Differential test of the two glTF loaders - include/tauray_gltf.hh behind `tauray_hip --dump-scene` and tauray_amd/gltf.py - on
files nobody exported: seeded random .glb files (node trees with TRS or matrix transforms, non-uniform and mirrored scales, meshes of
several primitives, with and without normals / tangents / texture coordinates / indices, 8-, 16- and 32-bit indices, interleaved
buffer views with byte strides, shared accessors, materials with every factor and extension the reference reads
(src/gltf.cc:186-290: KHR_materials_transmission / ior / emissive_strength, TR_data), embedded PNG textures, KHR_lights_punctual
lights of all three kinds, perspective and orthographic cameras).  Both loaders must flatten a file to the same scene, byte for
byte (camera block: to 1e-6, the two 4x4 inverses round differently), or refuse it both.  No GPU involved.
TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tauray_amd", "tauray_hip")
NAMES = ["instances", "spans", "vertices", "indices", "point_lights", "directional_lights", "texture_infos", "texels", "envmap", "alias_table",
         "cameras", "non_opaque"]


def _png(rgba):
    """RGBA, 8 bits per channel for a uint8 array, 16 (big-endian samples, kept as RGBA16 texels by both loaders) for uint16."""
    h, w, _ = rgba.shape
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    depth = 16 if rgba.dtype == np.uint16 else 8
    rows = rgba.astype(">u2") if depth == 16 else rgba
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


class _Builder:
    def __init__(self, rng):
        self.rng = rng
        self.bin = bytearray()
        self.views, self.accessors = [], []

    def view(self, data: bytes, stride=None, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, view, ctype, count, typ, offset=0, minmax=None):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": typ}
        if offset:
            a["byteOffset"] = offset
        if minmax is not None:
            a["min"], a["max"] = [float(x) for x in minmax[0]], [float(x) for x in minmax[1]]
        self.accessors.append(a)
        return len(self.accessors) - 1


def _random_glb(rng):
    b = _Builder(rng)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    images, textures, materials, meshes, nodes, cameras, lights = [], [], [], [], [], [], []
    for _ in range(int(rng.integers(0, 3))):
        w, h = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        if rng.uniform() < 0.35:      # a 16-bit image: RGBA16 texels next to the RGBA8 ones of the same file
            px = rng.integers(0, 65536, (h, w, 4)).astype(np.uint16)
            if rng.uniform() < 0.5:
                px[..., 3] = 65535
        else:
            px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            if rng.uniform() < 0.5:
                px[..., 3] = 255
        images.append({"bufferView": b.view(_png(px)), "mimeType": "image/png"})
        textures.append({"source": len(images) - 1})
    for _ in range(int(rng.integers(1, 4))):
        pbr = {}
        if rng.uniform() < 0.8:
            pbr["baseColorFactor"] = [float(x) for x in rng.uniform(0, 1, 3)] + [float(rng.choice([1.0, 1.0, rng.uniform(0.1, 0.9)]))]
        if rng.uniform() < 0.7:
            pbr["metallicFactor"] = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        if rng.uniform() < 0.7:
            pbr["roughnessFactor"] = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        if textures and rng.uniform() < 0.5:
            pbr["baseColorTexture"] = {"index": int(rng.integers(0, len(textures)))}
        if textures and rng.uniform() < 0.3:
            pbr["metallicRoughnessTexture"] = {"index": int(rng.integers(0, len(textures)))}
        m = {"pbrMetallicRoughness": pbr}
        if rng.uniform() < 0.4:
            m["emissiveFactor"] = [float(x) for x in rng.uniform(0, 1, 3)]
        if textures and rng.uniform() < 0.2:
            m["normalTexture"] = {"index": int(rng.integers(0, len(textures))), "scale": float(rng.uniform(0.5, 1.5))}
        if textures and rng.uniform() < 0.2:
            m["emissiveTexture"] = {"index": int(rng.integers(0, len(textures)))}
        if rng.uniform() < 0.5:
            m["doubleSided"] = bool(rng.integers(0, 2))
        ext = {}
        if rng.uniform() < 0.3:
            ext["KHR_materials_transmission"] = {"transmissionFactor": float(rng.uniform(0, 1))}
        if rng.uniform() < 0.3:
            ext["KHR_materials_ior"] = {"ior": float(rng.uniform(1.0, 2.5))}
        if rng.uniform() < 0.3:
            ext["KHR_materials_emissive_strength"] = {"emissiveStrength": float(rng.uniform(0.5, 20))}
        if ext:
            m["extensions"] = ext
        materials.append(m)
    for _ in range(int(rng.integers(1, 4))):
        prims = []
        for _ in range(int(rng.integers(1, 3))):
            nv = int(rng.integers(3, 40))
            pos = f32(rng.normal(size=(nv, 3)) * rng.choice([0.1, 1.0, 5.0]))
            attrs = {}
            have_n, have_uv = rng.uniform() < 0.6, rng.uniform() < 0.6
            nrm = rng.normal(size=(nv, 3)); nrm = f32(nrm / np.linalg.norm(nrm, axis=1, keepdims=True))
            if have_n and rng.uniform() < 0.5:      # interleaved position + normal, one view with a byte stride
                inter = np.concatenate([pos, nrm], axis=1).astype(np.float32)
                pad = int(rng.choice([0, 8]))
                rec = b"".join(inter[i].tobytes() + b"\0" * pad for i in range(nv))
                v = b.view(rec, stride=24 + pad, target=34962)
                attrs["POSITION"] = b.accessor(v, 5126, nv, "VEC3", 0, (pos.min(0), pos.max(0)))
                attrs["NORMAL"] = b.accessor(v, 5126, nv, "VEC3", 12)
            else:
                attrs["POSITION"] = b.accessor(b.view(pos.tobytes(), target=34962), 5126, nv, "VEC3", 0, (pos.min(0), pos.max(0)))
                if have_n:
                    attrs["NORMAL"] = b.accessor(b.view(nrm.tobytes()), 5126, nv, "VEC3")
            if have_uv:
                attrs["TEXCOORD_0"] = b.accessor(b.view(f32(rng.uniform(-1, 2, (nv, 2))).tobytes()), 5126, nv, "VEC2")
            if have_n and have_uv and rng.uniform() < 0.5:
                t = rng.normal(size=(nv, 3)); t = t / np.linalg.norm(t, axis=1, keepdims=True)
                tan = f32(np.concatenate([t, rng.choice([-1.0, 1.0], (nv, 1))], axis=1))
                attrs["TANGENT"] = b.accessor(b.view(tan.tobytes()), 5126, nv, "VEC4")
            p = {"attributes": attrs}
            if rng.uniform() < 0.8:
                nt = int(rng.integers(1, 30))
                idx = rng.integers(0, nv, 3 * nt)
                kind = int(rng.choice([5121, 5123, 5125])) if nv < 256 else int(rng.choice([5123, 5125]))
                dt = {5121: np.uint8, 5123: np.uint16, 5125: np.uint32}[kind]
                p["indices"] = b.accessor(b.view(idx.astype(dt).tobytes(), target=34963), kind, 3 * nt, "SCALAR")
            elif nv % 3:
                continue            # a non-indexed primitive needs whole triangles
            if rng.uniform() < 0.85:
                p["material"] = int(rng.integers(0, len(materials)))
            prims.append(p)
        if prims:
            meshes.append({"primitives": prims})
    if not meshes:
        pos = f32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
        meshes.append({"primitives": [{"attributes": {"POSITION": b.accessor(b.view(pos.tobytes()), 5126, 3, "VEC3", 0, (pos.min(0), pos.max(0)))}}]})
    for _ in range(int(rng.integers(1, 3))):
        if rng.uniform() < 0.7:
            c = {"type": "perspective", "perspective": {"yfov": float(rng.uniform(0.3, 1.8)), "znear": float(rng.uniform(0.01, 0.5))}}
            if rng.uniform() < 0.5:
                c["perspective"]["aspectRatio"] = float(rng.uniform(0.5, 2.0))
            if rng.uniform() < 0.5:
                c["perspective"]["zfar"] = float(rng.uniform(50, 500))
        else:
            c = {"type": "orthographic", "orthographic": {"xmag": float(rng.uniform(0.5, 4)), "ymag": float(rng.uniform(0.5, 4)), "znear": 0.01, "zfar": float(rng.uniform(10, 100))}}
        cameras.append(c)
    for _ in range(int(rng.integers(0, 4))):
        kind = str(rng.choice(["point", "spot", "directional"]))
        li = {"type": kind, "color": [float(x) for x in rng.uniform(0, 1, 3)], "intensity": float(rng.uniform(0.5, 50))}
        if kind == "spot":
            li["spot"] = {"innerConeAngle": float(rng.uniform(0, 0.4)), "outerConeAngle": float(rng.uniform(0.45, 1.2))}
        if rng.uniform() < 0.3:
            li["name"] = f"light{len(lights)}"
        lights.append(li)

    def transform(n):
        r = rng.uniform()
        if r < 0.45:
            if rng.uniform() < 0.8:
                n["translation"] = [float(x) for x in rng.uniform(-3, 3, 3)]
            if rng.uniform() < 0.8:
                q = rng.normal(size=4); q /= np.linalg.norm(q)
                n["rotation"] = [float(x) for x in q]
            if rng.uniform() < 0.6:
                s = rng.uniform(0.3, 2.0, 3) * rng.choice([1.0, 1.0, 1.0, -1.0], 3)      # mirrored scales flip the winding
                n["scale"] = [float(x) for x in s]
        elif r < 0.75:
            m = np.eye(4)
            a = rng.normal(size=(3, 3)) * 0.4 + np.eye(3)
            m[:3, :3] = a; m[:3, 3] = rng.uniform(-2, 2, 3)
            n["matrix"] = [float(x) for x in m.T.reshape(-1)]      # column-major
        return n

    used_cam = False
    for k in range(int(rng.integers(2, 9))):
        n = transform({})
        r = rng.uniform()
        if r < 0.55:
            n["mesh"] = int(rng.integers(0, len(meshes)))
        elif r < 0.7 and lights:
            n["extensions"] = {"KHR_lights_punctual": {"light": int(rng.integers(0, len(lights)))}}
        elif r < 0.8 or (k > 2 and not used_cam):
            n["camera"] = int(rng.integers(0, len(cameras))); used_cam = True
        nodes.append(n)
    if not used_cam:
        nodes.append(transform({"camera": 0}))
    # a forest: every node but the first may hang under an earlier one
    roots = []
    for i in range(len(nodes)):
        if i > 0 and rng.uniform() < 0.5:
            nodes[int(rng.integers(0, i))].setdefault("children", []).append(i)
        else:
            roots.append(i)
    j = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes, "meshes": meshes, "materials": materials, "cameras": cameras,
         "accessors": b.accessors, "bufferViews": b.views, "buffers": [{"byteLength": len(b.bin)}]}
    if images:
        j["images"], j["textures"] = images, textures
    if lights:
        j["extensions"] = {"KHR_lights_punctual": {"lights": lights}}
        j["extensionsUsed"] = ["KHR_lights_punctual"]
    js = json.dumps(j).encode()
    js += b" " * (-len(js) % 4)
    bn = bytes(b.bin) + b"\0" * (-len(b.bin) % 4)
    return struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bn)) + struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(bn), b"BIN\0") + bn


def _sections(path):
    d = open(path, "rb").read()
    assert d[:4] == b"TRSC"
    pos, out = 8, {}
    for n in NAMES:
        size = struct.unpack("<Q", d[pos:pos + 8])[0]
        out[n] = d[pos + 8:pos + 8 + size]
        pos += 8 + size
    out["tail"] = d[pos:]
    return out


def test_random_glb_files_load_the_same_in_both_hosts(tmp_path):
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene_io import write_scene_dump
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "3")))
    loaded = refused = 0
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "25"))):
        glb = str(tmp_path / f"r{k}.glb")
        open(glb, "wb").write(_random_glb(rng))
        w, h = int(rng.integers(16, 300)), int(rng.integers(16, 200))
        cpp, py = str(tmp_path / "cpp.trsc"), str(tmp_path / "py.trsc")
        r = subprocess.run([CLI, glb, f"--width={w}", f"--height={h}", f"--dump-scene={cpp}"], capture_output=True, text=True)
        try:
            write_scene_dump(load_glb(glb, w, h), py)
            py_error = None
        except Exception as e:      # noqa: BLE001
            py_error = e
        if r.returncode != 0 or py_error is not None:
            assert r.returncode != 0 and py_error is not None, f"draw {k} ({glb}): one loader refuses the file, the other takes it: C++ '{r.stderr.strip()[-200:]}', Python '{py_error}'"
            refused += 1
            continue
        a, b = _sections(cpp), _sections(py)
        for n in NAMES + ["tail"]:
            if n == "cameras":
                x, y = np.frombuffer(a[n], np.float32), np.frombuffer(b[n], np.float32)
                assert x.shape == y.shape, f"draw {k}: cameras"
                scale = np.maximum(np.abs(y), 1.0)
                assert np.all((np.abs(x.astype(np.float64) - y) <= 1e-5 * scale) | (np.isnan(x) & np.isnan(y))), f"draw {k} ({glb}): cameras differ by {np.nanmax(np.abs(x - y))}"
            else:
                if a[n] != b[n]:
                    x, y = np.frombuffer(a[n], np.uint8), np.frombuffer(b[n], np.uint8)
                    where = int(np.nonzero(x[:min(len(x), len(y))] != y[:min(len(x), len(y))])[0][0]) if len(x) == len(y) else -1
                    raise AssertionError(f"draw {k} ({glb}): {n} differs (sizes {len(x)} / {len(y)}, first difference at byte {where})")
        os.remove(glb)
        loaded += 1
    assert loaded > 0.6 * (loaded + refused), f"only {loaded} of {loaded + refused} random files load"


def _mutate(rng, data):
    """One of the ways a file goes wrong: an accessor count multiplied, a bufferView moved past the buffer, the binary chunk cut short, an
    index accessor pointing at floats, node / mesh / bufferView references out of range (-1 included: Python would take the last element),
    vertex indices beyond the vertex count, a node that is its own child, a byteStride that overflows the extent arithmetic."""
    jl = struct.unpack("<I", data[12:16])[0]
    j = json.loads(data[20:20 + jl])
    bn = bytes(data[20 + jl + 8:])
    m = int(rng.integers(0, 9))
    if m == 8:      # a hostile byteStride: 2^63 wraps `stride * (count - 1)` to a small number in 64-bit arithmetic; 1 is below any element; 1e30 and -8 are no sizes
        v = j["bufferViews"][int(rng.integers(0, len(j["bufferViews"])))]
        v["byteStride"] = [2 ** 63, 2 ** 62, 1, 1e30, -8, 256][int(rng.integers(0, 6))]
        for a in j["accessors"]:      # make sure an accessor with several elements reads through it
            if j["bufferViews"][a["bufferView"]] is v and a["count"] > 2:
                a["count"] = 3
    elif m == 0:
        a = j["accessors"][int(rng.integers(0, len(j["accessors"])))]
        a["count"] = int(a["count"] * rng.choice([3, 50, 10000]))
    elif m == 1:
        v = j["bufferViews"][int(rng.integers(0, len(j["bufferViews"])))]
        v["byteOffset"] = int(v.get("byteOffset", 0) + rng.choice([len(bn), 1 << 30]))
    elif m == 2:
        bn = bn[:int(len(bn) * rng.uniform(0, 0.9))]
    elif m == 3:
        for me in j["meshes"]:
            for p in me["primitives"]:
                if "indices" in p:
                    p["indices"] = p["attributes"]["POSITION"]
    elif m == 4:
        j["nodes"][int(rng.integers(0, len(j["nodes"])))]["mesh"] = int(rng.choice([len(j["meshes"]) + 3, -1]))
    elif m == 5:
        j["accessors"][int(rng.integers(0, len(j["accessors"])))]["bufferView"] = int(rng.choice([len(j["bufferViews"]) + 2, -1]))
    elif m == 6:
        b = bytearray(bn)
        for me in j["meshes"]:
            for p in me["primitives"]:
                if "indices" in p:
                    a = j["accessors"][p["indices"]]
                    off = j["bufferViews"][a["bufferView"]].get("byteOffset", 0) + a.get("byteOffset", 0)
                    sz = {5121: 1, 5123: 2, 5125: 4}[a["componentType"]]
                    b[off:off + sz] = b"\xff" * sz
        bn = bytes(b)
    else:
        j["nodes"][0]["children"] = [0]
    js = json.dumps(j).encode()
    js += b" " * (-len(js) % 4)
    bn += b"\0" * (-len(bn) % 4)
    return struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(bn)) + struct.pack("<I4s", len(js), b"JSON") + js + struct.pack("<I4s", len(bn), b"BIN\0") + bn, m


def test_malformed_glb_files_are_refused_alike(tmp_path):
    """Mutated random files: both loaders must end with an error message or with the same scene - never with a crash (the first version of
    this test found the C++ loader recursing into a node cycle until the stack ran out, the Python one taking index -1 for the last mesh
    and copying out-of-range vertex indices into the scene)."""
    from tauray_amd.gltf import load_glb
    from tauray_amd.scene_io import write_scene_dump
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "5")))
    kinds = set()
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "40"))):
        data, kind = _mutate(rng, _random_glb(rng))
        glb = str(tmp_path / "m.glb")
        open(glb, "wb").write(data)
        cpp, py = str(tmp_path / "cpp.trsc"), str(tmp_path / "py.trsc")
        r = subprocess.run([CLI, glb, "--width=32", "--height=32", f"--dump-scene={cpp}"], capture_output=True, text=True, timeout=120)
        assert r.returncode in (0, 1), f"draw {k}, mutation {kind}: the C++ loader died with {r.returncode}: {r.stderr[-300:]}"
        assert r.returncode == 0 or r.stderr.strip(), f"draw {k}: an error without a message"
        try:
            write_scene_dump(load_glb(glb, 32, 32), py)
            py_ok = True
        except ValueError:
            py_ok = False
        except Exception as e:      # noqa: BLE001  (TrhipError of the image decoder and the like)
            py_ok = False
            assert type(e).__name__ != "RecursionError", f"draw {k}, mutation {kind}: {e}"
        assert (r.returncode == 0) == py_ok, f"draw {k}, mutation {kind}: C++ {'loads' if r.returncode == 0 else 'refuses (' + r.stderr.strip()[-120:] + ')'}, Python {'loads' if py_ok else 'refuses'}"
        if py_ok:
            a, b = _sections(cpp), _sections(py)
            for n in NAMES:
                if n != "cameras":
                    assert a[n] == b[n], f"draw {k}, mutation {kind}: {n} differs"
        kinds.add(kind)
    assert len(kinds) >= 5
    # a JSON chunk of nothing but brackets: a message from both, not the end of the C++ host's stack
    js = b'{"asset":{"version":"2.0"},"x":' + b"[" * 100000 + b"]" * 100000 + b"}"
    js += b" " * (-len(js) % 4)
    deep = str(tmp_path / "deep.glb")
    open(deep, "wb").write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js)) + struct.pack("<I4s", len(js), b"JSON") + js)
    r = subprocess.run([CLI, deep, f"--dump-scene={tmp_path / 'x.trsc'}"], capture_output=True, text=True)
    assert r.returncode == 1 and "nested too deeply" in r.stderr
    with pytest.raises(ValueError, match="nested too deeply"):
        load_glb(deep, 32, 32)
