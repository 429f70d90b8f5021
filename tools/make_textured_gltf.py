#!/usr/bin/env python3
"""Writes tests/golden/textured/: one small scene in the file forms real assets come in, for the loader tests (the reference
ships only test.glb, whose single texture is an embedded 8-bit PNG):
  room.gltf + room.bin + albedo.jpg + rough%20metal.png + normal16.png    text glTF, external buffer, external images (a
                                                                            baseline 4:2:0 JPEG, an interlaced palette PNG behind a
                                                                            percent-encoded name, a 16-bit RGB PNG)
  room_embedded.glb                                                        the same scene as a .glb with the three files embedded
  room_datauri.gltf                                                        the same with buffer and images as data: URIs
Geometry: a floor, a back wall and a box carrying the textures, a point light, a camera.  Needs Pillow (a tool, not a dependency of
the package).  Deterministic: the files are committed, the tests only read them."""
import base64
import io
import json
import os
import struct
import zlib

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "textured")


def quad(p0, eu, ev, uv_scale=1.0):
    p0, eu, ev = (np.asarray(x, np.float32) for x in (p0, eu, ev))
    pos = np.stack([p0, p0 + eu, p0 + eu + ev, p0 + ev])
    n = np.cross(eu, ev); n /= np.linalg.norm(n)
    uv = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], np.float32) * uv_scale
    return pos, np.tile(n, (4, 1)).astype(np.float32), uv, np.array([0, 1, 2, 0, 2, 3], np.uint16)


def box(c, s):
    c, s = np.asarray(c, np.float32), np.asarray(s, np.float32)
    P, N, U, I = [], [], [], []
    for axis in range(3):
        for sign in (-1, 1):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            e = np.eye(3, dtype=np.float32)
            eu, ev = e[u] * s[u] * 2, e[v] * s[v] * 2
            if sign < 0:
                eu, ev = ev, eu
            p0 = c + e[axis] * s[axis] * sign - eu / 2 - ev / 2
            p, n, t, i = quad(p0, eu, ev)
            I.append(i + 4 * len(P)); P.append(p); N.append(n); U.append(t)
    return np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I).astype(np.uint16)


def png_bytes(raw_rows, w, h, depth, ctype, interlace=0, extra=b""):
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    raw = b"".join(b"\x00" + r for r in raw_rows)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")


def images():
    rng = np.random.default_rng(11)
    y, x = np.mgrid[0:48, 0:64]
    albedo = np.clip(np.stack([150 + 80 * np.sin(x / 6.0), 120 + 90 * np.cos(y / 5.0), 90 + 60 * np.sin((x + y) / 9.0)], -1) + rng.normal(0, 4, (48, 64, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(albedo, "RGB").save(b, "JPEG", quality=88, subsampling=2); jpg = b.getvalue()
    # metallic-roughness: an interlaced palette PNG (G = roughness, B = metallic), written by hand (Adam7, filter 0)
    w, h = 20, 12
    idx = ((x[:h, :w] // 4 + y[:h, :w] // 3) % 4).astype(np.uint8)
    pal = bytes([0, 230, 0, 0, 120, 255, 0, 60, 40, 0, 180, 200])
    rows = []
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = idx[y0::dy, x0::dx]
        rows += [sub[r].tobytes() for r in range(sub.shape[0])] if sub.size else []
    body = b"PLTE" + pal
    mr = png_bytes(rows, w, h, 8, 3, 1, struct.pack(">I", len(pal)) + body + struct.pack(">I", zlib.crc32(body)))
    # normal map: 16 bits per sample RGB
    nx, ny = 0.25 * np.sin(x[:32, :32] / 3.0), 0.25 * np.cos(y[:32, :32] / 4.0)
    nz = np.sqrt(1 - nx * nx - ny * ny)
    n16 = np.clip((np.stack([nx, ny, nz], -1) * 0.5 + 0.5) * 65535 + 0.5, 0, 65535).astype(">u2")
    nrm = png_bytes([n16[r].tobytes() for r in range(32)], 32, 32, 16, 2)
    return jpg, mr, nrm


def build():
    os.makedirs(OUT, exist_ok=True)
    jpg, mr, nrm = images()
    meshes = [quad((-3, 0, -3), (0, 0, 6), (6, 0, 0), 3.0), quad((-3, 0, -3), (6, 0, 0), (0, 4, 0), 2.0), box((0.2, 0.8, 0.0), (0.8, 0.8, 0.8))]
    blobs, views, accessors = [], [], []

    def add(a, atype, target=None, minmax=False):
        raw = a.tobytes()
        off = sum(len(b) for b in blobs)
        blobs.append(raw + b"\0" * ((-len(raw)) % 4))
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(raw), **({"target": target} if target else {})})
        acc = {"bufferView": len(views) - 1, "componentType": 5126 if a.dtype == np.float32 else 5123, "count": len(a), "type": atype}
        if minmax:
            acc["min"], acc["max"] = a.min(0).tolist(), a.max(0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    prims = []
    for (p, n, t, i) in meshes:
        prims.append({"attributes": {"POSITION": add(p, "VEC3", 34962, True), "NORMAL": add(n, "VEC3", 34962), "TEXCOORD_0": add(t, "VEC2", 34962)},
                      "indices": add(i, "SCALAR", 34963)})
    mats = [
        {"name": "floor", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.8}},
        {"name": "wall", "pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.7, 0.75, 1.0], "metallicFactor": 1.0, "roughnessFactor": 1.0, "metallicRoughnessTexture": {"index": 1}}},
        {"name": "box", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.5}, "normalTexture": {"index": 2}},
    ]
    doc = {
        "asset": {"version": "2.0", "generator": "tools/make_textured_gltf.py"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1.0, 0.96, 0.9], "intensity": 900.0}]}},
        "scene": 0, "scenes": [{"nodes": [0, 1, 2, 3, 4]}],
        "nodes": [{"name": "floor", "mesh": 0}, {"name": "wall", "mesh": 1}, {"name": "box", "mesh": 2, "rotation": [0, 0.2588190451, 0, 0.9659258263]},
                  {"name": "lamp", "translation": [1.5, 3.2, 2.5], "extensions": {"KHR_lights_punctual": {"light": 0}}},
                  {"name": "camera", "camera": 0, "translation": [0.3, 1.6, 5.5], "rotation": [-0.0871557427, 0, 0, 0.9961946981]}],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1, "zfar": 100.0, "aspectRatio": 1.0}}],
        "meshes": [{"name": n, "primitives": [dict(prims[k], material=k)]} for k, n in enumerate(("floor", "wall", "box"))],
        "materials": mats,
        "textures": [{"source": 0}, {"source": 1}, {"source": 2}],
        "accessors": accessors, "bufferViews": views,
    }
    binary = b"".join(blobs)
    files = {"albedo.jpg": jpg, "rough metal.png": mr, "normal16.png": nrm, "room.bin": binary}
    for n, d in files.items():
        open(os.path.join(OUT, n), "wb").write(d)
    ext = dict(doc, buffers=[{"byteLength": len(binary), "uri": "room.bin"}],
               images=[{"uri": "albedo.jpg"}, {"uri": "rough%20metal.png"}, {"uri": "normal16.png"}])
    open(os.path.join(OUT, "room.gltf"), "w").write(json.dumps(ext, indent=1))
    b64 = lambda d: base64.b64encode(d).decode()
    data = dict(doc, buffers=[{"byteLength": len(binary), "uri": "data:application/octet-stream;base64," + b64(binary)}],
                images=[{"uri": "data:image/jpeg;base64," + b64(jpg)}, {"uri": "data:image/png;base64," + b64(mr)}, {"uri": "data:image/png;base64," + b64(nrm)}])
    open(os.path.join(OUT, "room_datauri.gltf"), "w").write(json.dumps(data, separators=(",", ":")))
    # .glb: the images move into the binary chunk
    gviews, gblobs = list(views), list(blobs)
    imgs = []
    for d, mime in ((jpg, "image/jpeg"), (mr, "image/png"), (nrm, "image/png")):
        off = sum(len(b) for b in gblobs)
        gblobs.append(d + b"\0" * ((-len(d)) % 4))
        gviews.append({"buffer": 0, "byteOffset": off, "byteLength": len(d)})
        imgs.append({"bufferView": len(gviews) - 1, "mimeType": mime})
    gbin = b"".join(gblobs)
    g = dict(doc, bufferViews=gviews, buffers=[{"byteLength": len(gbin)}], images=imgs)
    js = json.dumps(g, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    total = 12 + 8 + len(js) + 8 + len(gbin)
    open(os.path.join(OUT, "room_embedded.glb"), "wb").write(
        struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(gbin), 0x004E4942) + gbin)
    for n in sorted(os.listdir(OUT)):
        print(n, os.path.getsize(os.path.join(OUT, n)))


if __name__ == "__main__":
    build()
