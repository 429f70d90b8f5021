#!/usr/bin/env python3
"""Writes tests/golden/skinned.glb: a small scene with one skinned mesh, made for the loader / skinning tests (no such
fixture exists in the reference).  A tube along +y (radius 0.25, height 2) bound to a three-joint chain whose *rest pose in
the file is bent* (the bind pose is straight: inverse bind matrices translate by -y of the joint), a floor, a point light
and a camera.  Some weight vectors are deliberately not normalised and the mesh node carries a transform, both of which
the loader has to undo (src/gltf.cc:722-731, 777-784).  The parameters below are what the tests recompute analytically."""
import json
import math
import os
import struct

import numpy as np

SEGMENTS, RINGS, RADIUS, HEIGHT = 24, 17, 0.25, 2.0
JOINT_Y = (0.0, 1.0, 2.0)                       # bind pose: joints on the axis
REST_Z_ROTATION_DEG = (0.0, 30.0, 40.0)         # rest pose: each joint turned about z relative to its parent
ROOT_TRANSLATION = (0.3, 0.0, -0.2)             # the skeleton's root node


def quat_z(deg):
    h = math.radians(deg) * 0.5
    return [0.0, 0.0, math.sin(h), math.cos(h)]


def build():
    pos, nrm, uv, joints, weights, idx = [], [], [], [], [], []
    for r in range(RINGS):
        y = HEIGHT * r / (RINGS - 1)
        t = min(y, 1.999)
        j0 = int(t)
        f = t - j0
        f = f * f * (3 - 2 * f)
        scale = 1.5 if r % 3 == 0 else 1.0      # weights that do not sum to one
        for s in range(SEGMENTS):
            a = 2 * math.pi * s / SEGMENTS
            pos.append((RADIUS * math.cos(a), y, RADIUS * math.sin(a)))
            nrm.append((math.cos(a), 0.0, math.sin(a)))
            uv.append((s / SEGMENTS, r / (RINGS - 1)))
            joints.append((j0, j0 + 1, 0, 0))
            weights.append(((1 - f) * scale, f * scale, 0.0, 0.0))
    for r in range(RINGS - 1):
        for s in range(SEGMENTS):
            a, b = r * SEGMENTS + s, r * SEGMENTS + (s + 1) % SEGMENTS
            c, d = a + SEGMENTS, b + SEGMENTS
            idx += [a, c, b, b, c, d]
    floor_pos = [(-3, -0.02, -3), (3, -0.02, -3), (3, -0.02, 3), (-3, -0.02, 3)]
    floor_nrm = [(0, 1, 0)] * 4
    floor_uv = [(0, 0), (1, 0), (1, 1), (0, 1)]
    floor_idx = [0, 2, 1, 0, 3, 2]
    ibm = []
    for y in JOINT_Y:                            # column-major 4x4: translate(0, -y, 0)
        ibm.append([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, -y, 0, 1])

    blobs, views, accessors = [], [], []

    def add(arr, dtype, atype, target=None, minmax=False):
        a = np.asarray(arr, dtype=dtype)
        raw = a.tobytes()
        off = sum(len(b) for b in blobs)
        pad = (-len(raw)) % 4
        blobs.append(raw + b"\0" * pad)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if target:
            v["target"] = target
        views.append(v)
        comp = {np.dtype("<f4"): 5126, np.dtype("<u2"): 5123, np.dtype("u1"): 5121, np.dtype("<u4"): 5125}[a.dtype]
        acc = {"bufferView": len(views) - 1, "componentType": comp, "count": len(a) if a.ndim > 1 else a.size, "type": atype}
        if minmax:
            acc["min"] = a.min(axis=0).tolist()
            acc["max"] = a.max(axis=0).tolist()
        accessors.append(acc)
        return len(accessors) - 1

    a_pos = add(pos, "<f4", "VEC3", 34962, True)
    a_nrm = add(nrm, "<f4", "VEC3", 34962)
    a_uv = add(uv, "<f4", "VEC2", 34962)
    a_jnt = add(joints, "u1", "VEC4", 34962)
    a_wgt = add(weights, "<f4", "VEC4", 34962)
    a_idx = add(idx, "<u2", "SCALAR", 34963)
    a_fpos = add(floor_pos, "<f4", "VEC3", 34962, True)
    a_fnrm = add(floor_nrm, "<f4", "VEC3", 34962)
    a_fuv = add(floor_uv, "<f4", "VEC2", 34962)
    a_fidx = add(floor_idx, "<u2", "SCALAR", 34963)
    a_ibm = add(ibm, "<f4", "MAT4")

    doc = {
        "asset": {"version": "2.0", "generator": "tools/make_skinned_glb.py"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1.0, 0.95, 0.9], "intensity": 400.0}]}},
        "scene": 0,
        "scenes": [{"nodes": [0, 1, 5, 6, 7]}],
        "nodes": [
            {"name": "skeleton", "translation": list(ROOT_TRANSLATION), "children": [2]},
            {"name": "tube", "mesh": 0, "skin": 0, "translation": [5.0, 5.0, 5.0]},          # must be ignored
            {"name": "joint0", "translation": [0, JOINT_Y[0], 0], "rotation": quat_z(REST_Z_ROTATION_DEG[0]), "children": [3]},
            {"name": "joint1", "translation": [0, JOINT_Y[1] - JOINT_Y[0], 0], "rotation": quat_z(REST_Z_ROTATION_DEG[1]), "children": [4]},
            {"name": "joint2", "translation": [0, JOINT_Y[2] - JOINT_Y[1], 0], "rotation": quat_z(REST_Z_ROTATION_DEG[2])},
            {"name": "floor", "mesh": 1},
            {"name": "lamp", "translation": [1.5, 3.0, 2.0], "extensions": {"KHR_lights_punctual": {"light": 0}}},
            {"name": "camera", "camera": 0, "translation": [0.0, 1.2, 5.0]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.1, "zfar": 100.0, "aspectRatio": 1.0}}],
        "skins": [{"joints": [2, 3, 4], "skeleton": 0, "inverseBindMatrices": a_ibm}],
        "materials": [
            {"name": "tube", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.3, 0.2, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.6}, "doubleSided": True},
            {"name": "floor", "pbrMetallicRoughness": {"baseColorFactor": [0.6, 0.6, 0.65, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
        ],
        "meshes": [
            {"name": "tube", "primitives": [{"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv, "JOINTS_0": a_jnt, "WEIGHTS_0": a_wgt},
                                              "indices": a_idx, "material": 0}]},
            {"name": "floor", "primitives": [{"attributes": {"POSITION": a_fpos, "NORMAL": a_fnrm, "TEXCOORD_0": a_fuv}, "indices": a_fidx, "material": 1}]},
        ],
        "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": sum(len(b) for b in blobs)}],
    }
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    binary = b"".join(blobs)
    total = 12 + 8 + len(js) + 8 + len(binary)
    out = struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + binary
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "skinned.glb")
    with open(path, "wb") as f:
        f.write(out)
    print("wrote", os.path.normpath(path), len(out), "bytes")


if __name__ == "__main__":
    build()
