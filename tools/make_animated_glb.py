#!/usr/bin/env python3
"""Writes tests/golden/animated.glb: tests/golden/skinned.glb (tools/make_skinned_glb.py) plus animation clips, made for the
animation loader / playback tests (the reference ships no animated asset).  Added to the skinned scene:

  * a box with a child box, animated by the clip "move": LINEAR translation (4 keys), CUBICSPLINE rotation about y (3 keys with
    tangents), STEP scale (3 keys); the child has a LINEAR rotation of its own, so its world transform is a product of two clips;
  * the skeleton's joints 1 and 2 bend further in "move" (LINEAR rotation), so the tube is re-skinned every frame;
  * the camera dollies and the lamp (a point light) drifts in "move" (LINEAR translations);
  * a second clip "spin" that only turns the box (what `--animation=spin` must pick, and what must NOT play by fallback on
    nodes that have "move", since "move" sorts first).

Keys sit at times that are not multiples of a frame at 24 or 60 fps.  The tests recompute the expected transforms from the
tables below."""
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "tests", "golden", "skinned.glb")
DST = os.path.join(HERE, "..", "tests", "golden", "animated.glb")

BOX_T_TIMES = [0.0, 0.35, 0.8, 1.25]
BOX_T_VALUES = [(-1.2, 0.4, 0.6), (-0.6, 0.9, 0.3), (0.2, 0.5, 0.9), (1.0, 0.4, 0.2)]
BOX_R_TIMES = [0.0, 0.6, 1.25]
BOX_R_DEG = [0.0, 100.0, 220.0]                 # about y; CUBICSPLINE with the tangents below (per second, quaternion space)
BOX_S_TIMES = [0.0, 0.5, 1.0]
BOX_S_VALUES = [(0.5, 0.5, 0.5), (0.8, 0.4, 0.6), (0.5, 0.9, 0.5)]
CHILD_R_TIMES = [0.0, 1.25]
CHILD_R_DEG = [0.0, 170.0]                      # about x
JOINT_TIMES = [0.0, 0.7, 1.25]
JOINT1_DEG = [30.0, -20.0, 55.0]                # about z, on top of nothing: the clip replaces the rest rotation
JOINT2_DEG = [40.0, 70.0, -10.0]
CAM_TIMES = [0.0, 1.25]
CAM_VALUES = [(0.0, 1.2, 5.0), (0.8, 1.6, 4.0)]
LAMP_TIMES = [0.0, 0.6, 1.25]
LAMP_VALUES = [(1.5, 3.0, 2.0), (0.2, 2.4, 2.6), (-1.4, 3.2, 1.0)]
SPIN_TIMES = [0.0, 0.5]
SPIN_DEG = [0.0, 90.0]                          # about z


def quat(axis, deg):
    h = math.radians(deg) * 0.5
    s = math.sin(h)
    return [axis[0] * s, axis[1] * s, axis[2] * s, math.cos(h)]


def box_mesh():
    pos, nrm, idx = [], [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            n = [0.0, 0.0, 0.0]
            n[axis] = sign
            u, v = (axis + 1) % 3, (axis + 2) % 3
            base = len(pos)
            for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = [0.0, 0.0, 0.0]
                p[axis] = sign
                p[u], p[v] = a, b
                pos.append(p)
                nrm.append(n)
            idx += [base, base + 1, base + 2, base, base + 2, base + 3] if sign > 0 else [base, base + 2, base + 1, base, base + 3, base + 2]
    return pos, nrm, idx


def build():
    raw = open(SRC, "rb").read()
    jlen = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jlen])
    blen = struct.unpack_from("<I", raw, 20 + jlen)[0]
    binary = bytearray(raw[28 + jlen:28 + jlen + blen])

    def add(arr, dtype, atype, target=None, minmax=False):
        a = np.asarray(arr, dtype=dtype)
        data = a.tobytes()
        while len(binary) % 4:
            binary.append(0)
        view = {"buffer": 0, "byteOffset": len(binary), "byteLength": len(data)}
        if target:
            view["target"] = target
        binary.extend(data)
        doc["bufferViews"].append(view)
        comp = {np.dtype("<f4"): 5126, np.dtype("<u2"): 5123}[a.dtype]
        acc = {"bufferView": len(doc["bufferViews"]) - 1, "componentType": comp, "count": len(a) if a.ndim > 1 else a.size, "type": atype}
        if minmax:
            acc["min"] = a.min(axis=0).tolist()
            acc["max"] = a.max(axis=0).tolist()
        doc["accessors"].append(acc)
        return len(doc["accessors"]) - 1

    pos, nrm, idx = box_mesh()
    a_pos, a_nrm, a_idx = add(pos, "<f4", "VEC3", 34962, True), add(nrm, "<f4", "VEC3", 34962), add(idx, "<u2", "SCALAR", 34963)
    doc["materials"].append({"name": "box", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.5, 0.8, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.4}})
    doc["meshes"].append({"name": "box", "primitives": [{"attributes": {"POSITION": a_pos, "NORMAL": a_nrm}, "indices": a_idx, "material": len(doc["materials"]) - 1}]})
    box_mesh_index = len(doc["meshes"]) - 1
    n_box, n_child = len(doc["nodes"]), len(doc["nodes"]) + 1
    doc["nodes"].append({"name": "box", "mesh": box_mesh_index, "translation": list(BOX_T_VALUES[0]), "scale": list(BOX_S_VALUES[0]), "children": [n_child]})
    doc["nodes"].append({"name": "box child", "mesh": box_mesh_index, "translation": [0.0, 1.8, 0.0], "scale": [0.4, 0.4, 0.4]})
    doc["scenes"][0]["nodes"].append(n_box)
    joint1 = next(i for i, n in enumerate(doc["nodes"]) if n.get("name") == "joint1")
    joint2 = next(i for i, n in enumerate(doc["nodes"]) if n.get("name") == "joint2")
    camera = next(i for i, n in enumerate(doc["nodes"]) if "camera" in n)
    lamp = next(i for i, n in enumerate(doc["nodes"]) if n.get("name") == "lamp")

    samplers, channels = [], []

    def channel(node, path, times, values, atype, interpolation="LINEAR"):
        samplers.append({"input": add(times, "<f4", "SCALAR", minmax=True), "output": add(values, "<f4", atype), "interpolation": interpolation})
        channels.append({"sampler": len(samplers) - 1, "target": {"node": node, "path": path}})

    channel(n_box, "translation", BOX_T_TIMES, BOX_T_VALUES, "VEC3")
    # CUBICSPLINE output: in-tangent, value, out-tangent per key
    spline = []
    for k, deg in enumerate(BOX_R_DEG):
        q = quat((0, 1, 0), deg)
        tan = [0.0, 0.9 * (1 if k else 0.5), 0.0, -0.4]
        spline += [tan, q, [t * 0.7 for t in tan]]
    channel(n_box, "rotation", BOX_R_TIMES, spline, "VEC4", "CUBICSPLINE")
    channel(n_box, "scale", BOX_S_TIMES, BOX_S_VALUES, "VEC3", "STEP")
    channel(n_child, "rotation", CHILD_R_TIMES, [quat((1, 0, 0), d) for d in CHILD_R_DEG], "VEC4")
    channel(joint1, "rotation", JOINT_TIMES, [quat((0, 0, 1), d) for d in JOINT1_DEG], "VEC4")
    channel(joint2, "rotation", JOINT_TIMES, [quat((0, 0, 1), d) for d in JOINT2_DEG], "VEC4")
    channel(camera, "translation", CAM_TIMES, CAM_VALUES, "VEC3")
    channel(lamp, "translation", LAMP_TIMES, LAMP_VALUES, "VEC3")
    doc["animations"] = [{"name": "move", "samplers": samplers, "channels": channels}]
    samplers, channels = [], []
    channel(n_box, "rotation", SPIN_TIMES, [quat((0, 0, 1), d) for d in SPIN_DEG], "VEC4")
    doc["animations"].append({"name": "spin", "samplers": samplers, "channels": channels})

    doc["asset"]["generator"] = "tools/make_skinned_glb.py + tools/make_animated_glb.py"
    while len(binary) % 4:
        binary.append(0)
    doc["buffers"][0]["byteLength"] = len(binary)
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    total = 12 + 8 + len(js) + 8 + len(binary)
    out = struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + bytes(binary)
    with open(DST, "wb") as f:
        f.write(out)
    print("wrote", os.path.normpath(DST), len(out), "bytes")


if __name__ == "__main__":
    build()
