"""glTF 2.0 reader (.glb containers, and .gltf text files with their buffers) -> SceneDesc.

Host mirror of the reference's loader for the subset the path tracer needs
(src/gltf.cc:199-280 materials, :330-505 nodes/lights/cameras, :510-798
meshes), followed by the instance flattening of scene_stage
(src/scene_stage.cc:664-819: one instance per (model, vertex group), in node
traversal order).  Supports KHR_lights_punctual, KHR_materials_transmission,
KHR_materials_ior, KHR_materials_emissive_strength, Tauray's TR_data and skins
(JOINTS_0 / WEIGHTS_0 + inverse bind matrices).  Textures: PNG of any colour type / bit depth / interlacing and baseline JPEG
(include/tauray_image.hh), embedded or behind a relative / data: uri.
"""
from __future__ import annotations

import json
import math
import struct
import zlib
from typing import List

import numpy as np

from . import animation as A
from . import scene as S

_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def decode_image(data: bytes) -> np.ndarray:
    """A texture file (PNG of any colour type / bit depth / interlacing, baseline, extended-sequential or progressive JPEG) -> HxWx4,
    row 0 = top row: what stb_image hands the reference's loader (src/gltf.cc:520-576) - uint8, or uint16 for a PNG of 16 bits per
    sample, which the reference keeps as R16G16B16A16Unorm.  One decoder for both hosts: include/tauray_image.hh through
    trhip_image_decode_texels, so the C++ loader flattens the same scene to the same bytes."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    w, h, ch, bits, p = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    _lib.check(L.trhip_image_decode_texels(bytes(data), len(data), C.byref(w), C.byref(h), C.byref(ch), C.byref(bits), C.byref(p)))
    if bits.value == 16:
        out = np.ctypeslib.as_array(p, (h.value, w.value, 8)).copy().view(np.uint16).reshape(h.value, w.value, 4)
    else:
        out = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    L.trhip_image_free(p)
    return out


decode_png = decode_image      # earlier name


def _read_uri(uri: str, base_dir: str) -> bytes:
    """A glTF `uri`: a data: URI (base64) or a file next to the scene file (percent-encoded relative path)."""
    import base64
    import os
    from urllib.parse import unquote
    if uri.startswith("data:"):
        head, _, payload = uri.partition(",")
        return base64.b64decode(payload) if head.endswith(";base64") else unquote(payload).encode("latin-1")
    with open(os.path.join(base_dir, unquote(uri)), "rb") as f:
        return f.read()


class _Glb:
    """The JSON document and the buffers of a .glb container (the only form the reference opens: LoadBinaryFromFile,
    src/gltf.cc:527) or of a .gltf text file with its external / data: buffers."""

    def __init__(self, path):
        import os
        self.dir = os.path.dirname(os.path.abspath(path))
        d = open(path, "rb").read()
        self.json = None
        glb_bin = None
        if d[:4] == b"glTF":
            off = 12
            while off < len(d):
                clen, ctype = struct.unpack("<II", d[off:off + 8])
                body = d[off + 8:off + 8 + clen]
                if ctype == 0x4E4F534A:
                    self.json = _loads(body)
                elif ctype == 0x004E4942:
                    glb_bin = body
                off += 8 + clen
            if self.json is None:
                raise ValueError("not a GLB file")
        else:
            try:
                self.json = _loads(d)
            except Exception:
                raise ValueError("not a GLB file") from None
        self.buffers = []
        for i, b in enumerate(self.json.get("buffers", [])):
            if "uri" in b:
                self.buffers.append(_read_uri(b["uri"], self.dir))
            elif i == 0 and glb_bin is not None:
                self.buffers.append(glb_bin)
            else:
                raise ValueError(f"glTF: buffer {i} has neither a uri nor a GLB chunk")
        self.bin = self.buffers[0] if self.buffers else b""

    def view(self, idx) -> bytes:
        bv = self.json["bufferViews"][idx]
        o = bv.get("byteOffset", 0)
        return self.buffers[bv.get("buffer", 0)][o:o + bv["byteLength"]]

    def image(self, img: dict) -> bytes:
        """The file behind an `images` entry: embedded (bufferView) or a uri (src/gltf.cc:532-576)."""
        if "bufferView" in img:
            return self.view(img["bufferView"])
        if "uri" in img:
            return _read_uri(img["uri"], self.dir)
        raise ValueError("glTF: image without bufferView or uri")

    def accessor(self, idx) -> np.ndarray:
        a = _item(self.json.get("accessors", []), idx, "accessor")
        bv = _item(self.json.get("bufferViews", []), a.get("bufferView", -1), "bufferView")
        if a["componentType"] not in _COMP or a["type"] not in _NCOMP:
            raise ValueError("glTF: unsupported accessor type")
        dt, sz = _COMP[a["componentType"]]
        nc = _NCOMP[a["type"]]
        base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or sz * nc
        count = a["count"]
        buf = _item(self.buffers, bv.get("buffer", 0), "buffer")
        if count < 0 or bv.get("byteOffset", 0) < 0 or a.get("byteOffset", 0) < 0 or bv.get("byteLength", 0) < 0 or bv.get("byteStride", 0) < 0:
            raise ValueError("glTF: negative count, offset or stride")
        # byteStride: 4 ... 252 and at least one element (glTF 2.0 section 5.11), 0 = tightly packed; as include/tauray_gltf.hh
        if bv.get("byteStride", 0) and not (sz * nc <= bv["byteStride"] <= 252):
            raise ValueError("glTF: byteStride out of range")
        # an accessor lives inside its bufferView, a bufferView inside its buffer (glTF 2.0 section 3.6.2)
        if bv.get("byteOffset", 0) + bv.get("byteLength", 0) > len(buf):
            raise ValueError("glTF: bufferView exceeds the buffer")
        if count and a.get("byteOffset", 0) + stride * (count - 1) + sz * nc > bv.get("byteLength", 0):
            raise ValueError("glTF: accessor exceeds its bufferView")
        if stride == sz * nc:
            arr = np.frombuffer(buf, dtype=dt, count=count * nc, offset=base).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(buf, dtype=dt, count=nc, offset=base + i * stride) for i in range(count)])
        return arr


def _loads(text):
    """json.loads with the loader's kind of error for text nested beyond what either host parses (the C++ host stops at 200 levels)."""
    try:
        return json.loads(text)
    except RecursionError:
        raise ValueError("glTF JSON: nested too deeply") from None


def _item(seq, index, what):
    """seq[index] for an index out of a file: integers inside the list only (Python would take -1 for the last element)."""
    if not isinstance(index, int) or isinstance(index, bool) or index < 0 or index >= len(seq):
        raise ValueError(f"glTF: {what} index out of range")
    return seq[index]


def _dot3(a, b):
    """glm::dot of vec3 rows in float32: (x * x' + y * y') + z * z', every step rounded."""
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1] + a[:, 2] * b[:, 2]


def _cross3(a, b):
    """glm::cross: (a.y b.z - b.y a.z, a.z b.x - b.z a.x, a.x b.y - b.x a.y) in float32."""
    return np.stack([a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2], a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0], a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]], axis=1).astype(np.float32)


def _hard_normals(pos, tri):
    d0 = pos[tri[:, 1]] - pos[tri[:, 0]]
    d1 = pos[tri[:, 2]] - pos[tri[:, 0]]
    hn = _cross3(d0, d1)
    ln = np.sqrt(_dot3(hn, hn))
    ok = ln > np.float32(1e-6)
    hn[ok] = hn[ok] / ln[ok, None]
    return d0, d1, hn


def _calculate_normals(vertices, indices):
    """mesh::calculate_normals (src/mesh.cc:113-143), operation by operation like include/tauray_gltf.hh: the hard normals are added
    triangle after triangle (np.add.at walks its index array in order: v0, v1, v2 of triangle 0, then triangle 1, ...), so a vertex's
    sum is rounded in the reference's order."""
    pos = vertices["pos"].astype(np.float32)
    tri = indices.reshape(-1, 3)
    _, _, hn = _hard_normals(pos, tri)
    n = np.zeros_like(pos)
    np.add.at(n, tri.reshape(-1), np.repeat(hn, 3, axis=0))
    ln = np.sqrt(_dot3(n, n))
    ok = ln > np.float32(1e-6)
    n[ok] = n[ok] / ln[ok, None]
    vertices["normal"] = n


def _normalize3(v):
    """glm::normalize: v * inversesqrt(dot(v, v)) with the reciprocal square root as 1 / sqrt in float32."""
    return v * (np.float32(1.0) / np.sqrt(_dot3(v, v)))[:, None]


def _calculate_tangents(vertices, indices):
    """mesh::calculate_tangents (src/mesh.cc:145-185); note only v0 accumulates."""
    pos = vertices["pos"].astype(np.float32)
    uv = vertices["uv"].astype(np.float32)
    nrm = vertices["normal"].astype(np.float32)
    tri = indices.reshape(-1, 3)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        d0, d1, hn = _hard_normals(pos, tri)
        uv0 = uv[tri[:, 1]] - uv[tri[:, 0]]
        uv1 = uv[tri[:, 2]] - uv[tri[:, 0]]
        ht = _normalize3(uv1[:, 1:2] * d0 - uv0[:, 1:2] * d1)
        hb = _normalize3(uv1[:, 0:1] * d1 - uv0[:, 0:1] * d0)
        sign = np.where(_dot3(_cross3(hn, ht), hb) < 0, -1.0, 1.0).astype(np.float32)
        t = np.zeros((len(pos), 4), dtype=np.float32)
        np.add.at(t, tri[:, 0], np.concatenate([ht, sign[:, None]], axis=1).astype(np.float32))
        t3 = _normalize3(t[:, :3] - nrm * _dot3(nrm, t[:, :3])[:, None])
    vertices["tangent"][:, :3] = t3
    vertices["tangent"][:, 3] = np.where(t[:, 3] < 0, -1.0, 1.0)


def _create_material(g: _Glb, mat: dict) -> np.ndarray:
    """create_material (src/gltf.cc:199-280)."""
    pbr = mat.get("pbrMetallicRoughness", {})

    def tex_source(info):
        if not info or info.get("index", -1) < 0:
            return -1
        return int(g.json["textures"][info["index"]]["source"])

    albedo = list(pbr.get("baseColorFactor", [1, 1, 1, 1]))
    emission = list(mat.get("emissiveFactor", [0, 0, 0]))
    transmittance = 0.0
    ior = 1.45
    ext = mat.get("extensions", {})
    discard_tr_emission = False
    if "KHR_materials_emissive_strength" in ext and "emissiveStrength" in ext["KHR_materials_emissive_strength"]:
        k = float(ext["KHR_materials_emissive_strength"]["emissiveStrength"])
        emission = [e * k for e in emission]
        discard_tr_emission = True
    tr = pbr.get("extensions", {}).get("TR_data")
    if tr is not None:
        if "transmission" in tr:
            transmittance = float(tr["transmission"])
        if "ior" in tr:
            ior = float(tr["ior"])
        if not discard_tr_emission and "emission" in tr:
            emission = [float(v) for v in tr["emission"][:3]]
    if "transmissionFactor" in ext.get("KHR_materials_transmission", {}):
        transmittance = float(ext["KHR_materials_transmission"]["transmissionFactor"])
    if "ior" in ext.get("KHR_materials_ior", {}):
        ior = float(ext["KHR_materials_ior"]["ior"])
    return S.make_material(
        albedo=albedo, metallic=pbr.get("metallicFactor", 1.0), roughness=pbr.get("roughnessFactor", 1.0),
        emission=emission, transmittance=transmittance, ior=ior, normal_factor=1.0,
        double_sided=bool(mat.get("doubleSided", False)),
        albedo_tex=tex_source(pbr.get("baseColorTexture")),
        mr_tex=tex_source(pbr.get("metallicRoughnessTexture")),
        normal_tex=tex_source(mat.get("normalTexture")),
        emission_tex=tex_source(mat.get("emissiveTexture")))


def load_glb(path: str, width: int = 512, height: int = 512, aspect_ratio: float = 0.0,
             force_single_sided: bool = False, force_double_sided: bool = False,
             gather_emissive_triangles: bool = True) -> S.SceneDesc:
    """load_gltf + scene flattening.  `width/height` drive set_camera_params
    (src/tauray.cc:68-110): the camera aspect is forced to width/height."""
    g = _Glb(path)
    j = g.json

    # The reference loads images flipped (stbi flag) and flips them back
    # (src/gltf.cc:525,557): net effect is row 0 = top row of the file.
    textures: List[np.ndarray] = []
    for img in j.get("images", []):
        textures.append(decode_image(g.image(img)))      # PNG or JPEG by signature, embedded or behind a uri

    # meshes -> list of vertex groups (material, vertices, indices)
    models = []
    for mesh in j.get("meshes", []):
        groups = []
        for p in mesh["primitives"]:
            if "material" in p:
                mat = _create_material(g, _item(j.get("materials", []), p["material"], "material"))
                if force_single_sided and mat["transmittance"] == 0:
                    mat["flags"] &= ~np.uint32(S.MATERIAL_FLAG_DOUBLE_SIDED)
                if force_double_sided:
                    mat["flags"] |= np.uint32(S.MATERIAL_FLAG_DOUBLE_SIDED)
            else:
                mat = S.make_material(albedo=(1, 1, 1, 1), metallic=0.0, roughness=1.0)
            at = p["attributes"]
            pos = g.accessor(at["POSITION"]).astype(np.float32)
            if pos.shape[1] < 3:
                raise ValueError("glTF: POSITION needs three components")
            v = np.zeros(len(pos), dtype=S.VERTEX)
            v["pos"] = pos[:, :3]

            def attribute(name, want):      # an attribute has a value for every vertex and enough components
                a = g.accessor(at[name])
                if len(a) < len(pos) or a.shape[1] < want:
                    raise ValueError(f"glTF: attribute {name} is shorter than POSITION")
                return a.astype(np.float32)[:len(pos), :want]
            if "NORMAL" in at:
                v["normal"] = attribute("NORMAL", 3)
            if "TEXCOORD_0" in at:
                v["uv"] = attribute("TEXCOORD_0", 2)
            if "TANGENT" in at:
                v["tangent"] = attribute("TANGENT", 4)
            if "indices" in p:
                if _item(j.get("accessors", []), p["indices"], "accessor").get("componentType") not in (5121, 5123, 5125):
                    raise ValueError("glTF: indices must be unsigned integers")      # glTF 2.0 section 3.7.2.1
                raw = g.accessor(p["indices"]).reshape(-1)
                idx = raw.astype(np.uint32)
                if len(raw) and (raw.min() < 0 or raw.max() >= len(pos)):
                    raise ValueError("glTF: index out of range")
            else:
                idx = np.arange(len(pos), dtype=np.uint32)
            if "NORMAL" not in at:
                _calculate_normals(v, idx)
            if "TANGENT" not in at:
                _calculate_tangents(v, idx)
            skin = None
            if "JOINTS_0" in at:        # mesh::skin_data, weights renormalised (src/gltf.cc:722-731)
                skin = np.zeros(len(pos), dtype=S.SKIN)
                skin["joints"] = g.accessor(at["JOINTS_0"]).astype(np.uint32)[:, :4]
                if "WEIGHTS_0" in at:
                    w = g.accessor(at["WEIGHTS_0"]).astype(np.float32)[:, :4]
                    ws = ((w[:, 0] + w[:, 1]) + w[:, 2]) + w[:, 3]
                    skin["weights"] = w / ws[:, None]
            groups.append((mat, v, idx, skin))
        models.append(groups)

    inst_list, span_list, vert_list, idx_list = [], [], [], []
    point_lights, spot_lights, dir_lights, cameras = [], [], [], []
    voff = ioff = 0
    light_meta = {"angle": 0.0, "radius": 0.0}
    node_globals, skinned_pending = {}, []
    nodes, roots = {}, []

    visited = set()

    def visit(node_index, parent, parent_index=-1):
        nonlocal voff, ioff
        node = _item(j.get("nodes", []), node_index, "node")
        if node_index in visited:
            raise ValueError(f"glTF: node {node_index} is reached twice: the node hierarchy must be a forest")
        visited.add(node_index)
        tr = node.get("extensions", {}).get("TR_data")
        if tr and "light" in tr:
            if "angle" in tr["light"]:
                light_meta["angle"] = float(tr["light"]["angle"])
            if "radius" in tr["light"]:
                light_meta["radius"] = float(tr["light"]["radius"])
        if "matrix" in node:
            local = np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T
            rec = A.Node(parent_index, list(node.get("children", [])), None, local)
        else:
            local = S.trs_matrix(node.get("translation", (0, 0, 0)), node.get("rotation", (0, 0, 0, 1)),
                                 node.get("scale", (1, 1, 1)))
            rec = A.Node(parent_index, list(node.get("children", [])),
                         {"translation": np.array(node.get("translation", (0, 0, 0)), dtype=np.float64),
                          "rotation": np.array(node.get("rotation", (0, 0, 0, 1)), dtype=np.float64),
                          "scale": np.array(node.get("scale", (1, 1, 1)), dtype=np.float64)}, None)
        nodes[node_index] = rec      # what SceneAnimator needs to move the node later (tauray_amd/animation.py)
        glob = parent @ local
        node_globals[node_index] = glob

        if "mesh" in node:
            sto = 0.0
            if tr and "mesh" in tr:
                sto = float(tr["mesh"].get("shadow_terminator_offset", 0.0))
            for mat, v, idx, skin in _item(models, node["mesh"], "mesh"):
                if "skin" in node and skin is not None:
                    # glTF places skinned meshes at the origin; the loader enforces it (src/gltf.cc:777-784)
                    skinned_pending.append((len(inst_list), node["skin"], skin))
                    inst_list.append(S.make_instance(np.eye(4), mat, sto))
                else:
                    rec.instances.append(len(inst_list))
                    inst_list.append(S.make_instance(glob, mat, sto))
                span_list.append((voff, len(v), ioff, len(idx) // 3))
                vert_list.append(v)
                idx_list.append(idx)
                voff += len(v)
                ioff += len(idx)

        if "camera" in node:
            c = _item(j.get("cameras", []), node["camera"], "camera")
            cam = S.Camera(transform=glob)
            if c["type"] == "perspective":
                pp = c["perspective"]
                cam.projection = S.PROJ_PERSPECTIVE
                cam.fov = math.degrees(pp["yfov"])
                cam.aspect = pp.get("aspectRatio", 1.0)
                cam.near = pp["znear"]
                cam.far = pp.get("zfar", math.inf)
            else:
                o = c["orthographic"]
                cam.projection = S.PROJ_ORTHOGRAPHIC
                cam.ortho = (-0.5 * o["xmag"], 0.5 * o["xmag"], -0.5 * o["ymag"], 0.5 * o["ymag"], o["znear"], o["zfar"])
            rec.cameras.append(len(cameras))
            cameras.append(cam)

        kl = node.get("extensions", {}).get("KHR_lights_punctual")
        if kl is not None:
            l = _item(j.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", []), kl["light"], "light")
            color = np.array(l.get("color", [1, 1, 1]), dtype=np.float64) * float(l.get("intensity", 1.0))
            angle, radius = light_meta["angle"], light_meta["radius"]

            def place(g):
                # get_global_direction: normalize(global orientation * (0,0,-1))
                rot = g[:3, :3] / np.linalg.norm(g[:3, :3], axis=0, keepdims=True)
                return rot @ np.array([0, 0, -1.0]), g[:3, 3]

            if l["type"] == "directional":
                make = lambda g: S.make_directional_light(color, place(g)[0], math.degrees(angle))
                rec.lights.append(("directional", len(dir_lights), make))
                dir_lights.append(make(glob))
            elif l["type"] == "point":
                make = lambda g: S.make_point_light(color / (4 * math.pi), place(g)[1], radius)
                rec.lights.append(("point", len(point_lights), make))
                point_lights.append(make(glob))
            elif l["type"] == "spot":
                outer = math.degrees(l["spot"].get("outerConeAngle", math.pi / 4))
                inner = math.degrees(l["spot"].get("innerConeAngle", 0.0))
                fall = S.spotlight_falloff_from_inner_angle(inner, outer, 4 / 255.0)
                make = lambda g: S.make_spotlight(color / (4 * math.pi), place(g)[1], place(g)[0], radius, outer, fall)
                rec.lights.append(("spot", len(spot_lights), make))
                spot_lights.append(make(glob))
        for ch in node.get("children", []):
            visit(ch, glob, node_index)

    for sc in j.get("scenes", []):
        for n in sc["nodes"]:
            roots.append(n)
            visit(n, np.eye(4))

    # animation clips per node (src/gltf.cc:580-627): channel -> the target node's pool entry of the clip's name
    animations = {}
    for anim in j.get("animations", []):
        for chan in anim["channels"]:
            target = chan["target"]
            if "node" not in target:
                continue
            sampler = anim["samplers"][chan["sampler"]]
            interp = A.INTERPOLATION.get(sampler.get("interpolation", "LINEAR"), A.LINEAR)
            path_name = target["path"]
            if path_name not in ("translation", "rotation", "scale"):
                continue            # morph-target weights
            clip = animations.setdefault(target["node"], {}).setdefault(anim.get("name", ""), A.Animation())
            track = A.read_track(g.accessor(sampler["input"]).astype(np.float32).reshape(-1),
                                 g.accessor(sampler["output"]).astype(np.float32).reshape(-1, 4 if path_name == "rotation" else 3), interp)
            setattr(clip, {"translation": "position", "rotation": "orientation", "scale": "scaling"}[path_name], track)

    aspect = aspect_ratio if aspect_ratio > 0 else width / float(height)
    for cam in cameras:
        cam.set_aspect(aspect)

    # point lights first, then spotlights (src/scene_stage.cc:1287-1317)
    pls = point_lights + spot_lights
    desc = S.SceneDesc(
        instances=np.concatenate(inst_list) if inst_list else np.zeros(0, dtype=S.INSTANCE),
        spans=np.array(span_list, dtype=S.MESH_SPAN),
        vertices=np.concatenate(vert_list) if vert_list else np.zeros(0, dtype=S.VERTEX),
        indices=np.concatenate(idx_list).astype(np.uint32) if idx_list else np.zeros(0, dtype=np.uint32),
        point_lights=np.concatenate(pls) if pls else np.zeros(0, dtype=S.POINT_LIGHT),
        directional_lights=np.concatenate(dir_lights) if dir_lights else np.zeros(0, dtype=S.DIRECTIONAL_LIGHT),
        textures=textures, envmap=None, environment_factor=(0, 0, 0, 0), cameras=cameras,
        name=path.split("/")[-1])
    desc.node_globals = node_globals
    desc.nodes, desc.roots, desc.animations = nodes, roots, animations
    desc.spotlight_base = len(point_lights)
    for inst, skin_index, skin in skinned_pending:
        sk = j["skins"][skin_index]
        joints = list(sk["joints"])
        if "inverseBindMatrices" in sk:
            ibm = g.accessor(sk["inverseBindMatrices"]).astype(np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)   # column-major in the file
        else:
            ibm = np.stack([np.eye(4)] * len(joints))
        desc.skinned.append(S.SkinnedMesh(instance=inst, skins=skin, joint_nodes=joints, inverse_bind=ibm))
    return desc.finalize(gather_emissive_triangles)
