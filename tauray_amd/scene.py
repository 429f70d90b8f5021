"""Scene data contract: the POD layouts the path tracer consumes.

Byte-for-byte mirrors of the reference's GPU-side structs (GLSL `scalar`
layout, tightly packed):

* ``VERTEX``             - src/mesh.hh:19-25, shader/scene.glsl:14-20        (48 B)
* ``MATERIAL``           - src/scene_stage.cc:17-30, shader/material.glsl:9-22 (80 B)
* ``INSTANCE``           - src/scene_stage.cc:32-44, shader/scene.glsl:43-53   (288 B)
* ``DIRECTIONAL_LIGHT``  - src/scene_stage.cc:46-62, shader/light.glsl:7-13    (32 B)
* ``POINT_LIGHT``        - src/scene_stage.cc:64-98, shader/light.glsl:15-27   (64 B)
* ``TRI_LIGHT``          - src/scene_stage.cc:103-111, shader/light.glsl:29-37 (64 B)
* ``ALIAS_ENTRY``        - src/environment_map.hh:37-43                        (16 B)
* ``CAMERA_DATA``        - src/camera.cc:397-407, shader/camera.glsl:13-23     (320 B)

Matrices are stored column-major like glm (``arr[col][row]``).  Host-side
packing (camera matrices, light unit handling, instance flattening) follows
src/camera.cc:323-478, src/scene_stage.cc:1066-1354 and src/light.cc.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

f4, i4, u4 = "<f4", "<i4", "<u4"

VERTEX = np.dtype([("pos", f4, 3), ("normal", f4, 3), ("uv", f4, 2), ("tangent", f4, 4)])
MATERIAL = np.dtype([
    ("albedo_factor", f4, 4), ("metallic_roughness_factor", f4, 4),
    ("emission_factor", f4, 4), ("transmittance", f4), ("ior", f4),
    ("normal_factor", f4), ("flags", u4), ("albedo_tex_id", i4),
    ("metallic_roughness_tex_id", i4), ("normal_tex_id", i4), ("emission_tex_id", i4)])
INSTANCE = np.dtype([
    ("light_base_id", i4), ("sh_grid_index", i4), ("pad", u4),
    ("shadow_terminator_mul", f4), ("model", f4, (4, 4)), ("model_normal", f4, (4, 4)),
    ("model_prev", f4, (4, 4)), ("mat", MATERIAL)])
DIRECTIONAL_LIGHT = np.dtype([("color", f4, 3), ("shadow_map_index", i4), ("dir", f4, 3), ("dir_cutoff", f4)])
POINT_LIGHT = np.dtype([
    ("color", f4, 3), ("dir", f4, 3), ("pos", f4, 3), ("radius", f4), ("dir_cutoff", f4),
    ("dir_falloff", f4), ("cutoff_radius", f4), ("spot_radius", f4),
    ("shadow_map_index", i4), ("padding", i4)])
TRI_LIGHT = np.dtype([
    ("pos", f4, (3, 3)), ("emission_factor", u4), ("instance_id", u4), ("primitive_id", u4),
    ("uv", u4, 3), ("emission_tex_id", i4)])
ALIAS_ENTRY = np.dtype([("alias_id", u4), ("probability", u4), ("pdf", f4), ("alias_pdf", f4)])
CAMERA_DATA = np.dtype([
    ("view", f4, (4, 4)), ("view_inverse", f4, (4, 4)), ("view_proj", f4, (4, 4)),
    ("proj_inverse", f4, (4, 4)), ("origin", f4, 4), ("dof_params", f4, 4),
    ("projection_info", f4, 4), ("pan", f4, 4)])
# Per-instance geometry span inside the concatenated vertex/index arrays.
MESH_SPAN = np.dtype([("vertex_offset", u4), ("vertex_count", u4), ("index_offset", u4), ("triangle_count", u4)])
SKIN = np.dtype([("joints", u4, 4), ("weights", f4, 4)])   # mesh::skin_data (src/mesh.hh:32-36)
# Texture table entry: texels concatenated in one byte array; texel_offset in 4-byte words, format 0 = RGBA8, 1 = RGBA16 (a 16-bit PNG,
# which the reference keeps as R16G16B16A16Unorm, src/gltf.cc:548-556)
TEXTURE_INFO = np.dtype([("width", u4), ("height", u4), ("texel_offset", u4), ("format", u4)])

assert VERTEX.itemsize == 48 and MATERIAL.itemsize == 80 and INSTANCE.itemsize == 288
assert DIRECTIONAL_LIGHT.itemsize == 32 and POINT_LIGHT.itemsize == 64 and TRI_LIGHT.itemsize == 64
assert ALIAS_ENTRY.itemsize == 16 and CAMERA_DATA.itemsize == 320

MATERIAL_FLAG_DOUBLE_SIDED = 1
MATERIAL_FLAG_TRANSIENT = 2

PROJ_PERSPECTIVE, PROJ_ORTHOGRAPHIC, PROJ_EQUIRECTANGULAR = 0, 1, 2


# ----------------------------------------------------------------------------
# small matrix helpers (mathematical row-major float64; stored transposed)
# ----------------------------------------------------------------------------
def quat_to_mat3(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def trs_matrix(translation=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1)):
    """transformable::get_transform (src/transformable.cc:203-212)."""
    m = np.eye(4)
    m[:3, :3] = quat_to_mat3(rotation) * np.asarray(scale, dtype=np.float64)[None, :]
    m[:3, 3] = translation
    return m


def to_glm(m) -> np.ndarray:
    """Mathematical 4x4 -> column-major float32 storage."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.astype(np.float32))


def from_glm(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float64).T


def perspective_matrix(fov_deg, aspect, near, far):
    """glm::perspective / glm::infinitePerspective (RH), src/camera.cc:363-381."""
    t = math.tan(math.radians(fov_deg) / 2.0)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[3, 2] = -1.0
    if math.isinf(far):
        m[2, 2] = -1.0
        m[2, 3] = -2.0 * near
    else:
        m[2, 2] = -(far + near) / (far - near)
        m[2, 3] = -(2.0 * far * near) / (far - near)
    return m


def ortho_matrix(l, r, b, t, n, f):
    m = np.eye(4)
    m[0, 0] = 2 / (r - l); m[1, 1] = 2 / (t - b); m[2, 2] = -2 / (f - n)
    m[0, 3] = -(r + l) / (r - l); m[1, 3] = -(t + b) / (t - b); m[2, 3] = -(f + n) / (f - n)
    return m


@dataclass
class Camera:
    """Host mirror of tr::camera + its transformable (src/camera.{hh,cc})."""
    transform: np.ndarray = field(default_factory=lambda: np.eye(4))  # global transform (= view_inverse)
    projection: int = PROJ_PERSPECTIVE
    fov: float = 90.0          # vertical fov, degrees
    aspect: float = 1.0
    near: float = 0.1
    far: float = 100.0
    fov_offset: tuple = (0.0, 0.0)          # pan
    focus: tuple = (1.0, 0.0, 0.0, 0.0)     # dof_params
    ortho: tuple = (-1, 1, -1, 1, 0, 100)   # l r b t n f
    equirect_fov: tuple = (360.0, 180.0)

    def set_aspect(self, aspect):
        # src/camera.cc:166-186
        if self.projection == PROJ_PERSPECTIVE:
            self.aspect = aspect
        elif self.projection == PROJ_ORTHOGRAPHIC:
            l, r, b, t, n, f = self.ortho
            yr = (r - l) / aspect
            yc = (b + t) * 0.5
            self.ortho = (l, r, yc - yr * 0.5, yc + yr * 0.5, n, f)

    def set_focus(self, f_stop, focus_distance, aperture_sides, aperture_angle, sensor_size):
        # src/camera.cc:144-164
        self.focus = (focus_distance, 0.0 if f_stop == 0 else sensor_size / f_stop,
                      math.radians(aperture_angle), float(aperture_sides))

    def hfov(self):
        return 2.0 * math.degrees(math.atan(self.aspect * math.tan(math.radians(self.fov * 0.5))))

    def projection_matrix(self):
        if self.projection == PROJ_PERSPECTIVE:
            p = perspective_matrix(self.fov, self.aspect, self.near, self.far)
            p[0, 2] = self.fov_offset[0]
            p[1, 2] = self.fov_offset[1]
            return p
        if self.projection == PROJ_ORTHOGRAPHIC:
            return ortho_matrix(*self.ortho)
        raise ValueError("no matrix for equirectangular cameras")

    def projection_info(self):
        # src/camera.cc:323-361
        if self.projection == PROJ_PERSPECTIVE:
            w = 2 * math.tan(math.radians(self.fov) / 2.0)
            z = w * self.aspect
            if math.isinf(self.far):
                return (-self.near, -1.0, z, w)
            n, f = self.near, self.far
            return (n * f / (n - f), (n + f) / (n - f), z, w)
        if self.projection == PROJ_ORTHOGRAPHIC:
            l, r, b, t, n, f = self.ortho
            return (f - n, -f, r - l, t - b)
        return (0.0, 0.0, 0.0, 0.0)

    def pack(self) -> np.ndarray:
        """camera::write_uniform_buffer (src/camera.cc:431-478) -> CAMERA_DATA[1]."""
        out = np.zeros(1, dtype=CAMERA_DATA)
        inv_view = np.asarray(self.transform, dtype=np.float64)
        view = np.linalg.inv(inv_view)
        out["view"][0] = to_glm(view)
        out["view_inverse"][0] = to_glm(inv_view)
        out["origin"][0] = (inv_view @ np.array([0, 0, 0, 1.0])).astype(np.float32)
        if self.projection == PROJ_EQUIRECTANGULAR:
            # equirectangular_camera_data_buffer (src/camera.cc:409-415): the first
            # 152 bytes of the slot are {view, view_inverse, origin, fov(half angles)}.
            raw = out.view(np.float32).reshape(-1)
            raw[32:36] = out["origin"][0]
            raw[36] = math.radians(self.equirect_fov[0]) * 0.5
            raw[37] = math.radians(self.equirect_fov[1]) * 0.5
            out["origin"][0] = 0
            return out
        proj = self.projection_matrix()
        out["view_proj"][0] = to_glm(proj @ view)
        out["proj_inverse"][0] = to_glm(np.linalg.inv(proj))
        out["dof_params"][0] = self.focus if self.projection == PROJ_PERSPECTIVE else (0, 0, 0, 0)
        out["projection_info"][0] = self.projection_info()
        out["pan"][0] = (self.fov_offset[0], self.fov_offset[1], 0, 0) if self.projection == PROJ_PERSPECTIVE else (0, 0, 0, 0)
        return out


def generate_camera_grid(cam: Camera, grid_w: int, grid_h: int, dx: float, dy: float,
                         recentering_distance: float = 5.0, roll_deg: float = 0.0,
                         offset=(0.0, 0.0, 0.0)) -> List[Camera]:
    """generate_cameras (src/tauray.cc:680-727): light-field camera grid."""
    import copy
    width = (grid_w - 1) * dx
    height = (grid_h - 1) * dy
    tfov = (math.tan(math.radians(cam.hfov()) * 0.5), math.tan(math.radians(cam.fov) * 0.5))
    c, s = math.cos(math.radians(roll_deg)), math.sin(math.radians(roll_deg))
    res = []
    for y in range(grid_h):
        for x in range(grid_w):
            gx, gy = -width * 0.5 + x * dx, height * 0.5 - y * dy
            gp = np.array([c * gx - s * gy, s * gx + c * gy, 0.0])
            sub = copy.deepcopy(cam)
            sub.fov_offset = (-gp[0] / (tfov[0] * recentering_distance),
                              -gp[1] / (tfov[1] * recentering_distance))
            local = np.eye(4)
            local[:3, 3] = gp + np.asarray(offset, dtype=np.float64)
            sub.transform = np.asarray(cam.transform) @ local
            res.append(sub)
    return res


def build_alias_table(envmap: np.ndarray) -> np.ndarray:
    """environment_map::generate_alias_table (src/environment_map.cc:39-140) with the
    importance of shader/alias_table_importance.comp:16-28 computed on the host."""
    h, w = envmap.shape[:2]
    n = w * h
    lum = (envmap[..., 0].astype(np.float32) * np.float32(0.2126)
           + envmap[..., 1].astype(np.float32) * np.float32(0.7152)
           + envmap[..., 2].astype(np.float32) * np.float32(0.0722))
    ys = np.arange(h, dtype=np.float32)
    y0 = ys / np.float32(h)
    y1 = (ys + 1) / np.float32(h)
    # cos / sin below: evaluated in double and rounded to float, so that the C++ host's table (include/tauray_envmap.hh, libm)
    # is this one bit for bit; the shader the reference runs here uses the GPU's float cos
    def cos32(x):
        return np.array([math.cos(float(v)) for v in x], dtype=np.float64).astype(np.float32)
    solid = (np.float32(2.0 * math.pi) * (cos32(np.float32(math.pi) * y0) - cos32(np.float32(math.pi) * y1))
             / np.float32(w)).astype(np.float32)
    importance = (lum * solid[:, None]).astype(np.float32).reshape(-1)
    total = float(np.cumsum(importance.astype(np.float64))[-1]) if n else 0.0      # one running double sum, in pixel order
    inv_average = np.float32(1.0 / (total / n)) if total > 0 else np.float32(0)
    importance = (importance * inv_average).astype(np.float32)

    table = np.zeros(n, dtype=ALIAS_ENTRY)
    table["alias_id"] = np.arange(n, dtype=np.uint32)
    table["probability"] = 0xFFFFFFFF
    imp = importance.astype(np.float64)  # float arithmetic below is done in float32 steps
    prob = table["probability"]
    alias = table["alias_id"]

    def ldexp32(v):
        r = math.ldexp(float(np.float32(v)), 32)
        return int(min(max(r, 0.0), 4294967295.0))

    i = j = 0
    while i < n and importance[i] > 1.0:
        i += 1
    while j < n and importance[j] <= 1.0:
        j += 1
    weight = np.float32(importance[j]) if j < n else np.float32(0)
    while j < n:
        if weight > 1.0:
            if i >= n:
                break
            prob[i] = ldexp32(importance[i])
            alias[i] = j
            weight = np.float32(np.float32(weight + importance[i]) - np.float32(1.0))
            i += 1
            while i < n and importance[i] > 1.0:
                i += 1
        else:
            prob[j] = ldexp32(weight)
            old_j = j
            j += 1
            while j < n and importance[j] <= 1.0:
                j += 1
            if j < n:
                alias[old_j] = j
                weight = np.float32(np.float32(weight + importance[j]) - np.float32(1.0))
    del imp
    sin_theta = np.array([math.sin(float(v)) for v in (np.arange(h, dtype=np.float32) + np.float32(0.5)) / np.float32(h) * np.float32(math.pi)],
                         dtype=np.float64).astype(np.float32)
    rows = np.arange(n) // w
    denom = (np.float32(2.0 * math.pi * math.pi) * sin_theta).astype(np.float32)
    table["pdf"] = importance / denom[rows]
    aj = table["alias_id"].astype(np.int64)
    table["alias_pdf"] = importance[aj] / denom[aj // w]
    return table


@dataclass
class SkinnedMesh:
    """A skinned vertex group of the scene (mesh::get_skin + model::get_joints, src/mesh.hh:32-36, src/model.hh): its
    instance, one SKIN record per vertex of that instance, and per joint the glTF node that drives it and the inverse
    bind matrix (mathematical 4x4)."""
    instance: int
    skins: np.ndarray            # SKIN[vertex_count]
    joint_nodes: List[int]
    inverse_bind: np.ndarray     # (n, 4, 4)


@dataclass
class SceneDesc:
    """Everything `trhip_scene_upload` takes: the flattened scene of
    scene_stage::update (src/scene_stage.cc:1026-1496)."""
    instances: np.ndarray                      # INSTANCE[n]
    spans: np.ndarray                          # MESH_SPAN[n]
    vertices: np.ndarray                       # VERTEX[...], model space, per-instance spans
    indices: np.ndarray                        # uint32[...], per-instance, relative to the span
    point_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=POINT_LIGHT))
    directional_lights: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=DIRECTIONAL_LIGHT))
    textures: List[np.ndarray] = field(default_factory=list)   # each HxWx4 uint8 (or uint16: RGBA16), row 0 first in memory
    envmap: Optional[np.ndarray] = None        # HxWx4 float32 lat-long, or None
    environment_factor: tuple = (0.0, 0.0, 0.0, 0.0)
    cameras: List[Camera] = field(default_factory=list)
    name: str = "scene"
    skinned: List["SkinnedMesh"] = field(default_factory=list)     # vertices of these instances are the bind pose
    node_globals: dict = field(default_factory=dict)               # glTF node index -> global transform of the rest pose
    nodes: dict = field(default_factory=dict)                      # glTF node index -> animation.Node (tree, local transform, instances, cameras)
    roots: list = field(default_factory=list)                      # root nodes of the file's scenes
    animations: dict = field(default_factory=dict)                 # glTF node index -> {clip name: animation.Animation}
    spotlight_base: int = 0                                        # index of the first spotlight in point_lights (point lights come first)

    def joint_transforms(self, sk: "SkinnedMesh", node_globals: Optional[dict] = None) -> np.ndarray:
        """model::update_joints (src/model.cc:107-118): joint node's global transform * inverse bind matrix, (n, 4, 4)."""
        g = node_globals if node_globals is not None else self.node_globals
        return np.stack([np.asarray(g[n], dtype=np.float64) @ sk.inverse_bind[i] for i, n in enumerate(sk.joint_nodes)]).astype(np.float32)

    @property
    def triangle_count(self) -> int:
        return int(self.spans["triangle_count"].sum())

    def finalize(self, gather_emissive_triangles: bool = True):
        """light_base_id assignment (src/scene_stage.cc:1069-1075)."""
        tri_light_count = 0
        for i in range(len(self.instances)):
            ef = self.instances["mat"]["emission_factor"][i][:3]
            if np.any(ef != 0):
                self.instances["light_base_id"][i] = tri_light_count
                tri_light_count += int(self.spans["triangle_count"][i])
            else:
                self.instances["light_base_id"][i] = -1
        self.tri_light_count = tri_light_count if gather_emissive_triangles else 0
        if not gather_emissive_triangles:
            self.instances["light_base_id"][:] = -1
        return self

    def texture_table(self):
        infos = np.zeros(len(self.textures), dtype=TEXTURE_INFO)
        off = 0
        chunks = []
        for i, t in enumerate(self.textures):
            wide = np.asarray(t).dtype == np.uint16
            t = np.ascontiguousarray(t, dtype=np.uint16 if wide else np.uint8)
            assert t.ndim == 3 and t.shape[2] == 4
            infos[i] = (t.shape[1], t.shape[0], off, 1 if wide else 0)
            off += t.shape[0] * t.shape[1] * (2 if wide else 1)
            chunks.append(t.reshape(-1).view(np.uint8))
        texels = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
        return infos, texels

    def camera_data(self) -> np.ndarray:
        if not self.cameras:
            return np.zeros(0, dtype=CAMERA_DATA)
        return np.concatenate([c.pack() for c in self.cameras])

    def has_tri_lights(self) -> bool:
        return bool(np.any(self.instances["mat"]["emission_factor"][:, :3] != 0))

    def potentially_transparent(self) -> np.ndarray:
        """material::potentially_transparent (src/material.cc:7-11) per instance:
        the BLAS geometry is flagged non-opaque for these (src/scene_stage.cc:851-860)."""
        mat = self.instances["mat"]
        res = (mat["transmittance"] > 0) | (mat["albedo_factor"][:, 3] < 1.0)
        for i, tid in enumerate(mat["albedo_tex_id"]):
            if tid >= 0 and not texture_is_opaque(self.textures[tid]):
                res[i] = True
        return res


def texture_is_opaque(tex: np.ndarray) -> bool:
    """check_opaque (src/gltf.cc:54-66): only an 8-bit image whose alpha is 255 everywhere counts as opaque."""
    return tex.dtype == np.uint8 and bool(np.all(tex[..., 3] == 255))


def make_instance(model: np.ndarray, material: np.ndarray, shadow_terminator_offset: float = 0.0) -> np.ndarray:
    """One INSTANCE record (src/scene_stage.cc:1085-1114)."""
    inst = np.zeros(1, dtype=INSTANCE)
    inst["light_base_id"] = -1
    inst["sh_grid_index"] = -1
    inst["shadow_terminator_mul"] = 1.0 / (1.0 - 0.5 * shadow_terminator_offset)
    inst["model"][0] = to_glm(model)
    inst["model_normal"][0] = to_glm(np.linalg.inv(model).T)
    inst["model_prev"][0] = to_glm(model)
    inst["mat"][0] = material
    return inst


def make_material(albedo=(1, 1, 1, 1), metallic=1.0, roughness=1.0, emission=(0, 0, 0),
                  transmittance=0.0, ior=1.45, normal_factor=1.0, double_sided=False,
                  albedo_tex=-1, mr_tex=-1, normal_tex=-1, emission_tex=-1) -> np.ndarray:
    m = np.zeros(1, dtype=MATERIAL)
    m["albedo_factor"] = albedo
    m["metallic_roughness_factor"] = (metallic, roughness, 0, 0)
    m["emission_factor"] = (emission[0], emission[1], emission[2], 0)
    m["transmittance"] = transmittance
    m["ior"] = ior
    m["normal_factor"] = normal_factor
    m["flags"] = MATERIAL_FLAG_DOUBLE_SIDED if double_sided else 0
    m["albedo_tex_id"] = albedo_tex
    m["metallic_roughness_tex_id"] = mr_tex
    m["normal_tex_id"] = normal_tex
    m["emission_tex_id"] = emission_tex
    return m[0]


def make_directional_light(color, direction, angle_deg) -> np.ndarray:
    """directional_light_entry (src/scene_stage.cc:46-62)."""
    d = np.zeros(1, dtype=DIRECTIONAL_LIGHT)
    dirn = np.asarray(direction, dtype=np.float64)
    d["color"] = color
    d["shadow_map_index"] = -1
    d["dir"] = dirn / np.linalg.norm(dirn)
    d["dir_cutoff"] = math.cos(math.radians(angle_deg))
    return d


def make_point_light(color, pos, radius, cutoff_brightness=5.0 / 256.0) -> np.ndarray:
    """point_light_entry (src/scene_stage.cc:64-98), point variant."""
    p = np.zeros(1, dtype=POINT_LIGHT)
    p["color"] = color
    p["pos"] = pos
    p["radius"] = radius
    p["cutoff_radius"] = math.sqrt(max(color) / cutoff_brightness) if cutoff_brightness > 0 else 0
    p["spot_radius"] = -1.0
    p["shadow_map_index"] = -1
    return p


def make_spotlight(color, pos, direction, radius, cutoff_angle_deg, falloff_exponent,
                   cutoff_brightness=5.0 / 256.0) -> np.ndarray:
    p = make_point_light(color, pos, radius, cutoff_brightness)
    dirn = np.asarray(direction, dtype=np.float64)
    p["dir"] = dirn / np.linalg.norm(dirn)
    p["dir_cutoff"] = math.cos(math.radians(cutoff_angle_deg))
    p["dir_falloff"] = falloff_exponent
    p["spot_radius"] = float(p["cutoff_radius"][0]) * math.tan(math.radians(cutoff_angle_deg))
    return p


def spotlight_falloff_from_inner_angle(inner_deg, cutoff_deg, ratio=4 / 255.0) -> float:
    """spotlight::set_inner_angle (src/light.cc:103-112)."""
    if inner_deg <= 0:
        return 1.0
    inner = math.cos(math.radians(inner_deg))
    outer = math.cos(math.radians(cutoff_deg))
    return math.log(ratio) / math.log(max(1.0 - inner, 0.0) / (1.0 - outer))
