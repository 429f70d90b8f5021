"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (tauray_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.cc", "oracle.h", "glsl.h", "sobol_table.inc", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


class SceneDescC(C.Structure):
    _fields_ = [
        ("instances", C.c_void_p), ("spans", C.c_void_p), ("instance_count", C.c_uint32),
        ("vertices", C.c_void_p), ("vertex_count", C.c_uint32),
        ("indices", C.c_void_p), ("index_count", C.c_uint32),
        ("point_lights", C.c_void_p), ("point_light_count", C.c_uint32),
        ("directional_lights", C.c_void_p), ("directional_light_count", C.c_uint32),
        ("texture_infos", C.c_void_p), ("texture_count", C.c_uint32), ("texels", C.c_void_p),
        ("envmap", C.c_void_p), ("envmap_width", C.c_uint32), ("envmap_height", C.c_uint32),
        ("alias_table", C.c_void_p), ("environment_factor", C.c_float * 4),
        ("cameras", C.c_void_p), ("camera_count", C.c_uint32),
        ("non_opaque", C.c_void_p), ("gather_emissive_triangles", C.c_uint32)]


class PtOptionsC(C.Structure):
    _fields_ = [
        ("max_bounces", C.c_int32), ("min_ray_dist", C.c_float), ("rng_seed", C.c_uint32), ("sampler", C.c_int32),
        ("samples_per_pixel", C.c_int32), ("samples_per_pass", C.c_int32), ("projection", C.c_int32),
        ("film", C.c_int32), ("film_radius", C.c_float), ("mis_mode", C.c_int32),
        ("russian_roulette_delta", C.c_float), ("indirect_clamping", C.c_float), ("regularization_gamma", C.c_float),
        ("depth_of_field", C.c_int32), ("nee_point", C.c_float), ("nee_directional", C.c_float),
        ("nee_envmap", C.c_float), ("nee_triangles", C.c_float), ("bounce_mode", C.c_int32),
        ("tri_light_mode", C.c_int32), ("hide_lights", C.c_int32), ("use_white_albedo_on_first_bounce", C.c_int32),
        ("transparent_background", C.c_int32), ("pre_transformed_vertices", C.c_int32)]


class PtTargetsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("color", "diffuse", "reflection", "albedo", "material", "normal", "pos", "instance_id", "screen_motion")]


class DistributionC(C.Structure):
    _fields_ = [("size_x", C.c_uint32), ("size_y", C.c_uint32), ("strategy", C.c_int32),
                ("index", C.c_uint32), ("count", C.c_uint32), ("primary", C.c_uint32)]


class CountersC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest_rays", "shadow_rays", "node_visits", "tri_tests", "alpha_tests", "surface_hits")]


HIT_DTYPE = np.dtype([("instance_id", "<i4"), ("primitive_id", "<i4"), ("bary_u", "<f4"), ("bary_v", "<f4"), ("t", "<f4")])


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.POINTER(SceneDescC)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_scene_tri_light_count.restype = C.c_uint32
        L.oracle_scene_tri_light_count.argtypes = [C.c_void_p]
        L.oracle_scene_get_tri_lights.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_skin_vertices.restype = None
        L.oracle_skin_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_scene_set_shard.restype = C.c_int
        L.oracle_scene_set_shard.argtypes = [C.c_void_p] + [C.c_uint32] * 4
        L.oracle_scene_set_previous_cameras.restype = C.c_int
        L.oracle_scene_set_previous_cameras.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_direct_render_targets.restype = C.c_int
        L.oracle_direct_render_targets.argtypes = [C.c_void_p, C.POINTER(PtOptionsC), C.POINTER(DistributionC), C.c_uint32, C.c_uint32,
                                                   C.c_uint32, C.POINTER(PtTargetsC), C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_pt_render_targets.restype = C.c_int
        L.oracle_pt_render_targets.argtypes = [C.c_void_p, C.POINTER(PtOptionsC), C.POINTER(DistributionC), C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.POINTER(PtTargetsC), C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_pt_render.restype = C.c_int
        L.oracle_pt_render.argtypes = [C.c_void_p, C.POINTER(PtOptionsC), C.POINTER(DistributionC), C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_feature_render.restype = C.c_int
        L.oracle_feature_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(DistributionC), C.c_int, C.c_uint32, C.c_float,
                                            C.POINTER(C.c_float), C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_trace_closest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.oracle_trace_shadow.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_float]
        L.oracle_get_counters.argtypes = [C.c_void_p, C.POINTER(CountersC)]
        L.oracle_reset_counters.argtypes = [C.c_void_p]
        u32p = C.POINTER(C.c_uint32)
        f32p = C.POINTER(C.c_float)
        L.oracle_pcg.restype = C.c_uint32; L.oracle_pcg.argtypes = [u32p]
        L.oracle_pcg2d.argtypes = [u32p, u32p]
        L.oracle_pcg4d.argtypes = [u32p, u32p]
        L.oracle_init_random_sampler.argtypes = [u32p, u32p]
        L.oracle_generate_sobol_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p]
        L.oracle_owen_scramble_2d.argtypes = [u32p, u32p, u32p]
        for n in ("oracle_owen_scramble_4d", "oracle_owen_scramble_8d", "oracle_morton_2d"):
            getattr(L, n).restype = C.c_uint32; getattr(L, n).argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_get_permutation_n.restype = C.c_uint32; L.oracle_get_permutation_n.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.oracle_morton_3d.restype = C.c_uint32; L.oracle_morton_3d.argtypes = [C.c_uint32] * 3
        L.oracle_ray_sample_uint.argtypes = [C.c_int, C.c_int, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p]
        L.oracle_rgb_to_r9g9b9e5.restype = C.c_uint32; L.oracle_rgb_to_r9g9b9e5.argtypes = [f32p]
        L.oracle_r9g9b9e5_to_rgb.argtypes = [C.c_uint32, f32p]
        L.oracle_pack_half2x16.restype = C.c_uint32; L.oracle_pack_half2x16.argtypes = [C.c_float, C.c_float]
        L.oracle_permute_region_id.restype = C.c_uint32; L.oracle_permute_region_id.argtypes = [C.c_uint32] * 4
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 6 + [C.c_int, f32p, f32p]
        L.oracle_sample_cone.argtypes = [C.c_float, C.c_float, f32p, C.c_float, f32p]
        L.oracle_sample_spherical_triangle.argtypes = [C.c_float, C.c_float, f32p, f32p, f32p, f32p, f32p]
        L.oracle_ggx_vndf_sample.argtypes = [f32p, C.c_float, C.c_float, C.c_float, f32p]
        L.oracle_ggx_bsdf_sample.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p]
        L.oracle_ggx_bsdf_pdf.restype = C.c_float; L.oracle_ggx_bsdf_pdf.argtypes = [f32p, f32p, f32p, f32p]
        L.oracle_sample_blackman_harris.restype = C.c_float; L.oracle_sample_blackman_harris.argtypes = [C.c_float]
    return _LIB


def _ptr(a):
    return a.ctypes.data if a is not None and a.size else None


def make_options(**kw) -> PtOptionsC:
    """CLI defaults of the reference unless overridden."""
    o = PtOptionsC(max_bounces=8, min_ray_dist=1e-4, rng_seed=0, sampler=0, samples_per_pixel=1, samples_per_pass=1,
                   projection=0, film=0, film_radius=0.5, mis_mode=2, russian_roulette_delta=0.0, indirect_clamping=0.0,
                   regularization_gamma=0.0, depth_of_field=0, nee_point=1.0, nee_directional=1.0, nee_envmap=1.0,
                   nee_triangles=1.0, bounce_mode=2, tri_light_mode=1, hide_lights=0, use_white_albedo_on_first_bounce=0,
                   transparent_background=0, pre_transformed_vertices=0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def options_for_scene(scene, **kw) -> PtOptionsC:
    """create_renderer's per-scene NEE weight zeroing (src/tauray.cc:417-421)."""
    o = make_options(**kw)
    if len(scene.point_lights) == 0:
        o.nee_point = 0.0
    if len(scene.directional_lights) == 0:
        o.nee_directional = 0.0
    if scene.envmap is None:
        o.nee_envmap = 0.0
    if not scene.has_tri_lights():
        o.nee_triangles = 0.0
    return o


class OracleScene:
    def __init__(self, scene, node_globals=None):
        """`node_globals`: pose of the joint nodes of a scene with skinned meshes (default: the file's rest pose)."""
        L = lib()
        self._keep = []
        if getattr(scene, "skinned", None):
            import copy
            posed = copy.copy(scene)
            posed.vertices = scene.vertices.copy()
            for sk in scene.skinned:
                sp = scene.spans[sk.instance]
                lo, hi = int(sp["vertex_offset"]), int(sp["vertex_offset"]) + int(sp["vertex_count"])
                posed.vertices[lo:hi] = skin_vertices(scene.vertices[lo:hi], sk.skins, scene.joint_transforms(sk, node_globals))
            scene = posed
        infos, texels = scene.texture_table()
        cams = scene.camera_data()
        non_opaque = scene.potentially_transparent().astype(np.uint8)
        d = SceneDescC()

        def keep(a):
            a = np.ascontiguousarray(a)
            self._keep.append(a)
            return a

        inst, spans, verts, idx = keep(scene.instances), keep(scene.spans), keep(scene.vertices), keep(scene.indices)
        pls, dls = keep(scene.point_lights), keep(scene.directional_lights)
        infos, texels, cams, non_opaque = keep(infos), keep(texels), keep(cams), keep(non_opaque)
        d.instances, d.spans, d.instance_count = _ptr(inst), _ptr(spans), len(inst)
        d.vertices, d.vertex_count = _ptr(verts), len(verts)
        d.indices, d.index_count = _ptr(idx), len(idx)
        d.point_lights, d.point_light_count = _ptr(pls), len(pls)
        d.directional_lights, d.directional_light_count = _ptr(dls), len(dls)
        d.texture_infos, d.texture_count, d.texels = _ptr(infos), len(infos), _ptr(texels)
        if scene.envmap is not None:
            from tauray_amd.scene import build_alias_table
            env = keep(np.asarray(scene.envmap, dtype=np.float32))
            at = keep(build_alias_table(env))
            d.envmap, d.envmap_width, d.envmap_height, d.alias_table = _ptr(env), env.shape[1], env.shape[0], _ptr(at)
        d.environment_factor = (C.c_float * 4)(*[float(x) for x in scene.environment_factor])
        d.cameras, d.camera_count = _ptr(cams), len(cams)
        d.non_opaque = _ptr(non_opaque)
        d.gather_emissive_triangles = 1 if getattr(scene, "tri_light_count", 0) > 0 else 0
        self.h = L.oracle_scene_create(C.byref(d))
        self.scene = scene

    def __del__(self):
        try:
            if self.h:
                lib().oracle_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tri_lights(self):
        from tauray_amd.scene import TRI_LIGHT
        n = lib().oracle_scene_tri_light_count(self.h)
        out = np.zeros(n, dtype=TRI_LIGHT)
        if n:
            lib().oracle_scene_get_tri_lights(self.h, out.ctypes.data)
        return out

    def render_pt(self, opt: PtOptionsC, width, height, dist: DistributionC = None, viewports=1, frame_counter=0,
                  samples_accumulated=0, color=None, target_size=None, threads=0):
        if dist is None:
            dist = DistributionC(width, height, 0, 0, 1, 1)
        tw, th = target_size if target_size else (width, height)
        if color is None:
            color = np.zeros((viewports, th, tw, 4), dtype=np.float32)
        rc = lib().oracle_pt_render(self.h, C.byref(opt), C.byref(dist), viewports, frame_counter, samples_accumulated,
                                    color.ctypes.data, tw, th, threads)
        if rc != 0:
            raise RuntimeError("oracle_pt_render failed")
        return color

    TARGETS = {"color": (4, np.float32), "diffuse": (4, np.float32), "reflection": (4, np.float32), "albedo": (4, np.float32),
               "material": (4, np.float32), "normal": (2, np.float32), "pos": (4, np.float32), "instance_id": (1, np.int32),
               "screen_motion": (2, np.float32)}

    def set_previous_cameras(self, cameras):
        data = np.concatenate([c.pack() for c in cameras])
        if lib().oracle_scene_set_previous_cameras(self.h, data.ctypes.data, len(cameras)) != 0:
            raise RuntimeError("oracle_scene_set_previous_cameras: camera count mismatch")

    def render_pt_targets(self, opt: PtOptionsC, width, height, names, dist: DistributionC = None, viewports=1, frame_counter=0,
                          samples_accumulated=0, targets=None, target_size=None, threads=0, direct=False):
        """oracle_pt_render_targets (or, with `direct`, oracle_direct_render_targets = direct_stage): returns {name: array[viewports, th, tw, channels]} for the requested gbuffer targets."""
        if dist is None:
            dist = DistributionC(width, height, 0, 0, 1, 1)
        tw, th = target_size if target_size else (width, height)
        out = dict(targets) if targets else {}
        t = PtTargetsC()
        for n in names:
            ch, dt = self.TARGETS[n]
            if n not in out:
                out[n] = np.zeros((viewports, th, tw, ch), dtype=dt)
            setattr(t, n, out[n].ctypes.data)
        fn = lib().oracle_direct_render_targets if direct else lib().oracle_pt_render_targets
        rc = fn(self.h, C.byref(opt), C.byref(dist), viewports, frame_counter, samples_accumulated, C.byref(t), tw, th, threads)
        if rc != 0:
            raise RuntimeError("oracle render_targets failed")
        return out

    def render_feature(self, feature, width, height, dist: DistributionC = None, projection=0, viewport=0,
                       min_ray_dist=1e-4, default_value=(np.nan,) * 4, threads=0, target_size=None):
        if dist is None:
            dist = DistributionC(width, height, 0, 0, 1, 1)
        tw, th = target_size if target_size else (width, height)
        color = np.zeros((th, tw, 4), dtype=np.float32)
        dv = (C.c_float * 4)(*default_value)
        rc = lib().oracle_feature_render(self.h, feature, C.byref(dist), projection, viewport, min_ray_dist, dv,
                                         color.ctypes.data, tw, th, threads)
        if rc != 0:
            raise RuntimeError("oracle_feature_render failed")
        return color

    def trace_closest(self, rays, seeds=None, include_lights=False, threads=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        lib().oracle_trace_closest(self.h, len(rays), rays.ctypes.data, _ptr(seeds) if seeds is not None else None,
                                   1 if include_lights else 0, out.ctypes.data, threads)
        return out

    def trace_shadow(self, rays, threads=0):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), dtype=np.float32)
        lib().oracle_trace_shadow(self.h, len(rays), rays.ctypes.data, out.ctypes.data, threads)
        return out

    def set_shard(self, viewport_base=0, viewport_stride=1, sample_base=0, sample_stride=1):
        if lib().oracle_scene_set_shard(self.h, viewport_base, viewport_stride, sample_base, sample_stride) != 0:
            raise RuntimeError("oracle_scene_set_shard: bad shard")

    def counters(self):
        c = CountersC()
        lib().oracle_get_counters(self.h, C.byref(c))
        return {n: int(getattr(c, n)) for n, _ in CountersC._fields_}

    def reset_counters(self):
        lib().oracle_reset_counters(self.h)


def tonemap(img, op=2, exposure=1.0, gamma=2.2):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.zeros_like(img)
    lib().oracle_tonemap(img.ctypes.data, out.ctypes.data, img.size // 4, op, exposure, gamma)
    return out


def skin_vertices(source: np.ndarray, skins: np.ndarray, joint_transforms: np.ndarray) -> np.ndarray:
    """shader/skinning.comp over one mesh.  `joint_transforms`: (n, 4, 4) matrices as numpy writes them (row-major)."""
    source = np.ascontiguousarray(source)
    skins = np.ascontiguousarray(skins)
    j = np.ascontiguousarray(np.asarray(joint_transforms, dtype=np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
    out = np.zeros_like(source)
    lib().oracle_skin_vertices(source.ctypes.data, skins.ctypes.data, len(source), j.ctypes.data, len(j), out.ctypes.data)
    return out
