"""Host-side mirror of the reference's renderer/stage surface for the path-tracer hot path,
driving the HIP kernels through the C ABI (include/trhip.h).

Reference classes mirrored (same names, argument meaning and error behaviour; errors surface as
TrhipError where the reference throws std::runtime_error):

* ``Context``            - tr::context / tr::device pair for ONE device (src/context.hh, src/device.hh)
* ``SceneStage``         - scene_stage uploads + acceleration structure (src/scene_stage.cc)
* ``PathTracerStage``    - path_tracer_stage / rt_camera_stage / rt_stage (src/path_tracer_stage.{hh,cc})
* ``FeatureStage``       - feature_stage (src/feature_stage.{hh,cc})
* ``StitchStage``        - stitch_stage (src/stitch_stage.{hh,cc})
* ``TonemapStage``       - tonemap_stage (src/tonemap_stage.{hh,cc})
* ``RtRenderer``         - rt_renderer<path_tracer_stage> (src/rt_renderer.{hh,cc}); one process per GPU,
                           partial frames gathered with torch.distributed (RCCL) instead of host-bounce copies
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import (AccelInfoC, CountersC, DistributionC, PtOptionsC, SceneDescC, TimingsC, TonemapInfoC, TrhipError, check)
from .distribution import (DISTRIBUTION_DUPLICATE, DISTRIBUTION_SCANLINE, DISTRIBUTION_SHUFFLED_STRIPS, DistributionParams,
                           get_device_distribution_params, get_distribution_target_size)
from .scene import SceneDesc, build_alias_table

# film_filter, multiple_importance_sampling_mode, bounce_sampling_mode, tri_light_sampling_mode (src/rt_common.hh)
FILM_POINT, FILM_BOX, FILM_BLACKMAN_HARRIS = 0, 1, 2
MIS_DISABLED, MIS_BALANCE_HEURISTIC, MIS_POWER_HEURISTIC = 0, 1, 2
BOUNCE_HEMISPHERE, BOUNCE_COSINE_HEMISPHERE, BOUNCE_MATERIAL = 0, 1, 2
TRI_LIGHT_AREA, TRI_LIGHT_SOLID_ANGLE, TRI_LIGHT_HYBRID = 0, 1, 2
SAMPLER_UNIFORM_RANDOM, SAMPLER_SOBOL_OWEN, SAMPLER_SOBOL_Z_ORDER_2D, SAMPLER_SOBOL_Z_ORDER_3D = 0, 1, 2, 3
TONEMAP_LINEAR, TONEMAP_GAMMA_CORRECTION, TONEMAP_FILMIC, TONEMAP_REINHARD, TONEMAP_REINHARD_LUMINANCE = 0, 1, 2, 3, 4
FEATURE_ALBEDO, FEATURE_WORLD_NORMAL, FEATURE_VIEW_NORMAL, FEATURE_WORLD_POS, FEATURE_VIEW_POS, FEATURE_DISTANCE = 0, 1, 2, 3, 4, 5
FEATURE_INSTANCE_ID = 9


def make_options(**kw) -> PtOptionsC:
    """path_tracer_stage::options at the reference's CLI defaults (src/options.hh)."""
    o = PtOptionsC(max_bounces=8, min_ray_dist=1e-4, rng_seed=0, sampler=SAMPLER_UNIFORM_RANDOM, samples_per_pixel=1,
                   samples_per_pass=1, projection=0, film=FILM_POINT, film_radius=0.5, mis_mode=MIS_POWER_HEURISTIC,
                   russian_roulette_delta=0.0, indirect_clamping=0.0, regularization_gamma=0.0, depth_of_field=0,
                   nee_point=1.0, nee_directional=1.0, nee_envmap=1.0, nee_triangles=1.0, bounce_mode=BOUNCE_MATERIAL,
                   tri_light_mode=TRI_LIGHT_SOLID_ANGLE, hide_lights=0, use_white_albedo_on_first_bounce=0,
                   transparent_background=0, pre_transformed_vertices=0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(f"unknown path tracer option {k!r}")
        setattr(o, k, v)
    return o


def options_for_scene(scene: SceneDesc, **kw) -> PtOptionsC:
    """create_renderer's per-scene set-up (src/tauray.cc:355-421): NEE weights are zeroed for light
    classes the scene does not have, projection follows the scene camera."""
    o = make_options(**kw)
    if len(scene.point_lights) == 0:
        o.nee_point = 0.0
    if len(scene.directional_lights) == 0:
        o.nee_directional = 0.0
    if scene.envmap is None:
        o.nee_envmap = 0.0
    if not scene.has_tri_lights():
        o.nee_triangles = 0.0
    if scene.cameras and "projection" not in kw:
        o.projection = scene.cameras[0].projection
    return o


def copy_options(o: PtOptionsC, **changes) -> PtOptionsC:
    c = PtOptionsC()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(PtOptionsC))
    for k, v in changes.items():
        setattr(c, k, v)
    return c


class DeviceBuffer:
    """gpu_buffer-like owner of one device allocation."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(_lib.lib().trhip_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def data_ptr(self):
        return self.ptr

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(_lib.lib().trhip_upload(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes, None))
        return self

    def download(self, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(_lib.lib().trhip_download(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes, None))
        return out

    def zero(self):
        check(_lib.lib().trhip_memset(self.ctx.h, self.ptr, 0, self.nbytes, None))
        return self

    def free(self):
        if self.ptr:
            _lib.lib().trhip_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(x):
    return x.data_ptr() if hasattr(x, "data_ptr") else int(x)


class Context:
    """One HIP device (the reference's context enumerates all Vulkan devices in one process; here each
    process owns one GPU and peers are reached through torch.distributed)."""

    def __init__(self, hip_device: int = 0):
        h = C.c_void_p()
        check(_lib.lib().trhip_device_create(hip_device, C.byref(h)))
        self.h = h.value
        self.hip_device = hip_device

    def alloc(self, nbytes) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def sync(self, stream=None):
        check(_lib.lib().trhip_sync(self.h, stream))

    def create_stream(self) -> int:
        st = C.c_void_p()
        check(_lib.lib().trhip_stream_create(self.h, C.byref(st)))
        return st.value

    def destroy_stream(self, stream):
        check(_lib.lib().trhip_stream_destroy(self.h, stream))

    def stream_pipe_class(self, stream=None) -> int:
        """The hardware pipe (a small integer) the queue of `stream` sits on (None = the default stream); launches that fill the chip
        overlap only between streams of different pipes (csrc/stream_pool.hip)."""
        c = C.c_int32(-1)
        check(_lib.lib().trhip_stream_pipe_class(self.h, stream, C.byref(c)))
        return c.value

    def info(self) -> dict:
        """trhip_device_get_info: which device this is (arch, PCI bus id, UUID) and how many hardware pipes the process reaches on it
        (include/trhip.h, "process requirements": 4 on MI355X with GPU_MAX_HW_QUEUES >= 8)."""
        class Info(C.Structure):
            _fields_ = [("struct_size", C.c_uint32), ("hip_device", C.c_int32), ("name", C.c_char * 64), ("pci_bus_id", C.c_char * 24),
                        ("uuid", C.c_uint8 * 16), ("compute_units", C.c_int32), ("pipe_classes", C.c_int32), ("pool_streams", C.c_int32),
                        ("hw_queues_env", C.c_int32)]
        i = Info()
        check(_lib.lib().trhip_device_get_info(self.h, C.byref(i)))
        return {"hip_device": i.hip_device, "name": i.name.decode("ascii", "replace"), "pci_bus_id": i.pci_bus_id.decode("ascii", "replace"),
                "uuid": bytes(i.uuid).hex(), "compute_units": i.compute_units, "pipe_classes": i.pipe_classes, "pool_streams": i.pool_streams,
                "hw_queues_env": i.hw_queues_env}

    def stream_wait(self, stream, on):
        """Work enqueued on `stream` from now on waits for what is on `on` now (None = the default stream)."""
        check(_lib.lib().trhip_stream_wait(self.h, stream, on))

    def calibrate_valu(self) -> float:
        """Peak vector-instruction issue rate of the device as it runs now, in 10^9 wave-level instructions per second."""
        g = C.c_float()
        check(_lib.lib().trhip_calibrate_valu(self.h, C.byref(g)))
        return float(g.value)

    def calibrate_l1(self) -> float:
        """Peak rate of cache-line accesses of the vector L1 caches of the device, in 10^9 accesses per second."""
        g = C.c_float()
        check(_lib.lib().trhip_calibrate_l1(self.h, C.byref(g)))
        return float(g.value)

    def close(self):
        if self.h:
            _lib.lib().trhip_device_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SceneStage:
    """scene_stage: uploads the flattened scene and builds the acceleration structure on the device."""

    def __init__(self, ctx: Context, scene: Optional[SceneDesc] = None, fast_trace_rebuilds: bool = False):
        self.ctx = ctx
        self.scene = None
        self.accel = None
        # the first build of a scene is a static build (tree optimisation on); rebuilds after a change are fast builds unless asked
        self.fast_trace_rebuilds = fast_trace_rebuilds
        if scene is not None:
            self.set_scene(scene)

    def set_scene(self, scene: SceneDesc):
        L = _lib.lib()
        keep = []

        def k(a):
            a = np.ascontiguousarray(a)
            keep.append(a)
            return a

        def p(a):
            return a.ctypes.data if a.size else None

        infos, texels = scene.texture_table()
        inst, spans, verts, idx = k(scene.instances), k(scene.spans), k(scene.vertices), k(scene.indices)
        pls, dls, infos, texels = k(scene.point_lights), k(scene.directional_lights), k(infos), k(texels)
        cams, non_opaque = k(scene.camera_data()), k(scene.potentially_transparent().astype(np.uint8))
        d = SceneDescC()
        d.instances, d.spans, d.instance_count = p(inst), p(spans), len(inst)
        d.vertices, d.vertex_count = p(verts), len(verts)
        d.indices, d.index_count = p(idx), len(idx)
        d.point_lights, d.point_light_count = p(pls), len(pls)
        d.directional_lights, d.directional_light_count = p(dls), len(dls)
        d.texture_infos, d.texture_count, d.texels = p(infos), len(infos), p(texels)
        if scene.envmap is not None:
            env = k(np.asarray(scene.envmap, dtype=np.float32))
            at = k(build_alias_table(env))
            d.envmap, d.envmap_width, d.envmap_height, d.alias_table = p(env), env.shape[1], env.shape[0], p(at)
        d.environment_factor = (C.c_float * 4)(*[float(x) for x in scene.environment_factor])
        d.cameras, d.camera_count = p(cams), len(cams)
        d.non_opaque = p(non_opaque)
        d.gather_emissive_triangles = 1 if getattr(scene, "tri_light_count", 0) > 0 else 0
        check(L.trhip_scene_upload(self.ctx.h, C.byref(d)))
        self.scene = scene
        # skinned meshes: the uploaded vertices are the bind pose; pose them with the file's rest pose before the build
        # (the reference runs skinning.comp on the first scene update, src/scene_stage.cc:1543-1567)
        for sk in getattr(scene, "skinned", []):
            self.set_skin(sk.instance, sk.skins)
            self.skin(sk.instance, scene.joint_transforms(sk), refit=None)
        info = AccelInfoC()
        check(L.trhip_scene_set_build_mode(self.ctx.h, 0))
        check(L.trhip_scene_build_accel(self.ctx.h, C.byref(info)))
        self.accel = dict(triangle_count=info.triangle_count, node_count=info.node_count,
                          tri_light_count=info.tri_light_count, build_ms=info.build_ms,
                          bounds_min=tuple(info.bounds_min), bounds_max=tuple(info.bounds_max),
                          node_bytes=info.node_bytes, leaf_count=info.leaf_count)
        return self.accel

    def pose(self, node_globals: dict, refit: bool = True):
        """New global transforms of the joint nodes (an animation step of the caller's): every skinned mesh is skinned again
        and the acceleration structure updated."""
        for sk in self.scene.skinned:
            self.skin(sk.instance, self.scene.joint_transforms(sk, node_globals), refit=None)
        return self._accel_after_change(refit)

    def animate(self, animator, dt_ticks: int, refit: bool = True):
        """One frame of a playing animation (update(scene, dt) of src/scene.cc:226-235 followed by scene_stage::update):
        `animator` (tauray_amd.animation.SceneAnimator over this stage's scene) advances by dt microseconds; the new instance
        records, cameras (with last frame's as camera_pair.previous) and joint matrices go to the device, and the acceleration
        structure is updated once - refitted, or rebuilt with `refit=False`."""
        instances, cameras, node_globals = animator.update(dt_ticks)
        inst = np.ascontiguousarray(instances)
        check(_lib.lib().trhip_scene_update_instances(self.ctx.h, inst.ctypes.data, len(inst)))
        for sk in self.scene.skinned:
            self.skin(sk.instance, self.scene.joint_transforms(sk, node_globals), refit=None)
        self.update_cameras(cameras)
        self.set_previous_cameras(animator.previous_cameras)
        pl, dl = np.ascontiguousarray(self.scene.point_lights), np.ascontiguousarray(self.scene.directional_lights)
        if len(pl) or len(dl):          # lights on moving nodes
            check(_lib.lib().trhip_scene_update_lights(self.ctx.h, pl.ctypes.data if len(pl) else None, len(pl), dl.ctypes.data if len(dl) else None, len(dl)))
        return self._accel_after_change(refit)

    def update_cameras(self, cameras):
        data = np.concatenate([c.pack() for c in cameras])
        check(_lib.lib().trhip_scene_update_cameras(self.ctx.h, data.ctypes.data, len(data)))

    def set_previous_cameras(self, cameras):
        """camera_pair.previous per viewport (motion features, screen-motion target)."""
        data = np.concatenate([c.pack() for c in cameras])
        check(_lib.lib().trhip_scene_set_previous_cameras(self.ctx.h, data.ctypes.data, len(data)))

    def update_instances(self, instances: np.ndarray, refit: bool = False):
        """Dynamic scenes: new instance records (transforms / materials) for the same meshes, then a full rebuild of the
        acceleration structure on the device (what scene_stage::update + the TLAS rebuild do per frame) or, with
        `refit`, an update that keeps the tree and recomputes its boxes."""
        inst = np.ascontiguousarray(instances)
        check(_lib.lib().trhip_scene_update_instances(self.ctx.h, inst.ctypes.data, len(inst)))
        return self._accel_after_change(refit)

    def _accel_after_change(self, refit: bool):
        info = AccelInfoC()
        if refit:
            check(_lib.lib().trhip_scene_refit_accel(self.ctx.h, C.byref(info)))
        else:
            # a scene that is rebuilt after its first build is dynamic: ePreferFastBuild (src/acceleration_structure.cc:129-131)
            check(_lib.lib().trhip_scene_set_build_mode(self.ctx.h, 0 if self.fast_trace_rebuilds else 1))
            check(_lib.lib().trhip_scene_build_accel(self.ctx.h, C.byref(info)))
        self.accel.update(node_count=info.node_count, build_ms=info.build_ms, tri_light_count=info.tri_light_count,
                          bounds_min=tuple(info.bounds_min), bounds_max=tuple(info.bounds_max), leaf_count=info.leaf_count)
        return self.accel

    def set_skin(self, instance: int, skins: np.ndarray, source: Optional[np.ndarray] = None):
        """Marks the mesh of `instance` as skinned (mesh::get_animation_source + skin buffer, src/mesh.hh:32-73): `skins` is
        one SKIN record per vertex, `source` the bind-pose vertices (default: the uploaded ones)."""
        from .scene import SKIN
        skins = np.ascontiguousarray(skins, dtype=SKIN)
        src_ptr = None
        if source is not None:
            source = np.ascontiguousarray(source)
            if len(source) != len(skins):
                raise ValueError("set_skin: source and skins differ in length")
            src_ptr = source.ctypes.data
        check(_lib.lib().trhip_scene_set_skin(self.ctx.h, instance, src_ptr, skins.ctypes.data if len(skins) else None, len(skins)))

    def skin(self, instance: int, joint_transforms: np.ndarray, refit: Optional[bool] = True):
        """scene_stage::record_skinning (src/scene_stage.cc:1543-1612): shader/skinning.comp over the instance's mesh with
        the given joint matrices ((n, 4, 4), row-major as numpy writes a matrix; uploaded column-major), then the
        acceleration-structure update (`refit`), a rebuild (`refit=False`) or nothing (`refit=None`: caller batches)."""
        j = np.ascontiguousarray(np.asarray(joint_transforms, dtype=np.float32).reshape(-1, 4, 4).transpose(0, 2, 1))
        check(_lib.lib().trhip_scene_skin(self.ctx.h, instance, j.ctypes.data, len(j)))
        return None if refit is None else self._accel_after_change(refit)

    def vertices(self, instance: int) -> np.ndarray:
        from .scene import VERTEX
        n = int(self.scene.spans[instance]["vertex_count"])
        out = np.zeros(n, dtype=VERTEX)
        if n:
            check(_lib.lib().trhip_scene_get_vertices(self.ctx.h, instance, out.ctypes.data, n))
        return out

    def tri_lights(self) -> np.ndarray:
        from .scene import TRI_LIGHT
        n = self.accel["tri_light_count"]
        out = np.zeros(n, dtype=TRI_LIGHT)
        if n:
            check(_lib.lib().trhip_scene_get_tri_lights(self.ctx.h, out.ctypes.data, n))
        return out

    def trace_closest(self, rays: np.ndarray, seeds: Optional[np.ndarray] = None, include_lights=False) -> np.ndarray:
        """traceRayEXT closest-hit query on explicit rays (parity hook)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        hit_dtype = np.dtype([("instance_id", "<i4"), ("primitive_id", "<i4"), ("bary_u", "<f4"), ("bary_v", "<f4"), ("t", "<f4")])
        if n == 0:
            return np.zeros(0, dtype=hit_dtype)
        d_rays = self.ctx.alloc(rays.nbytes).upload(rays)
        d_seeds = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
            d_seeds = self.ctx.alloc(seeds.nbytes).upload(seeds)
        d_hits = self.ctx.alloc(n * 20)
        check(_lib.lib().trhip_trace_closest(self.ctx.h, n, d_rays.ptr, d_seeds.ptr if d_seeds else None,
                                             1 if include_lights else 0, d_hits.ptr, None))
        return d_hits.download((n,), hit_dtype)

    def trace_shadow(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        if n == 0:
            return np.zeros(0, dtype=np.float32)
        d_rays = self.ctx.alloc(rays.nbytes).upload(rays)
        d_vis = self.ctx.alloc(n * 4)
        check(_lib.lib().trhip_trace_shadow(self.ctx.h, n, d_rays.ptr, d_vis.ptr, None))
        return d_vis.download((n,), np.float32)


def _dist_c(p: DistributionParams) -> DistributionC:
    return DistributionC(int(p.size[0]), int(p.size[1]), int(p.strategy), int(p.index), int(p.count), 1 if p.primary else 0)


class PathTracerStage:
    """path_tracer_stage(device&, scene_stage&, const gbuffer_target&, const options&)."""

    _create = "trhip_pt_create"

    def __init__(self, ctx: Context, scene_stage: SceneStage, options: PtOptionsC, distribution: Optional[DistributionParams] = None):
        self.ctx, self.ss, self.opt = ctx, scene_stage, options
        h = C.c_void_p()
        check(getattr(_lib.lib(), self._create)(ctx.h, C.byref(options), C.byref(h)))
        self.h = h.value
        self.distribution = None
        if distribution is not None:
            self.reset_distribution_params(distribution)

    def reset_distribution_params(self, distribution: DistributionParams):
        d = _dist_c(distribution)
        check(_lib.lib().trhip_pt_set_distribution(self.h, C.byref(d)))
        self.distribution = distribution

    def set_shard(self, viewport_base=0, viewport_stride=1, sample_base=0, sample_stride=1):
        """View / sample sharding (trhip_pt_set_shard): local layer l = viewport base + l * stride, local sample s = sample
        base + s * stride of the pixel's sequence."""
        check(_lib.lib().trhip_pt_set_shard(self.h, viewport_base, viewport_stride, sample_base, sample_stride))

    def set_frame_counter(self, frame_counter: int):
        check(_lib.lib().trhip_pt_set_frame_counter(self.h, frame_counter))

    def set_frame_batch(self, frames: int):
        """trhip_pt_set_frame_batch: `frames` consecutive frames per run(), as frame-major layer groups of the target."""
        check(_lib.lib().trhip_pt_set_frame_batch(self.h, int(frames)))

    def set_shading_arithmetic(self, ieee: bool):
        """k_shade at IEEE fp32 with the C library's sin / cos / pow (True), or at the accuracy Vulkan asks of the reference's GLSL
        for the command-line option set (False, the default unless TRHIP_SHADE_FAST=0): include/trhip.h."""
        check(_lib.lib().trhip_pt_set_shading_arithmetic(self.h, int(bool(ieee))))

    def set_specialization(self, enable: bool):
        """A shading program compiled for this stage's option set the first time it renders (True, the default unless TRHIP_SPECIALIZE=0)
        or always the general kernels (False): include/trhip.h trhip_pt_set_specialization."""
        check(_lib.lib().trhip_pt_set_specialization(self.h, int(bool(enable))))

    def set_lanes(self, lanes: int):
        check(_lib.lib().trhip_pt_set_lanes(self.h, lanes))

    def set_fused_tonemap(self, display, info):
        """The stage's last pass writes tonemap(colour) into `display` as it writes the colour target (None: off); trhip_pt_set_fused_tonemap."""
        check(_lib.lib().trhip_pt_set_fused_tonemap(self.h, None if display is None else _ptr(display), None if info is None else C.byref(info)))

    def lane_pipes(self):
        """(lanes, [hardware pipe class of each lane's stream]) of the last render; trhip_pt_get_lane_pipes."""
        if not hasattr(_lib.lib(), "trhip_pt_get_lane_pipes"):
            return 0, []
        n = C.c_int32(0)
        cls = (C.c_int32 * 4)()
        check(_lib.lib().trhip_pt_get_lane_pipes(self.h, C.byref(n), cls))
        return n.value, [cls[i] for i in range(n.value)]

    def set_frame_slots(self, slots: int):
        """Hint: how many stages render next to this one (a renderer's frames in flight); trhip_pt_set_frame_slots."""
        if hasattr(_lib.lib(), "trhip_pt_set_frame_slots"):      # an older build named by TRHIP_LIB (A/B runs) has no such hint
            check(_lib.lib().trhip_pt_set_frame_slots(self.h, slots))

    def reset_accumulated_samples(self):
        check(_lib.lib().trhip_pt_reset_accumulation(self.h, 0))

    def reset_sample_counter(self):
        check(_lib.lib().trhip_pt_reset_accumulation(self.h, 1))

    def run(self, color_target, viewports=1, stream=None):
        """stage::run: enqueue one frame (all passes) into `color_target` (device RGBA32F)."""
        tw, th = get_distribution_target_size(self.distribution)
        check(_lib.lib().trhip_pt_render(self.h, _ptr(color_target), tw, th, viewports, stream))

    # gbuffer_target entries path_tracer.rgen can write (src/gbuffer.hh; shader/path_tracer.glsl:535-576):
    # name -> (channels, numpy dtype)
    TARGETS = {"color": (4, np.float32), "diffuse": (4, np.float32), "reflection": (4, np.float32), "albedo": (4, np.float32),
               "material": (4, np.float32), "normal": (2, np.float32), "pos": (4, np.float32), "instance_id": (1, np.int32),
               "screen_motion": (2, np.float32)}

    def run_targets(self, targets: dict, viewports=1, stream=None):
        """stage::run with a gbuffer: `targets` maps any subset of TARGETS to device images."""
        unknown = set(targets) - set(self.TARGETS)
        if unknown:
            raise ValueError(f"unknown gbuffer targets: {sorted(unknown)}")
        t = _lib.PtTargetsC()
        for name, buf in targets.items():
            setattr(t, name, _ptr(buf))
        tw, th = get_distribution_target_size(self.distribution)
        check(_lib.lib().trhip_pt_render_targets(self.h, C.byref(t), tw, th, viewports, stream))

    def set_profiling(self, count_work=False, detailed_timing=False):
        check(_lib.lib().trhip_pt_set_profiling(self.h, int(count_work), int(detailed_timing)))

    def counters(self) -> dict:
        c = CountersC()
        check(_lib.lib().trhip_pt_get_counters(self.h, C.byref(c)))
        return {n: int(getattr(c, n)) for n, _ in CountersC._fields_}

    def reset_counters(self):
        check(_lib.lib().trhip_pt_reset_counters(self.h))

    def timings(self) -> dict:
        t = TimingsC()
        check(_lib.lib().trhip_pt_get_timings(self.h, C.byref(t)))
        return {n: float(getattr(t, n)) for n, _ in TimingsC._fields_}

    def program(self) -> dict:
        """Which shading program renders this stage, resolved now (trhip_pt_get_program): kind ("general" | "cli" | "compiled"), ieee,
        identity (what the ranks of a job compare), key (the pinned option fields as text)."""
        from ._lib import ProgramInfoC
        p = ProgramInfoC()
        if not hasattr(_lib.lib(), "trhip_pt_get_program"):      # an older build named by TRHIP_LIB (A/B runs)
            return {"kind": "general", "ieee": False, "identity": 0, "key": "unknown: the library predates trhip_pt_get_program"}
        check(_lib.lib().trhip_pt_get_program(self.h, C.byref(p)))
        return {"kind": ("general", "cli", "compiled")[p.kind], "ieee": bool(p.ieee), "identity": int(p.identity), "key": p.key.decode()}

    def phase_counters(self) -> dict:
        """Wave-level phase statistics of the counting trace kernels (trhip_pt_get_phase_counters)."""
        from ._lib import PhaseCountersC
        p = PhaseCountersC()
        check(_lib.lib().trhip_pt_get_phase_counters(self.h, C.byref(p)))
        out = {n: int(getattr(p, n)) for n, _ in PhaseCountersC._fields_ if n != "lane_node_phase_hist"}
        out["lane_node_phase_hist"] = [int(x) for x in p.lane_node_phase_hist]
        return out

    def close(self):
        if self.h:
            _lib.lib().trhip_pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DirectStage(PathTracerStage):
    """direct_stage(device&, scene_stage&, const gbuffer_target&, const options&): first hit + samples_per_pass light samples
    (src/direct_stage.{hh,cc}); same surface as PathTracerStage."""
    _create = "trhip_direct_create"


class FeatureStage:
    """feature_stage: primary-hit AOVs from the same traversal."""

    def __init__(self, ctx: Context, scene_stage: SceneStage, feature: int, distribution: DistributionParams, projection=0,
                 min_ray_dist=1e-4, default_value=(np.nan,) * 4):
        self.ctx, self.ss, self.feature, self.distribution = ctx, scene_stage, feature, distribution
        self.projection, self.min_ray_dist, self.default_value = projection, min_ray_dist, default_value

    def run(self, color_target, viewport=0, stream=None):
        tw, th = get_distribution_target_size(self.distribution)
        d = _dist_c(self.distribution)
        dv = (C.c_float * 4)(*self.default_value)
        check(_lib.lib().trhip_feature_render(self.ctx.h, self.feature, C.byref(d), self.projection, viewport, self.min_ray_dist, dv,
                                              _ptr(color_target), tw, th, stream))


class StitchStage:
    """stitch_stage: scatter non-primary partial images into the primary image."""

    def __init__(self, ctx: Context, size, blend_ratio=1.0):
        self.ctx, self.size, self.blend_ratio = ctx, tuple(size), blend_ratio

    def set_blend_ratio(self, r):
        self.blend_ratio = r

    def run_one(self, partial_dist: DistributionParams, partial, primary, viewports=1, stream=None):
        pw, ph = get_distribution_target_size(partial_dist)
        d = _dist_c(partial_dist)
        check(_lib.lib().trhip_stitch(self.ctx.h, C.byref(d), _ptr(partial), pw, ph, _ptr(primary), viewports, self.blend_ratio, stream))


    def run_all(self, partial_dists, partials, primary, viewports=1, stream=None):
        """Every partial image in one launch (trhip_stitch_batch)."""
        n = len(partial_dists)
        if n == 0:
            return
        key = tuple((id(p), d.index, d.count, d.strategy) for d, p in zip(partial_dists, partials))
        if getattr(self, "_batch_key", None) != key:      # argument arrays are rebuilt only when the partials change
            ds = (DistributionC * n)(*[_dist_c(d) for d in partial_dists])
            ptrs = (C.c_void_p * n)(*[_ptr(p) for p in partials])
            sizes = [get_distribution_target_size(d) for d in partial_dists]
            pws = (C.c_uint32 * n)(*[w for w, _ in sizes])
            phs = (C.c_uint32 * n)(*[h for _, h in sizes])
            self._batch_key, self._batch_args = key, (ds, ptrs, pws, phs)
        ds, ptrs, pws, phs = self._batch_args
        check(_lib.lib().trhip_stitch_batch(self.ctx.h, n, ds, ptrs, pws, phs, _ptr(primary), viewports, self.blend_ratio, stream))


class TonemapStage:
    """tonemap_stage (filmic default, exposure 1, gamma 2.2; alpha grid only when not headless)."""

    def __init__(self, ctx: Context, op=TONEMAP_FILMIC, exposure=1.0, gamma=2.2, alpha_grid_background=False):
        self.ctx = ctx
        self.info = TonemapInfoC(op, exposure, gamma, 16 if alpha_grid_background else 0)

    def run(self, src, dst, width, height, layers=1, stream=None):
        check(_lib.lib().trhip_tonemap(self.ctx.h, _ptr(src), _ptr(dst), width, height, layers, C.byref(self.info), stream))


class _FrameSlot:
    """What one frame in flight owns: its stage (path buffers, counters), its images and the stream it is ordered on."""

    def __init__(self):
        self.pt = None
        self.color = None
        self.display = None
        self.stream = None
        self.fused_info = None


class RtRenderer:
    """rt_renderer<path_tracer_stage> for one rank of an N-GPU job.

    Each rank renders its share (`get_device_distribution_params`) into its own target; the display
    rank (0) owns the full-size image.  `render()` = scene update -> ray tracer -> transfer -> stitch ->
    tonemap (src/rt_renderer.cc:84-133).  Transfers use torch.distributed (backend "nccl" = RCCL over
    xGMI): non-display ranks `send` their partial, the display rank `recv`s it straight into device
    memory and runs the stitch kernel - replacing the GPU->pinned host->GPU copies of
    src/device_transfer.cc:21-347.

    `frames_in_flight` (MAX_FRAMES_IN_FLIGHT = 2 in the reference, src/context.hh:26): that many frame slots, each with its
    own stage, images and stream; frame i goes to slot i mod F and starts while the previous frames are still running, so
    the tails of one frame's kernels are filled by the next frame.  The exchange between ranks, the stitch and the tonemap
    of a multi-GPU frame stay on the default stream (= torch's current stream, where RCCL orders itself); only the path
    tracing runs ahead on the slot streams.
    """

    def __init__(self, ctx: Context, scene: SceneDesc, options: PtOptionsC, size, strategy=DISTRIBUTION_SCANLINE,
                 rank=0, world_size=1, viewports=1, tonemap: Optional[dict] = None, accumulate=False, use_torch=None,
                 shard="pixels", frames_in_flight=1, stage_cls=None, exchange=None, frames_per_launch=1):
        """`shard`: what the ranks divide among themselves - "pixels" (the reference's distribution strategies, partial frames
        stitched on rank 0), "views" (viewport v on rank v mod N; nothing is exchanged before output) or "samples" (every
        rank renders samples_per_pixel / N samples of every pixel; one reduce to rank 0).  SURVEY.md section 8(e).
        `exchange`: what carries the partial frames of a pixel-sharded job to rank 0 - None = torch.distributed (RCCL), or a
        transfer.LocalExchange shared by the ranks of one process (device-to-device copies on the default stream).
        `frames_per_launch`: B > 1 makes every render() call B consecutive frames (trhip_pt_set_frame_batch): the images grow B
        layer groups, frame-major, and everything after the path tracing - exchange, stitch, tonemap - handles the B frames in
        one go.  For frames that do not accumulate; what the ranks of a pixel-sharded job use, whose launches are small."""
        if shard not in ("pixels", "views", "samples"):
            raise ValueError("shard must be pixels, views or samples")
        if frames_in_flight < 1:
            raise ValueError("frames_in_flight must be >= 1")
        if frames_in_flight > 1 and accumulate:
            raise ValueError("accumulating frames depend on each other: frames_in_flight must be 1")
        if frames_per_launch < 1 or (frames_per_launch > 1 and (accumulate or shard != "pixels")):
            raise ValueError("frames_per_launch > 1 batches independent frames of a pixel-sharded (or single-device) renderer")
        if shard == "samples" and accumulate and world_size > 1:
            # the reduce sums the ranks' running means into rank 0's target in place: a second accumulated frame would blend
            # new samples into an already reduced mean
            raise ValueError("sample sharding reduces into the colour target: it cannot accumulate across frames")
        self.ctx, self.opt, self.size = ctx, options, (int(size[0]), int(size[1]))
        self.rank, self.world_size = rank, world_size
        self.exchange = exchange
        if exchange is not None:
            if shard != "pixels":
                raise ValueError("an in-process exchange carries pixel shards only")
            exchange.attach(rank, ctx)
            use_torch = False if use_torch is None else use_torch
        self.shard = shard if world_size > 1 else "pixels"
        self.total_viewports = viewports
        if self.shard == "views":
            from .transfer import shard_viewports
            viewports = len(shard_viewports(viewports, rank, world_size))
        self.frames_per_launch = frames_per_launch
        self.frame_viewports = viewports                    # layers of one frame
        viewports = viewports * frames_per_launch           # layers of one launch: every buffer, transfer, stitch and tonemap below
        self.viewports = viewports
        self.strategy = DISTRIBUTION_DUPLICATE if (world_size == 1 or self.shard != "pixels") else strategy   # src/tauray.cc:519-521
        self.accumulate = accumulate
        self.scene_update = SceneStage(ctx, scene)
        if self.shard == "pixels":
            workloads = [1.0 / world_size] * world_size
            self.dists = self._device_dists(workloads)
        else:       # every rank owns full-size images
            self.dists = [DistributionParams(self.size, DISTRIBUTION_DUPLICATE, 0, 1, True)] * world_size
        self.dist = self.dists[rank]
        if self.shard == "samples":
            if options.samples_per_pixel % world_size or (options.samples_per_pixel // world_size) % options.samples_per_pass:
                raise ValueError("sample sharding needs samples_per_pixel divisible by the device count (and the share by samples_per_pass)")
            options = copy_options(options, samples_per_pixel=options.samples_per_pixel // world_size)
            self.opt = options
        tw, th = get_distribution_target_size(self.dist)
        self.target_size = (tw, th)
        self.use_torch = (world_size > 1) if use_torch is None else use_torch
        self._torch = None
        if self.use_torch:
            import torch
            self._torch = torch
        self.frames_in_flight = frames_in_flight
        self.tonemap = TonemapStage(ctx, **(tonemap or {}))
        # rt_renderer on one device has nothing between path_tracer_stage and tonemap_stage: the stage writes the display image itself
        # (trhip_pt_set_fused_tonemap: the same bits without the second pass over the frame); TRHIP_FUSED_TONEMAP=0 keeps the stage
        self.fused_tonemap = (world_size == 1 and (stage_cls is None or stage_cls is PathTracerStage) and viewports > 0
                              and hasattr(_lib.lib(), "trhip_pt_set_fused_tonemap") and os.environ.get("TRHIP_FUSED_TONEMAP", "1") != "0")
        self.slots = []
        for k in range(frames_in_flight):
            slot = _FrameSlot()
            slot.pt = (stage_cls or PathTracerStage)(ctx, self.scene_update, options, self.dist)   # rt_renderer<Pipeline>: path_tracer_stage or direct_stage
            if self.shard == "views":
                slot.pt.set_shard(viewport_base=rank, viewport_stride=world_size)
            elif self.shard == "samples":
                slot.pt.set_shard(sample_base=rank, sample_stride=world_size)
            if frames_per_launch > 1:
                slot.pt.set_frame_batch(frames_per_launch)
            if frames_in_flight > 1:
                slot.pt.set_frame_slots(frames_in_flight)    # the frames in flight fill the chip between them: the stage picks one lane (two with two slots)
                slot.stream = ctx.create_stream()
            slot.color = self._alloc_color(viewports, tw, th)
            if self.fused_tonemap:
                slot.display = self._alloc_display(viewports)
                slot.fused_info = None       # what the stage was last told (bytes of the tonemap info), None = off
            self.slots.append(slot)
        self.current = self.slots[0]
        self.stitch = StitchStage(ctx, self.size) if (world_size > 1 and self.shard == "pixels") else None
        self.all_views = None
        self.recv_buffers = {}
        self.accumulated_frames = 0
        self.frame_index = 0

    # the stage / images of the most recent frame (the only ones there are with frames_in_flight = 1)
    @property
    def ray_tracer(self) -> PathTracerStage:
        return self.current.pt

    @property
    def color(self):
        return self.current.color

    @property
    def display(self):
        return self.current.display

    def _alloc_color(self, viewports, tw, th):
        if self.use_torch:
            return self._torch.zeros((viewports, th, tw, 4), dtype=self._torch.float32, device=f"cuda:{self.ctx.hip_device}")
        return self.ctx.alloc(max(viewports * tw * th, 1) * 16).zero()

    def _alloc_display(self, viewports):
        w, h = self.size
        if self.use_torch:
            return self._torch.empty((viewports, h, w, 4), dtype=self._torch.float32, device=f"cuda:{self.ctx.hip_device}")
        return self.ctx.alloc(viewports * w * h * 16)

    def _device_dists(self, ratios) -> List[DistributionParams]:
        out, cumulative = [], 0.0
        for i in range(self.world_size):
            ratio = min(max(ratios[i], 0.0), 1.0 - cumulative)
            out.append(get_device_distribution_params(self.size, self.strategy, cumulative, ratio, i, self.world_size, i == 0))
            cumulative += ratio
        return out

    def set_scene(self, scene: SceneDesc):
        self.sync()
        self.scene_update.set_scene(scene)

    def program(self) -> dict:
        """The shading program of this rank's stage (PathTracerStage.program)."""
        return self.slots[0].pt.program()

    def check_same_program(self, allgather=None):
        """All ranks of a job render with the same shading program, or none renders: the reference compiles one pipeline per stage from the
        options and every device gets it (src/path_tracer_stage.cc:30-116).  Here a rank whose run-time compilation failed ("renders with the
        general kernels and says so"), that runs under TRHIP_SPECIALIZE=0 or loads another build of libtrhip.so would shade its strips with
        other kernels - at the default arithmetic another implementation inside Vulkan's accuracy, i.e. strips that differ from their
        neighbours' in the last bits.  `allgather(bytes) -> [bytes of rank 0, ...]`: the job's transport for small blobs (the one
        comm.Ipc takes); default: torch.distributed.all_gather_object when a process group exists.  Every rank calls this before the first
        frame (bench.py, tests/test_multi_rank_gloo.py); raises RuntimeError on every rank, naming both programs."""
        if self.world_size == 1:
            return self.program()
        mine = self.program()
        blob = mine["identity"].to_bytes(8, "little") + mine["key"].encode()
        if allgather is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("check_same_program: no transport (pass allgather, or initialise torch.distributed)")
            every = [None] * self.world_size
            dist.all_gather_object(every, blob)
        else:
            every = allgather(blob)
        for r, b in enumerate(every):
            if b[:8] != blob[:8]:
                raise RuntimeError(f"the ranks of this job would render with different shading programs: rank {self.rank} {{{mine['key']}}} but rank {r} "
                                   f"{{{b[8:].decode(errors='replace')}}} (same libtrhip.so, kernel cache and TRHIP_* environment on every rank?)")
        return mine

    def sync(self):
        """Waits for every frame in flight.  An exchange that can tell that a frame is incomplete (the copy-engine exchange: a device-side
        wait for a peer that gave up) says so here, before anybody looks at the frame."""
        for slot in self.slots:
            if slot.stream is not None:
                self.ctx.sync(slot.stream)
        self.ctx.sync()
        if self.exchange is not None and hasattr(self.exchange, "check"):
            self.exchange.check()

    def reset_accumulation(self, reset_sample_counter=False):
        for slot in self.slots:
            slot.pt.reset_accumulated_samples()
            if reset_sample_counter:
                slot.pt.reset_sample_counter()
        if reset_sample_counter:
            self.frame_index = 0
        self.accumulated_frames = 0

    def set_profiling(self, count_work=False, detailed_timing=False):
        for slot in self.slots:
            slot.pt.set_profiling(count_work, detailed_timing)

    def reset_counters(self):
        self.sync()
        for slot in self.slots:
            slot.pt.reset_counters()

    def counters(self) -> dict:
        """Work counters summed over the frame slots."""
        self.sync()
        total = {}
        for slot in self.slots:
            for k, v in slot.pt.counters().items():
                total[k] = max(total.get(k, 0), v) if k == "stack_overflows" else total.get(k, 0) + v
        return total

    def timings(self) -> dict:
        self.sync()
        total = {}
        for slot in self.slots:
            for k, v in slot.pt.timings().items():
                total[k] = total.get(k, 0) + v
        return total

    def phase_counters(self) -> dict:
        self.sync()
        total = {}
        for slot in self.slots:
            for k, v in slot.pt.phase_counters().items():
                total[k] = [a + b for a, b in zip(total[k], v)] if (k in total and isinstance(v, list)) else (total.get(k, 0) + v if not isinstance(v, list) else v)
        return total

    def path_tracing_ms(self) -> float:
        """The "path tracing" timer the load balancer reads (src/load_balancer.cc:17,25): the last frame of every slot, averaged.
        Waits for the slots."""
        self.sync()
        t = [slot.pt.timings()["path_tracing_ms"] for slot in self.slots]
        return sum(t) / len(t)

    def set_device_workloads(self, ratios):
        """rt_renderer::set_device_workloads (src/rt_renderer.cc:135-183): only for shuffled strips."""
        if self.strategy in (DISTRIBUTION_SCANLINE, DISTRIBUTION_DUPLICATE):
            return
        self.sync()
        self.dists = self._device_dists(ratios)
        self.dist = self.dists[self.rank]
        new_size = get_distribution_target_size(self.dist)
        for slot in self.slots:
            slot.pt.reset_distribution_params(self.dist)
            if self.rank != 0:
                slot.pt.reset_accumulated_samples()
                if new_size != self.target_size:
                    # A non-primary target has the size of the share (the partial frame that travels is the whole image):
                    # a new share is a new image.  The reference allocates get_distribution_target_max_size once instead
                    # (src/rt_renderer.cc init_resources); either way the kernels are bounded by the allocated size.
                    slot.color = self._alloc_color(self.viewports, *new_size)
        self.target_size = new_size
        self.recv_buffers.pop("ops", None)      # the display rank's receive list is rebuilt for the new shapes
        if self.accumulate and self.stitch is not None:
            # the other devices start over with one sample: blend it into what has accumulated (src/rt_renderer.cc:176-181)
            self.stitch.set_blend_ratio(1.0 / (self.accumulated_frames + 1))

    def _sync_fused_tonemap(self, slot, want: bool):
        """The stage's copy of the tonemap parameters follows self.tonemap.info: an edit of exposure / operator / gamma between frames
        takes effect on the next frame, as it does with the tonemap stage of a multi-device renderer; render(tonemap=False) switches
        the display write off for that frame."""
        now = bytes(self.tonemap.info) if want else None
        if now != slot.fused_info:
            slot.pt.set_fused_tonemap(slot.display if want else None, self.tonemap.info if want else None)
            slot.fused_info = now

    def render_partial(self, stream=None, tonemap=True):
        """The path-tracing part of the next frame on its slot (`stream` overrides the slot's stream)."""
        slot = self.slots[(self.frame_index // self.frames_per_launch) % self.frames_in_flight]
        self.current = slot
        if self.fused_tonemap:
            self._sync_fused_tonemap(slot, tonemap)
        if not self.accumulate:
            slot.pt.reset_accumulated_samples()
        if self.frames_in_flight > 1 or self.frames_per_launch > 1:
            slot.pt.set_frame_counter(self.frame_index)      # one stage per slot: slot k renders frames k, k + F, ... (B at a time)
        self.frame_index += self.frames_per_launch
        if self.viewports > 0:      # a view shard can be empty (more devices than views)
            slot.pt.run(slot.color, self.viewports, stream if stream is not None else slot.stream)

    def transfer_and_stitch(self, own_slot=None):
        """device_transfer + stitch_stage over RCCL: gather partial frames on rank 0 (default stream).  `own_slot`: the display
        rank's slot whose path tracing the stitch (not the receives) has to wait for."""
        if self.world_size == 1:
            return
        if self.exchange is not None:
            partials = self.exchange.gather_to_display(self.color, self.dists, self.rank, self.world_size, self.viewports, self.recv_buffers, self.ctx)
        else:
            from .transfer import gather_to_display
            partials = gather_to_display(self.color, self.dists, self.rank, self.world_size, self.viewports, self.recv_buffers)
        if own_slot is not None and own_slot.stream is not None:
            self.ctx.stream_wait(None, own_slot.stream)
        if partials:
            peers = sorted(partials)
            self.stitch.run_all([self.dists[r] for r in peers], [partials[r] for r in peers], self.color, self.viewports)
        if self.rank == 0:
            self.stitch.set_blend_ratio(1.0)

    def render(self, tonemap=True, gather_views=False):
        self.render_partial(tonemap=tonemap)
        slot = self.current
        if self.world_size == 1:
            if tonemap and not self.fused_tonemap:
                self.post_process(slot.stream)       # the whole frame stays on its slot's stream
            self.accumulated_frames += 1
            return
        # Several ranks.  Everything after the path tracing runs on the default stream, which is also torch's current
        # stream: RCCL orders itself after the kernels enqueued there.  With frames in flight the default stream first
        # waits for this slot's path tracing, and the slot's stream afterwards waits for the default stream, so that the
        # next frame of this slot does not overwrite images that are still being sent, stitched or tonemapped.
        # The display rank of a pixel-sharded frame receives into buffers of its own: its receives need not wait for its own
        # path tracing, only the stitch into its image does.
        display_of_pixels = self.shard == "pixels" and self.rank == 0
        if slot.stream is not None and not display_of_pixels:
            self.ctx.stream_wait(None, slot.stream)
        if self.shard == "views":
            # every rank finishes its own views (tonemap is per pixel); `gather_views` ships them to the writer on rank 0
            if tonemap and self.viewports > 0:
                self.post_process()
            if gather_views:
                from .transfer import gather_views_to_display
                src = self.display if tonemap else self.color
                self.all_views = gather_views_to_display(src, self.total_viewports, self.rank, self.world_size, self.all_views)
        elif self.shard == "samples":
            from .transfer import reduce_samples_to_display
            reduce_samples_to_display(self.color, self.rank, self.world_size)
            if tonemap and self.rank == 0:
                self.post_process()
        else:
            self.transfer_and_stitch(slot if display_of_pixels else None)
            if tonemap and self.rank == 0:
                self.post_process()
        if slot.stream is not None:
            self.ctx.stream_wait(slot.stream, None)
        self.accumulated_frames += 1

    def post_process(self, stream=None):
        w, h = self.size
        if self.viewports == 0:
            return
        slot = self.current
        if slot.display is None:
            slot.display = self._alloc_display(self.viewports)
        self.tonemap.run(slot.color, slot.display, w, h, self.viewports, stream)

    def download(self, which="color") -> np.ndarray:
        """The most recent frame's partial colour target or tonemapped display image."""
        self.sync()
        buf = self.color if which == "color" else self.display
        if which == "color":
            tw, th = self.target_size
        else:
            tw, th = self.size
        if self.use_torch:
            self._torch.cuda.synchronize()
            return buf.cpu().numpy()
        return buf.download((self.viewports, th, tw, 4), np.float32)

    def close(self):
        if not self.slots:
            return
        self.sync()
        for slot in self.slots:
            slot.pt.close()
            if slot.stream is not None:
                self.ctx.destroy_stream(slot.stream)
                slot.stream = None
        self.slots = []

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass
