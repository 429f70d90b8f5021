"""ctypes binding of libtrhip.so (include/trhip.h).  There is no CPU fallback: if the HIP
library is missing or no GPU is present the calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

# Hardware queues the HIP runtime spreads its streams over (default 4): frame slots beyond three only pay off with more
# (DESIGN.md section 5).  Read by the runtime when it starts, so it has to be in the environment before the library loads.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Memory shared between processes (RCCL's buffers, trhip_ipc_*): the host driver of the target boxes supports dmabuf IPC only.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRHIP_LIB", os.path.join(_HERE, "libtrhip.so"))   # TRHIP_LIB: A/B builds while tuning


class TrhipError(RuntimeError):
    pass


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


class AccelInfoC(C.Structure):
    _fields_ = [("triangle_count", C.c_uint32), ("node_count", C.c_uint32), ("tri_light_count", C.c_uint32),
                ("build_ms", C.c_float), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("node_bytes", C.c_uint32), ("leaf_count", C.c_uint32)]


class PtOptionsC(C.Structure):
    """== path_tracer_stage::options (reference src/path_tracer_stage.hh:13-30), flattened."""
    _fields_ = [
        ("max_bounces", C.c_int32), ("min_ray_dist", C.c_float), ("rng_seed", C.c_uint32), ("sampler", C.c_int32),
        ("samples_per_pixel", C.c_int32), ("samples_per_pass", C.c_int32), ("projection", C.c_int32),
        ("film", C.c_int32), ("film_radius", C.c_float), ("mis_mode", C.c_int32),
        ("russian_roulette_delta", C.c_float), ("indirect_clamping", C.c_float), ("regularization_gamma", C.c_float),
        ("depth_of_field", C.c_int32), ("nee_point", C.c_float), ("nee_directional", C.c_float),
        ("nee_envmap", C.c_float), ("nee_triangles", C.c_float), ("bounce_mode", C.c_int32),
        ("tri_light_mode", C.c_int32), ("hide_lights", C.c_int32), ("use_white_albedo_on_first_bounce", C.c_int32),
        ("transparent_background", C.c_int32), ("pre_transformed_vertices", C.c_int32)]


class DistributionC(C.Structure):
    _fields_ = [("size_x", C.c_uint32), ("size_y", C.c_uint32), ("strategy", C.c_int32),
                ("index", C.c_uint32), ("count", C.c_uint32), ("primary", C.c_uint32)]


class CountersC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("closest_rays", "shadow_rays", "node_visits", "tri_tests", "alpha_tests",
                                          "surface_hits", "stack_overflows")]


class TimingsC(C.Structure):
    _fields_ = ([(n, C.c_float) for n in ("path_tracing_ms", "trace_closest_ms", "trace_shadow_ms", "shade_ms", "raygen_ms", "resolve_ms")]
                + [(n, C.c_uint32) for n in ("trace_closest_launches", "trace_shadow_launches", "shade_launches", "frames")])


class PhaseCountersC(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("lane_node_phases", "lane_tri_phases", "quad_node_phases", "quad_tri_phases",
                                           "lane_node_phases_le16", "lane_node_visits_le16")] + [("lane_node_phase_hist", C.c_uint64 * 8),
                                                                                                 ("closest_node_visits", C.c_uint64)])


class ProgramInfoC(C.Structure):
    """trhip_program_info: which shading program renders a stage (kind 0 general / 1 command-line set ahead of time / 2 compiled), its arithmetic, identity."""
    _fields_ = [("kind", C.c_int32), ("ieee", C.c_int32), ("identity", C.c_uint64), ("key", C.c_char * 240)]


class PtTargetsC(C.Structure):
    """trhip_pt_targets: device images, None = not requested."""
    _fields_ = [(n, C.c_void_p) for n in ("color", "diffuse", "reflection", "albedo", "material", "normal", "pos", "instance_id", "screen_motion")]


class TonemapInfoC(C.Structure):
    _fields_ = [("op", C.c_int32), ("exposure", C.c_float), ("gamma", C.c_float), ("alpha_grid_background", C.c_int32)]


# every symbol include/trhip.h declares: (name, restype, argtypes)
_vp, _u32, _i, _f = C.c_void_p, C.c_uint32, C.c_int, C.c_float
SYMBOLS = {
    "trhip_image_decode": (_i, [_vp, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.POINTER(C.c_uint8))]),
    "trhip_image_free": (None, [C.POINTER(C.c_uint8)]),
    "trhip_exr_decode": (_i, [_vp, C.c_size_t, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.POINTER(C.c_float))]),
    "trhip_exr_encode": (_i, [_vp, _u32, _u32, _i, _i, _i, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]),
    "trhip_exr_free": (None, [_vp]),
    "trhip_device_create": (_i, [_i, C.POINTER(_vp)]),
    "trhip_device_destroy": (None, [_vp]),
    "trhip_last_error": (C.c_char_p, []),
    "trhip_malloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "trhip_image_decode_texels": (_i, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8))]),
    "trhip_free": (_i, [_vp, _vp]),
    "trhip_upload": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "trhip_download": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "trhip_memset": (_i, [_vp, _vp, _i, C.c_size_t, _vp]),
    "trhip_sync": (_i, [_vp, _vp]),
    "trhip_copy_peer": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "trhip_scene_upload": (_i, [_vp, C.POINTER(SceneDescC)]),
    "trhip_scene_update_cameras": (_i, [_vp, _vp, _u32]),
    "trhip_scene_set_previous_cameras": (_i, [_vp, _vp, _u32]),
    "trhip_scene_update_instances": (_i, [_vp, _vp, _u32]),
    "trhip_scene_build_accel": (_i, [_vp, C.POINTER(AccelInfoC)]),
    "trhip_scene_refit_accel": (_i, [_vp, C.POINTER(AccelInfoC)]),
    "trhip_pt_set_frame_batch": (_i, [_vp, C.c_uint32]),
    "trhip_scene_update_lights": (_i, [_vp, _vp, C.c_uint32, _vp, C.c_uint32]),
    "trhip_scene_set_build_mode": (_i, [_vp, _i]),
    "trhip_stitch_batch": (_i, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _u32, C.c_float, _vp]),
    "trhip_stream_create": (_i, [_vp, C.POINTER(C.c_void_p)]),
    "trhip_stream_destroy": (_i, [_vp, _vp]),
    "trhip_stream_pipe_class": (_i, [_vp, _vp, C.POINTER(C.c_int32)]),
    "trhip_device_get_info": (_i, [_vp, _vp]),
    "trhip_build_id": (C.c_uint64, []),
    "trhip_stream_wait": (_i, [_vp, _vp, _vp]),
    "trhip_stream_wait_peer": (_i, [_vp, _vp, _vp, _vp]),
    "trhip_pt_set_frame_counter": (_i, [_vp, _u32]),
    "trhip_pt_set_lanes": (_i, [_vp, C.c_int]),
    "trhip_pt_set_frame_slots": (_i, [_vp, C.c_int]),
    "trhip_pt_set_fused_tonemap": (_i, [_vp, _vp, _vp]),
    "trhip_pt_get_lane_pipes": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "trhip_pt_set_shading_arithmetic": (_i, [_vp, C.c_int]),
    "trhip_pt_set_specialization": (_i, [_vp, C.c_int]),
    "trhip_pt_precompile": (_i, [C.POINTER(PtOptionsC), C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "trhip_kernel_cache_dir": (C.c_char_p, []),
    "trhip_pt_set_shard": (_i, [_vp, _u32, _u32, _u32, _u32]),
    "trhip_scene_set_skin": (_i, [_vp, _u32, _vp, _vp, _u32]),
    "trhip_scene_skin": (_i, [_vp, _u32, _vp, _u32]),
    "trhip_scene_get_vertices": (_i, [_vp, _u32, _vp, _u32]),
    "trhip_scene_get_tri_lights": (_i, [_vp, _vp, _u32]),
    "trhip_pt_create": (_i, [_vp, C.POINTER(PtOptionsC), C.POINTER(_vp)]),
    "trhip_direct_create": (_i, [_vp, C.POINTER(PtOptionsC), C.POINTER(_vp)]),
    "trhip_pt_destroy": (None, [_vp]),
    "trhip_pt_set_distribution": (_i, [_vp, C.POINTER(DistributionC)]),
    "trhip_pt_reset_accumulation": (_i, [_vp, _i]),
    "trhip_pt_render": (_i, [_vp, _vp, _u32, _u32, _u32, _vp]),
    "trhip_pt_render_targets": (_i, [_vp, C.POINTER(PtTargetsC), _u32, _u32, _u32, _vp]),
    "trhip_pt_set_profiling": (_i, [_vp, _i, _i]),
    "trhip_pt_get_counters": (_i, [_vp, C.POINTER(CountersC)]),
    "trhip_pt_reset_counters": (_i, [_vp]),
    "trhip_pt_get_timings": (_i, [_vp, C.POINTER(TimingsC)]),
    "trhip_pt_get_phase_counters": (_i, [_vp, C.POINTER(PhaseCountersC)]),
    "trhip_pt_get_program": (_i, [_vp, C.POINTER(ProgramInfoC)]),
    "trhip_calibrate_valu": (_i, [_vp, C.POINTER(C.c_float)]),
    "trhip_calibrate_l1": (_i, [_vp, C.POINTER(C.c_float)]),
    "trhip_feature_render": (_i, [_vp, _i, C.POINTER(DistributionC), _i, _u32, _f, C.POINTER(_f), _vp, _u32, _u32, _vp]),
    "trhip_trace_closest": (_i, [_vp, _u32, _vp, _vp, _i, _vp, _vp]),
    "trhip_trace_shadow": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "trhip_stitch": (_i, [_vp, C.POINTER(DistributionC), _vp, _u32, _u32, _vp, _u32, _f, _vp]),
    "trhip_tonemap": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, C.POINTER(TonemapInfoC), _vp]),
}

_LIB = None


def lib():
    """Loads libtrhip.so; raises TrhipError when it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise TrhipError(f"{LIB_PATH} is missing: build it with `make -C tauray_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                # the tree's own library exports everything (tests/test_abi_and_host.py); a library named by TRHIP_LIB may be an older
                # build kept for an A/B (tools/ab_libs.sh): calls it does not have fail when they are made
                if "TRHIP_LIB" not in os.environ:
                    raise
                continue
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise TrhipError(lib().trhip_last_error().decode("utf-8", "replace"))
