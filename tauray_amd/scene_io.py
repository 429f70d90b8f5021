"""Binary scene dump (.trsc): the flattened SceneDesc as raw Appendix-A arrays, so the C++ host layer
(include/tauray_hip.hh, tauray_amd/host/tauray_hip_cli.cc) can run without a glTF parser.

Layout (little endian): magic 'TRSC', u32 version = 1, then for each section a u64 byte count followed by
the bytes: instances(288 B), spans(16 B), vertices(48 B), indices(u32), point_lights(64 B),
directional_lights(32 B), texture_infos(16 B), texels(RGBA8), envmap(RGBA32F), alias_table(16 B),
cameras(320 B), non_opaque(u8); then u32 envmap_width, u32 envmap_height, f32[4] environment_factor,
u32 gather_emissive_triangles, u32 projection.
"""
import struct

import numpy as np

from .scene import SceneDesc, build_alias_table


def write_scene_dump(scene: SceneDesc, path: str):
    infos, texels = scene.texture_table()
    env = np.zeros(0, dtype=np.float32)
    at = np.zeros(0, dtype=np.uint8)
    ew = eh = 0
    if scene.envmap is not None:
        env = np.ascontiguousarray(scene.envmap, dtype=np.float32)
        eh, ew = env.shape[:2]
        at = build_alias_table(env)
    sections = [scene.instances, scene.spans, scene.vertices, scene.indices, scene.point_lights, scene.directional_lights,
                infos, texels, env, at, scene.camera_data(), scene.potentially_transparent().astype(np.uint8)]
    with open(path, "wb") as f:
        f.write(b"TRSC")
        f.write(struct.pack("<I", 1))
        for a in sections:
            b = np.ascontiguousarray(a).tobytes()
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
        f.write(struct.pack("<II4fII", ew, eh, *[float(x) for x in scene.environment_factor],
                            1 if getattr(scene, "tri_light_count", 0) > 0 else 0,
                            scene.cameras[0].projection if scene.cameras else 0))
