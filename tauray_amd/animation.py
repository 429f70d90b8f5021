"""glTF animation clips for the scene loaders: tr::animation / animation_controller / animated
(src/animation.{hh,cc,tcc}) and the part of src/scene.cc that plays them (play / update / is_playing, :213-244).

A clip holds up to three tracks per node - position, scaling, orientation - sampled at integer microsecond ticks
(src/gltf.cc:167-190 rounds the file's seconds).  `SceneAnimator` is what `tauray --animation[=name] --framerate F` does
to a loaded scene per frame (src/tauray.cc:252-253, 1052-1092): advance every animated node's timer, rebuild the global
transforms below it, and hand the renderer new instance records, cameras and joint matrices
(`SceneStage.update_instances / update_cameras / pose`), and the records of the punctual lights that hang on moving nodes
(`trhip_scene_update_lights`).
"""
from __future__ import annotations

import bisect
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import scene as S

LINEAR, STEP, CUBICSPLINE = 0, 1, 2
INTERPOLATION = {"LINEAR": LINEAR, "STEP": STEP, "CUBICSPLINE": CUBICSPLINE}


def cubic_spline(p1, m1, p2, m2, t):
    """src/math.tcc:24-35 (float arithmetic in the reference; the tracks feed host-side matrices only)."""
    t = np.float32(t)
    t2 = t * t
    t3 = t2 * t
    tmp = np.float32(2) * t3 - np.float32(3) * t2
    return (tmp + 1) * p1 + (t3 - 2 * t2 + t) * m1 + (-tmp) * p2 + (t3 - t2) * m2


def slerp(a, b, t):
    """glm::slerp for quaternions stored (x, y, z, w): the shorter arc, linear when the two nearly coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    cos_theta = float(a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3])
    if cos_theta < 0:
        b, cos_theta = -b, -cos_theta
    if cos_theta > 1.0 - np.finfo(np.float32).eps:
        return a * (1.0 - t) + b * t
    angle = math.acos(cos_theta)
    return (math.sin((1.0 - t) * angle) * a + math.sin(t * angle) * b) / math.sin(angle)


@dataclass
class Track:
    """std::vector<animation::sample<T>> + its interpolation."""
    interpolation: int
    timestamps: List[int]            # microsecond ticks, ascending
    data: np.ndarray                 # (n, 3) or (n, 4)
    in_tangent: Optional[np.ndarray] = None
    out_tangent: Optional[np.ndarray] = None

    def sample(self, time: int, quaternion: bool = False):
        """animation::interpolate (src/animation.tcc:39-77)."""
        i = bisect.bisect_right(self.timestamps, time)       # std::upper_bound
        if i == len(self.timestamps):
            return self.data[-1].astype(np.float64)
        if i == 0:
            return self.data[0].astype(np.float64)
        frame_ticks = np.float32(self.timestamps[i] - self.timestamps[i - 1])
        ratio = float(np.float32(time - self.timestamps[i - 1]) / frame_ticks)
        if self.interpolation == STEP:
            return self.data[i - 1].astype(np.float64)
        if self.interpolation == CUBICSPLINE and self.in_tangent is not None:
            scale = float(frame_ticks * np.float32(0.000001))       # tangents are per second
            return np.asarray(cubic_spline(self.data[i - 1].astype(np.float64), self.out_tangent[i - 1].astype(np.float64) * scale,
                                           self.data[i].astype(np.float64), self.in_tangent[i].astype(np.float64) * scale, ratio), dtype=np.float64)
        if quaternion:
            return slerp(self.data[i - 1], self.data[i], ratio)
        return self.data[i - 1].astype(np.float64) * (1.0 - ratio) + self.data[i].astype(np.float64) * ratio


@dataclass
class Animation:
    """tr::animation: one named clip of one node."""
    position: Optional[Track] = None
    scaling: Optional[Track] = None
    orientation: Optional[Track] = None

    @property
    def loop_time(self) -> int:
        """animation::determine_loop_time: the last timestamp of any track."""
        return max([t.timestamps[-1] for t in (self.position, self.scaling, self.orientation) if t is not None and t.timestamps] + [0])

    def apply(self, trs: dict, time: int):
        """animation::apply: overwrite the node's translation / scale / rotation with the tracks' values at `time`."""
        if self.position is not None:
            trs["translation"] = self.position.sample(time)
        if self.scaling is not None:
            trs["scale"] = self.scaling.sample(time)
        if self.orientation is not None:
            q = self.orientation.sample(time, quaternion=True)
            if self.orientation.interpolation == CUBICSPLINE:
                q = q / math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
            trs["rotation"] = q


def read_track(timestamps_s: np.ndarray, values: np.ndarray, interpolation: int) -> Track:
    """read_animation_accessors (src/gltf.cc:167-190): ticks = round(seconds * 1e6); three values per key (in-tangent, value,
    out-tangent) when the output accessor holds that many."""
    ts = [int(math.floor(float(np.float32(t) * np.float32(1000000)) + 0.5)) for t in timestamps_s]      # C round() of a float product
    n = len(ts)
    if len(values) >= 3 * n:
        v = values[:3 * n].reshape(n, 3, -1)
        return Track(interpolation, ts, v[:, 1].copy(), v[:, 0].copy(), v[:, 2].copy())
    return Track(interpolation, ts, values[:n].copy())


class Controller:
    """animation_controller<animated> of one node (src/animation.tcc:79-205): play / update; a queue of one clip."""

    def __init__(self, pool: Dict[str, Animation]):
        self.pool = dict(sorted(pool.items()))     # animation_pool is a std::map: alphabetical
        self.current: Optional[Animation] = None
        self.loop = False
        self.playing = False
        self.timer = 0
        self.loop_time = 0

    def play(self, name: str, loop: bool = False, use_fallback: bool = False):
        self.timer = 0
        self.current = self.pool.get(name)
        if self.current is None and use_fallback and self.pool:
            self.current = next(iter(self.pool.values()))
        self.loop_time = self.current.loop_time if self.current is not None else 0
        self.playing = self.loop_time != 0
        self.loop = loop

    def update(self, trs: dict, dt: int):
        if not self.playing:
            return
        self.timer += dt
        if self.loop:
            self.timer %= self.loop_time
        elif self.timer >= self.loop_time:     # past the end of a clip that does not loop: stop, the node keeps its last pose
            self.playing = False
            self.loop_time = 0
            self.timer = 0
            return
        self.current.apply(trs, self.timer)


@dataclass
class Node:
    """What the loader keeps of a glTF node for animation: its place in the tree, its local transform, what hangs on it."""
    parent: int
    children: List[int]
    trs: Optional[dict]              # {"translation", "rotation", "scale"}; None when the node has a matrix
    matrix: Optional[np.ndarray]
    instances: List[int] = field(default_factory=list)     # rigid instances placed by this node's global transform
    cameras: List[int] = field(default_factory=list)
    lights: List[tuple] = field(default_factory=list)      # (kind, index in its list, make(global transform) -> record)

    def local(self) -> np.ndarray:
        if self.trs is None:
            return self.matrix
        return S.trs_matrix(self.trs["translation"], self.trs["rotation"], self.trs["scale"])


class SceneAnimator:
    """play(scene, name, loop, fallback) + update(scene, dt) of src/scene.cc over a loaded SceneDesc (`desc.nodes`,
    `desc.animations`, `desc.roots` from load_glb)."""

    def __init__(self, desc):
        self.desc = desc
        self.controllers = {n: Controller(pool) for n, pool in desc.animations.items()}
        self.node_globals = dict(desc.node_globals)
        self.previous_cameras = list(desc.cameras)

    def play(self, name: str = "", loop: bool = False):
        """`--animation[=name]`: the named clip on every animated node, any clip the node has when no name is given."""
        for c in self.controllers.values():
            c.play(name, loop, use_fallback=(name == ""))

    def is_playing(self) -> bool:
        return any(c.playing for c in self.controllers.values())

    def update(self, dt_ticks: int):
        """Advance by dt (0 for the first frame, round(1e6 / framerate) afterwards: src/tauray.cc:1052,1090) and rebuild
        instances, cameras and node globals.  Returns (instances, cameras, node_globals) for the renderer; model_prev of
        every instance is what its model was before this update (src/scene_stage.cc:1092-1105)."""
        import copy
        desc = self.desc
        self.previous_cameras = [copy.copy(c) for c in desc.cameras]     # camera_pair.previous of this frame (shader/scene.glsl:176-185)
        for n, c in self.controllers.items():
            node = desc.nodes[n]
            if node.trs is not None:
                c.update(node.trs, dt_ticks)
        instances = desc.instances.copy()
        instances["model_prev"] = desc.instances["model"]
        point_lights, directional_lights = desc.point_lights.copy(), desc.directional_lights.copy()

        def visit(n, parent):
            node = desc.nodes[n]
            glob = parent @ node.local()
            self.node_globals[n] = glob
            for i in node.instances:
                instances["model"][i] = S.to_glm(glob)
                instances["model_normal"][i] = S.to_glm(np.linalg.inv(glob).T)
            for ci in node.cameras:
                desc.cameras[ci].transform = glob
            for kind, index, make in node.lights:       # point lights first, then spotlights, in one array (src/scene_stage.cc:1287-1317)
                if kind == "directional":
                    directional_lights[index] = make(glob)
                else:
                    point_lights[index + (desc.spotlight_base if kind == "spot" else 0)] = make(glob)
            for ch in node.children:
                visit(ch, glob)

        for r in desc.roots:
            visit(r, np.eye(4))
        desc.instances = instances
        desc.point_lights, desc.directional_lights = point_lights, directional_lights
        desc.node_globals = dict(self.node_globals)
        return instances, desc.cameras, self.node_globals
