"""Host mirror of the reference's distribution_strategy math (src/distribution_strategy.{hh,cc})
and the load balancer (src/load_balancer.cc)."""
from __future__ import annotations

import math
from dataclasses import dataclass

DISTRIBUTION_DUPLICATE = 0
DISTRIBUTION_SCANLINE = 1
DISTRIBUTION_SHUFFLED_STRIPS = 2


@dataclass
class DistributionParams:       # distribution_params (distribution_strategy.hh:21-28)
    size: tuple = (0, 0)
    strategy: int = DISTRIBUTION_SCANLINE
    index: int = 0
    count: int = 1
    primary: bool = True


def get_distribution_render_size(p: DistributionParams):   # :33-49
    if p.strategy == DISTRIBUTION_DUPLICATE:
        return p.size
    if p.strategy == DISTRIBUTION_SCANLINE:
        return (p.size[0], (p.size[1] - p.index + p.count - 1) // p.count)
    return (p.count, 1)


def get_distribution_target_size(p: DistributionParams):   # :6-19
    if p.primary:
        return p.size
    if p.strategy == DISTRIBUTION_SHUFFLED_STRIPS:
        return (p.size[0], (p.count + p.size[0] - 1) // p.size[0])
    return get_distribution_render_size(p)


def get_distribution_target_max_size(p: DistributionParams):   # :21-31, with the padded id range (see include/tauray_hip.hh)
    if p.strategy == DISTRIBUTION_SHUFFLED_STRIPS:
        # the regions are padded to a common size, so a device with (nearly) the whole frame is handed more ids than there are pixels
        # and its partial image is a row taller than the frame; the reference's Vulkan image drops those stores, a buffer must hold them
        n = p.size[0] * p.size[1]
        b = 31
        while (n >> b) < 128 and b > 0:
            b -= 1
        padded = ((n + (1 << b) - 1) >> b) << b
        return (p.size[0], (padded + p.size[0] - 1) // p.size[0])
    return get_distribution_target_size(p)


def get_ray_count(p: DistributionParams):   # :51-60
    if p.strategy == DISTRIBUTION_SHUFFLED_STRIPS:
        return (p.count, 1)
    return get_distribution_render_size(p)


def calculate_shuffled_strips_b(size) -> int:   # :62-69
    n = (size[0] * size[1]) & 0xFFFFFFFF
    b = 31
    while (n >> b) < 128 and b > 0:
        b -= 1
    return b


def get_region_size(image_size: int, b: int) -> int:   # :73-77
    n_regions = 1 << b
    return (image_size + n_regions - 1) // n_regions


def calculate_shuffled_strips_pixels_per_device(size, max_ratio: float) -> int:   # :79-83
    b = calculate_shuffled_strips_b(size)
    import numpy as np
    # the reference evaluates `max_ratio * region_size * (1 << b)` in float (max_ratio is a float parameter)
    return int(math.ceil(float(np.float32(max_ratio) * np.float32(get_region_size(size[0] * size[1], b)) * np.float32(1 << b))))


def get_device_distribution_params(full_image_size, strategy, workload_offset, workload_size, device_index,
                                   device_count, primary) -> DistributionParams:   # :85-126
    d = DistributionParams(strategy=strategy)
    if strategy in (DISTRIBUTION_DUPLICATE, DISTRIBUTION_SCANLINE):
        d.size = tuple(full_image_size)
        d.index = device_index
        d.count = device_count
        d.primary = primary
    else:
        before = calculate_shuffled_strips_pixels_per_device(full_image_size, workload_offset)
        after = calculate_shuffled_strips_pixels_per_device(full_image_size, workload_offset + workload_size)
        d.size = tuple(full_image_size)
        d.index = before
        d.count = after - before
        d.primary = primary
    return d


def permute_region_id(i: int, size, b: int) -> int:   # shader/rt.glsl:170-179
    region_size = ((size[0] * size[1]) + (1 << b) - 1) >> b
    region_id = i // region_size
    k = int(format(region_id & 0xFFFFFFFF, "032b")[::-1], 2) >> (32 - b) if b > 0 else 0
    return k * region_size + i % region_size


class LoadBalancer:
    """load_balancer (src/load_balancer.cc): EMA of per-device speed on the "path tracing" timer."""

    def __init__(self, device_count, initial_weights=None):
        w = list(initial_weights) if initial_weights else [0.0] * device_count
        w = (w + [0.0] * device_count)[:device_count]
        s = sum(w)
        add = 0.0
        if s == 0:
            add, s = 1.0, float(device_count)
        self.workloads = [(max(x, 0.0) + add) / s for x in w]

    def update(self, path_tracing_times):
        # speed = share / time; a zero time (empty share, missing timer) is an infinite speed, the sum is then not finite and
        # the update is skipped, as in src/load_balancer.cc:12-32
        def speed(w, t):
            if t == 0:
                return math.inf if w > 0 else (math.nan if w == 0 else -math.inf)
            return w / t
        speeds = [speed(w, t) for w, t in zip(self.workloads, path_tracing_times)]
        sum_speed = sum(max(v, 0.0) if not math.isnan(v) else v for v in speeds)
        if sum_speed > 0 and math.isfinite(sum_speed):
            self.workloads = [w * 0.9 + v / sum_speed * 0.1 for w, v in zip(self.workloads, speeds)]
        return self.workloads
