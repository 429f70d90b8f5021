"""Named option sets of the path tracer: the reference's presets (data/presets/*.cfg, restricted to what path_tracer_stage reads),
the sets the parity tests walk through, and the kernel-cache warm-up that __graft_entry__.build() runs for them
(trhip_pt_precompile: the specialised shading programs are compiled ahead of time, without a GPU, so a GPU box starts warm)."""
import ctypes as C
import os

from . import _lib
from .renderer import make_options

# data/presets/quality.cfg, reference.cfg, accumulation.cfg as path_tracer_stage options (max-ray-depth d -> max_bounces d;
# blackman-harris film at the default radius; `force-double-sided`, tonemap and file type belong to other stages).
# samples_per_pixel is the preset's; a caller that times frames overrides it.
REFERENCE_PRESETS = {
    "quality": dict(film=2, max_bounces=4, samples_per_pixel=4096, sampler=0, regularization_gamma=0.1),
    "reference": dict(film=2, max_bounces=8, samples_per_pixel=16384, sampler=0, tri_light_mode=2),
    "accumulation": dict(film=2, max_bounces=5, samples_per_pixel=1, sampler=0, regularization_gamma=0.2),
}

# what tests/test_gpu_parity.py::test_path_tracer_matches_oracle renders on test.glb
NAMED_OPTION_SETS = {
    "cli-defaults-8-bounces": dict(),
    "4-bounces": dict(max_bounces=4),
    "1-bounce": dict(max_bounces=1),
    "sobol-owen": dict(max_bounces=4, sampler=1),
    "sobol-z2": dict(max_bounces=4, sampler=2),
    "sobol-z3": dict(max_bounces=4, sampler=3, samples_per_pixel=2),
    "box-film": dict(max_bounces=3, film=1),
    "blackman-harris": dict(max_bounces=3, film=2, film_radius=1.0),
    "mis-balance": dict(max_bounces=3, mis_mode=1),
    "mis-off": dict(max_bounces=3, mis_mode=0),
    "bounce-hemisphere": dict(max_bounces=3, bounce_mode=0),
    "bounce-cosine": dict(max_bounces=3, bounce_mode=1),
    "tri-area": dict(max_bounces=3, tri_light_mode=0),
    "tri-hybrid": dict(max_bounces=3, tri_light_mode=2),
    "regularization+clamp": dict(max_bounces=5, regularization_gamma=0.2, indirect_clamping=4.0),
    "russian-roulette": dict(max_bounces=6, russian_roulette_delta=2.0),
    "hide-lights-seed": dict(max_bounces=3, hide_lights=1, rng_seed=1234),
    "no-nee": dict(max_bounces=3, nee_point=0.0, nee_directional=0.0, nee_triangles=0.0),
    "nee-weights": dict(max_bounces=3, nee_point=3.0, nee_directional=0.5, nee_triangles=2.0),
    "white-albedo-transparent": dict(max_bounces=3, use_white_albedo_on_first_bounce=1, transparent_background=1),
    "4spp-2-per-pass": dict(max_bounces=3, samples_per_pixel=4, samples_per_pass=2),
    "dof": dict(max_bounces=2, depth_of_field=1),
}

def scene_classes():
    """NEE weights options_for_scene zeroes, per scene family (a class the scene lacks is switched off, and the classes in use are part
    of a specialised program): read off the scenes themselves - test.glb and a small instance of the procedural bench scenes."""
    from . import scenes
    from .renderer import options_for_scene
    out = {}
    for name, sc in (("test_glb", scenes.test_glb(64, 64)), ("bench", scenes.sponza_class(seed=1, target_tris=2000, width=64, height=64))):
        o = options_for_scene(sc)
        out[name] = {k: 0.0 for k in ("nee_point", "nee_directional", "nee_envmap", "nee_triangles") if getattr(o, k) == 0.0}
    return out


def precompile(kw, shade_tris=True, ieee=False, count_work=False, arch=None):
    """One option set into the kernel cache (ray generation and shading programs); returns the seconds it took."""
    import time
    o = make_options(**kw)
    t = time.perf_counter()
    _lib.check(_lib.lib().trhip_pt_precompile(C.byref(o), int(shade_tris), int(ieee), int(count_work), arch.encode() if arch else None))
    return time.perf_counter() - t


def _job(job):
    kw, ieee, count = job
    return precompile(kw, True, ieee, count)


def warm_up_jobs():
    jobs = []
    SCENE_CLASSES = scene_classes()
    for name, kw in NAMED_OPTION_SETS.items():
        base = dict(kw, **SCENE_CLASSES["test_glb"])      # a weight the set names itself survives only where the scene has the class
        jobs += [(base, False, False), (base, True, False)]
    for sampler in (1, 2, 3):
        jobs.append((dict(SCENE_CLASSES["bench"], max_bounces=4, sampler=sampler), False, False))
    for name, kw in REFERENCE_PRESETS.items():
        for scene in ("bench", "test_glb"):
            jobs.append((dict(SCENE_CLASSES[scene], **dict(kw, samples_per_pixel=1)), False, False))
    # what the bench's counting and timing passes launch for those sets
    for sampler in (1, 3):
        jobs.append((dict(SCENE_CLASSES["bench"], max_bounces=4, sampler=sampler), False, True))
    for name in ("quality", "reference"):
        jobs.append((dict(SCENE_CLASSES["bench"], **dict(REFERENCE_PRESETS[name], samples_per_pixel=1)), False, True))
    return jobs


def warm_kernel_cache(workers=None, verbose=False, clean=False):
    """Compiles the shading programs of every named option set into the kernel cache (trhip_kernel_cache_dir) in parallel processes.
    clean: programs of earlier builds (other sources, other hashes) are removed first."""
    import concurrent.futures as F
    import glob
    import time
    jobs = warm_up_jobs()
    if clean:
        for f in glob.glob(os.path.join(_lib.lib().trhip_kernel_cache_dir().decode(), "spec_*.hsaco")):
            os.remove(f)
    t0 = time.perf_counter()
    workers = workers or min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 8)
    with F.ProcessPoolExecutor(max_workers=workers) as ex:
        times = list(ex.map(_job, jobs))
    if verbose:
        print(f"kernel cache {_lib.lib().trhip_kernel_cache_dir().decode()}: {len(jobs)} option sets, {sum(t > 0.05 for t in times)} compiled, "
              f"{time.perf_counter() - t0:.1f} s on {workers} processes")
    return len(jobs)
