"""Shading programs compiled for one option set (csrc/shade_spec.hip through csrc/specialize.cc): what the reference does with its
options as #defines when it builds a stage's pipeline (src/path_tracer_stage.cc:30-116).  CPU part: the compiler path needs no GPU
(trhip_pt_precompile), the kernel cache is keyed by option set, arithmetic and sources.  GPU part: a specialised program renders the
bits of the general kernels at IEEE fp32 and agrees with them to a few ulps per operation at the default arithmetic - whole frames,
every sampler / film / MIS / bounce / light mode, counters too."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

_PRECOMPILE = r"""
import ctypes as C, sys, time, json
sys.path.insert(0, sys.argv[1])
from tauray_amd import _lib
from tauray_amd.renderer import make_options
L = _lib.lib()
out = {"dir": L.trhip_kernel_cache_dir().decode(), "times": []}
for kw in json.loads(sys.argv[2]):
    ieee = kw.pop("_ieee", 0)
    o = make_options(**kw)
    t = time.perf_counter()
    rc = L.trhip_pt_precompile(C.byref(o), 1, ieee, 0, None)
    out["times"].append([rc, time.perf_counter() - t])
print(json.dumps(out))
"""


def _run_precompile(cache_dir, sets):
    import json
    env = dict(os.environ, TRHIP_KERNEL_CACHE=str(cache_dir))
    r = subprocess.run([sys.executable, "-c", _PRECOMPILE, ROOT, json.dumps(sets)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_precompile_needs_no_gpu_and_fills_the_cache(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    sets = [dict(sampler=1), dict(sampler=1, _ieee=1), dict(sampler=1, film=2), dict()]
    first = _run_precompile(cache, sets)
    assert first["dir"] == str(cache) and all(rc == 0 for rc, _ in first["times"])
    files = sorted(os.listdir(cache))
    # two programs (ray generation, shading) per new option set; the ray-generation program does not depend on the arithmetic, and the
    # command-line set (the last one) is served by the ahead-of-time instances of libtrhip.so: nothing to compile
    assert len(files) == 2 + 1 + 2 and all(f.startswith("spec_") and f.endswith(".hsaco") for f in files)
    assert all(open(cache / f, "rb").read(4) == b"\x7fELF" for f in files)
    assert first["times"][3][1] < 0.05
    second = _run_precompile(cache, sets)      # another process: everything comes from the cache
    assert sorted(os.listdir(cache)) == files and all(t < 0.2 for _, t in second["times"])


def test_concurrent_first_compilations_do_not_share_files(tmp_path):
    """Eight ranks whose first frame compiles the same program at the same moment (one process per GPU, an empty cache).  hipRTC
    materialises every embedded header under one temporary directory per compilation - as long as no header name leaves it: a name
    with "../" in it landed in /tmp/include/, shared by all processes, and compilations failed on each other's half-written files."""
    import re
    inc = open(os.path.join(ROOT, "tauray_amd", "csrc", "rtc_sources.inc")).read()
    names = re.findall(r'^    \{"([^"]+)",$', inc, flags=re.M)
    assert len(names) >= 8 and all("/" not in n and ".." not in n for n in names)
    assert all("/" not in m for m in re.findall(r'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', inc, flags=re.M))
    cache = tmp_path / "cache"
    cache.mkdir()
    import json
    env = dict(os.environ, TRHIP_KERNEL_CACHE=str(cache), AMD_COMGR_CACHE_DIR=str(tmp_path / "comgr"))
    sets = json.dumps([dict(sampler=2, film=1, mis_mode=0, max_bounces=2)])
    procs = [subprocess.Popen([sys.executable, "-c", _PRECOMPILE, ROOT, sets], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(8)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [e[-1500:] for _, e in outs if e][:1]
    assert all(json.loads(o.strip().splitlines()[-1])["times"][0][0] == 0 for o, _ in outs)
    files = sorted(os.listdir(cache))
    assert len(files) == 2 and not [f for f in files if ".tmp" in f]
    assert all(open(cache / f, "rb").read(4) == b"\x7fELF" and open(cache / f, "rb").read()[-24:-16] == b"TRHSACO1" for f in files)


def test_build_warmed_the_cache_for_the_named_option_sets():
    """__graft_entry__.build() compiles the programs of the reference's presets and of the sets the tests and the bench render."""
    from tauray_amd import _lib, presets
    d = _lib.lib().trhip_kernel_cache_dir().decode()
    assert d and os.path.isdir(d), "run __graft_entry__.build()"
    t = time.perf_counter()
    for kw, ieee, count in presets.warm_up_jobs()[:6]:
        assert presets.precompile(kw, True, ieee, count) < 0.5, f"{kw} was not in {d}: run __graft_entry__.build()"
    # six look-ups, not six compilations (4-8 s each); the bound leaves room for a loaded machine and the first load of libhiprtc
    assert time.perf_counter() - t < 12.0


def test_reference_presets_are_what_the_cfg_files_say():
    """tauray_amd/presets.py against data/presets/*.cfg of the reference (restated here: the reference is not on the GPU box)."""
    from tauray_amd.presets import REFERENCE_PRESETS as P
    # quality.cfg: film blackman-harris, max-ray-depth 4, samples-per-pixel 4096, sampler uniform-random, regularization 0.1
    assert P["quality"] == dict(film=2, max_bounces=4, samples_per_pixel=4096, sampler=0, regularization_gamma=0.1)
    # reference.cfg: film blackman-harris, max-ray-depth 8, samples-per-pixel 16384, sampler uniform-random, tri-light-mode hybrid
    assert P["reference"] == dict(film=2, max_bounces=8, samples_per_pixel=16384, sampler=0, tri_light_mode=2)
    # accumulation.cfg: film blackman-harris, max-ray-depth 5, sampler uniform-random, samples-per-pixel 1, regularization 0.2
    assert P["accumulation"] == dict(film=2, max_bounces=5, samples_per_pixel=1, sampler=0, regularization_gamma=0.2)


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def R():
    from tauray_amd import renderer
    return renderer


def _frame(R, ctx, ss, scene, size, specialize, ieee, count=False, **kw):
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, **kw), DistributionParams(tuple(size), DISTRIBUTION_DUPLICATE, 0, 1, True))
    pt.set_specialization(specialize)
    pt.set_shading_arithmetic(ieee)
    if count:
        pt.set_profiling(True, False)
    buf = ctx.alloc(size[0] * size[1] * 16).zero()
    pt.run(buf)
    img = buf.download((size[1], size[0], 4))
    c = pt.counters()
    pt.close()
    return img, c


@pytest.mark.gpu
def test_specialised_programs_render_the_bits_of_the_general_kernels(R):
    """Every axis a program pins, one at a time and together, in both arithmetics, on the bench scene's little sister (textures, a sun,
    an environment map, emissive triangles) at a size that takes the four-lane schedule; counting instances report the same work."""
    from tauray_amd import scenes
    from tauray_amd.presets import REFERENCE_PRESETS
    W, H = 960, 540
    scene = scenes.sponza_class(seed=3, target_tris=40000, width=W, height=H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    sets = [dict(max_bounces=4, sampler=1), dict(max_bounces=4, sampler=2), dict(max_bounces=4, sampler=3, film=1),
            dict(REFERENCE_PRESETS["quality"], samples_per_pixel=1), dict(REFERENCE_PRESETS["reference"], samples_per_pixel=1),
            dict(max_bounces=3, mis_mode=0, bounce_mode=0, tri_light_mode=0, nee_envmap=0.0),
            dict(max_bounces=5, mis_mode=1, bounce_mode=1, russian_roulette_delta=1.5, indirect_clamping=5.0, hide_lights=1, use_white_albedo_on_first_bounce=1,
                 transparent_background=1, depth_of_field=1, pre_transformed_vertices=1, samples_per_pixel=2)]
    for kw in sets:
        for ieee in (True, False):
            spec, cs = _frame(R, ctx, ss, scene, (W, H), True, ieee, **kw)
            gen, cg = _frame(R, ctx, ss, scene, (W, H), False, ieee, **kw)
            assert np.isfinite(gen[..., :3]).mean() > 0.999 and np.nanmean(gen[..., :3]) > 1e-3
            if ieee:
                assert np.array_equal(spec, gen, equal_nan=True), f"{kw}, IEEE fp32: {int((spec != gen).any(-1).sum())} pixels differ"
                assert cs["closest_rays"] == cg["closest_rays"] and cs["shadow_rays"] == cg["shadow_rays"]
            else:
                # at the accuracy Vulkan asks for, two instances of the kernel are two implementations: a few ulps per operation apart
                # (the compiler picks v_rsq_f32 or v_rcp_f32(v_sqrt_f32) by the code around a 1 / sqrt(x)), a flipped decision on a rare path
                ok = np.isfinite(spec).all(-1) & np.isfinite(gen).all(-1)
                close = (np.abs(spec - gen) <= 1e-4 * np.abs(gen) + 1e-6).all(-1)
                assert close[ok].mean() > 0.97, f"{kw}: {1 - close[ok].mean():.3%} of the pixels differ by more than 1e-4"
                assert abs(float(spec[ok][:, :3].mean()) / float(gen[ok][:, :3].mean()) - 1) < 2e-3
                assert abs(cs["closest_rays"] / cg["closest_rays"] - 1) < 1e-3 and abs(cs["shadow_rays"] / cg["shadow_rays"] - 1) < 1e-3
    kw = sets[0]
    spec, cs = _frame(R, ctx, ss, scene, (W, H), True, True, count=True, **kw)
    gen, cg = _frame(R, ctx, ss, scene, (W, H), False, True, count=True, **kw)
    # (node visits, triangle and alpha tests are not compared: the quad tail of a trace wave starts when at most sixteen of its rays are
    # left, i.e. they depend on which rays share a wave, which is the order the queue appends happened to land in)
    assert np.array_equal(spec, gen) and cs["surface_hits"] > 0
    assert all(cs[k] == cg[k] for k in ("closest_rays", "shadow_rays", "surface_hits"))


@pytest.mark.gpu
def test_a_program_warmed_ahead_of_time_is_found_at_run_time(tmp_path):
    """trhip_pt_precompile (what __graft_entry__.build() runs, for the bare processor name) and the stage's first render have to name the
    same cache file.  They did not: the run-time side hashed hipDeviceProp_t::gcnArchName as it comes ("gfx950:sramecc+:xnack-"), so every
    warmed program missed and was compiled again (round 4's advisor finding)."""
    import json
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, TRHIP_KERNEL_CACHE=str(cache), TRHIP_DEBUG="1")
    from tauray_amd import presets
    kw = dict(max_bounces=3, sampler=3, film=1, mis_mode=2, tri_light_mode=1)
    # the light classes in use are part of a program: ahead of time they are named the way options_for_scene will set them
    warm = dict(kw, **presets.scene_classes()["test_glb"])
    r = subprocess.run([sys.executable, "-c", _PRECOMPILE, ROOT, json.dumps([warm])], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    warmed = sorted(os.listdir(cache))
    assert len(warmed) == 2, warmed
    script = tmp_path / "render.py"
    script.write_text(r"""
import sys, json
sys.path.insert(0, sys.argv[1])
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
sc = scenes.test_glb(64, 64)
ctx = R.Context(0)
ss = R.SceneStage(ctx, sc)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(sc, **json.loads(sys.argv[2])), DistributionParams((64, 64), DISTRIBUTION_DUPLICATE, 0, 1, True))
buf = ctx.alloc(64 * 64 * 16).zero()
pt.run(buf)
buf.download((64, 64, 4))
""")
    r = subprocess.run([sys.executable, str(script), ROOT, json.dumps(kw)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("from the kernel cache") == 2 and "compiled through hipRTC" not in r.stderr, r.stderr[-1500:]
    assert sorted(os.listdir(cache)) == warmed


@pytest.mark.gpu
def test_a_new_option_set_is_compiled_when_it_first_renders(R, tmp_path):
    """A stage whose option set is in no cache: the first frame compiles (hipRTC), the next process finds the program in the cache."""
    script = tmp_path / "first_frame.py"
    script.write_text(r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
sc = scenes.test_glb(96, 96)
ctx = R.Context(0)
ss = R.SceneStage(ctx, sc)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(sc, max_bounces=3, sampler=2, film=1, mis_mode=1, tri_light_mode=2), DistributionParams((96, 96), DISTRIBUTION_DUPLICATE, 0, 1, True))
buf = ctx.alloc(96 * 96 * 16).zero()
t = time.perf_counter()
pt.run(buf)
img = buf.download((96, 96, 4))
print("FIRST", time.perf_counter() - t)
np.save(sys.argv[2], img)
""")
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, TRHIP_KERNEL_CACHE=str(cache), TRHIP_DEBUG="1")
    runs = []
    for k in range(2):
        out = str(tmp_path / f"f{k}.npy")
        r = subprocess.run([sys.executable, str(script), ROOT, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append((float(r.stdout.split("FIRST")[1].split()[0]), r.stderr, np.load(out)))
    assert "compiled through hipRTC" in runs[0][1] and "from the kernel cache" in runs[1][1] and "compiled through hipRTC" not in runs[1][1]
    assert len(os.listdir(cache)) == 2 and np.array_equal(runs[0][2], runs[1][2]) and np.isfinite(runs[0][2]).all()
    assert runs[1][0] < runs[0][0]


@pytest.mark.gpu
def test_a_cache_file_that_does_not_load_is_compiled_again(R, tmp_path):
    """A truncated or foreign file under the cache's name (a full disk, another writer) fails the trailer check of csrc/specialize.cc: it is
    not handed to the loader, the program is compiled again and the file replaced - the stage does not fall to the general kernels."""
    script = tmp_path / "frame.py"
    script.write_text(r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from tauray_amd import renderer as R, scenes
from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
sc = scenes.test_glb(64, 64)
ctx = R.Context(0)
ss = R.SceneStage(ctx, sc)
pt = R.PathTracerStage(ctx, ss, R.options_for_scene(sc, max_bounces=2, sampler=3, mis_mode=1), DistributionParams((64, 64), DISTRIBUTION_DUPLICATE, 0, 1, True))
pt.set_shading_arithmetic(True)
buf = ctx.alloc(64 * 64 * 16).zero()
pt.run(buf)
np.save(sys.argv[2], buf.download((64, 64, 4)))
""")
    cache = tmp_path / "cache"
    cache.mkdir()
    env = dict(os.environ, TRHIP_KERNEL_CACHE=str(cache), TRHIP_DEBUG="1")

    def run(k):
        out = str(tmp_path / f"f{k}.npy")
        r = subprocess.run([sys.executable, str(script), ROOT, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stderr, np.load(out)

    first = run(0)
    files = sorted(os.listdir(cache))
    assert len(files) == 2 and "compiled through hipRTC" in first[0]
    sizes = {f: os.path.getsize(cache / f) for f in files}
    (cache / files[0]).write_bytes(b"not a code object")
    (cache / files[1]).write_bytes((cache / files[1]).read_bytes()[: sizes[files[1]] // 2])
    again = run(1)
    assert again[0].count("compiled through hipRTC") == 2 and "from the kernel cache" not in again[0] and "general kernels" not in again[0], again[0][-1500:]
    assert np.array_equal(first[1], again[1])
    assert {f: os.path.getsize(cache / f) for f in sorted(os.listdir(cache))} == sizes
    third = run(2)
    assert third[0].count("from the kernel cache") == 2 and "compiled through hipRTC" not in third[0] and np.array_equal(first[1], third[1])
