#!/usr/bin/env python3
"""Where does the HIP path tracer differ from the one Vulkan image of the reference (test/references/validate_path-tracer.exr,
decoded to tests/golden/validate_path-tracer.npz)?  Renders test/test.glb at 512x512 with the options the image was made with
(CLI defaults: 8 bounces, uniform-random sampler, point film, filmic + gamma 2.2) at many samples per pixel and breaks the
residual down by what the primary ray hits (instance ids from the feature renderer): room faces, teapot, Suzanne (glass), the
emissive torus, the alpha-blended plane.  Per region: pixels, mean of both images, signed mean offset, RMS of 16x16 block means
(noise of both images averaged out), per-pixel RMS; and the same for the linear radiance split into the demodulated diffuse /
reflection targets to see which lobe an offset sits in.  usage (GPU box): python tools/golden_residual.py [spp] [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "room face 0", 1: "room face 1", 2: "room face 2", 3: "room face 3", 4: "teapot", 5: "suzanne (glass)", 6: "torus (emitter)", 7: "plane (alpha)", -1: "miss"}


def main():
    from tauray_amd import renderer as R
    from tauray_amd.gltf import load_glb
    from tauray_amd.distribution import DistributionParams, DISTRIBUTION_DUPLICATE
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    W = H = 512
    scene = load_glb(os.path.join(ROOT, "tests", "golden", "test.glb"), W, H)
    ctx = R.Context(0)
    ss = R.SceneStage(ctx, scene)
    d = DistributionParams((W, H), DISTRIBUTION_DUPLICATE, 0, 1, True)
    fs = R.FeatureStage(ctx, ss, 9, d)
    buf = ctx.alloc(W * H * 16).zero()
    fs.run(buf)
    ids = buf.download((H, W, 4))[..., 0]
    ids = np.where(np.isnan(ids), -1, ids).astype(np.int32)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "validate_path-tracer.npz"))["rgb"].astype(np.float64)
    out = {"spp": spp, "regions": {}}
    images = {}
    for tag, ieee, half in (("default", False, 0), ("first_half", False, 1), ("second_half", False, 2)):
        n = spp if half == 0 else spp // 2
        pt = R.PathTracerStage(ctx, ss, R.options_for_scene(scene, samples_per_pixel=n, samples_per_pass=1, rng_seed=0 if half < 2 else 7), d)
        color, disp = ctx.alloc(W * H * 16).zero(), ctx.alloc(W * H * 16)
        pt.run(color)
        R.TonemapStage(ctx).run(color, disp, W, H)
        images[tag] = (color.download((H, W, 4))[..., :3].astype(np.float64), disp.download((H, W, 4))[..., :3].astype(np.float64))
        pt.close()
    lin, ours = images["default"]
    own_noise = images["first_half"][1] - images["second_half"][1]      # two independent halves: per-pixel noise of an spp/2 image
    b = 16

    def block_means(img, mask):
        s = np.where(mask[..., None], img, 0).reshape(H // b, b, W // b, b, 3).sum((1, 3))
        n = mask.reshape(H // b, b, W // b, b).sum((1, 3))
        return s, n

    for k in sorted(set(ids.reshape(-1).tolist())) + ["all", "all but the torus"]:
        mask = (ids == k) if isinstance(k, int) else (np.ones_like(ids, bool) if k == "all" else ids != 6)
        if mask.sum() < 64:
            continue
        o, g = ours[mask], gold[mask]
        so, n = block_means(ours, mask)
        sg, _ = block_means(gold, mask)
        valid = n >= 64
        blk = ((so - sg) / np.maximum(n, 1)[..., None])[valid]
        sn, _ = block_means(own_noise, mask)
        blk_own = (sn / np.maximum(n, 1)[..., None])[valid] / 2.0      # block means of (a - b) / 2 = noise of the full image's block means
        out["regions"][NAMES.get(k, k) if isinstance(k, int) else k] = {
            "pixels": int(mask.sum()), "mean_ours": float(o.mean()), "mean_gold": float(g.mean()), "rel_mean_offset": float((o.mean() - g.mean()) / g.mean()),
            "rel_mean_offset_rgb": [float((o[:, c].mean() - g[:, c].mean()) / max(g[:, c].mean(), 1e-9)) for c in range(3)],
            "pixel_rms": float(np.sqrt(((o - g) ** 2).mean())), "pixel_rms_own_noise": float(np.sqrt((own_noise[mask] ** 2).mean()) / 2.0),
            "block16_rms": float(np.sqrt((blk ** 2).mean())) if blk.size else None, "block16_rms_own_noise": float(np.sqrt((blk_own ** 2).mean())) if blk_own.size else None,
            "block16_mean_signed": float(blk.mean()) if blk.size else None, "blocks": int(valid.sum())}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        np.savez_compressed(sys.argv[2].replace(".json", "_images.npz"), ours=ours.astype(np.float32), ids=ids)


if __name__ == "__main__":
    main()
