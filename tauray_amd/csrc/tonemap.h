// tonemap_stage per pixel: shader/tonemap.glsl:35-55 + tonemap_{gamma,filmic,reinhard,reinhard_luminance}.comp.  One function for the
// stage's own kernel (k_tonemap, api.hip) and for k_resolve when a renderer without a stage between the two asks for the display image
// straight from the resolve (trhip_pt_set_fused_tonemap): the same expressions in the same translation-unit flags, so the same bits.
#pragma once
#include "shading.h"

namespace tr {
namespace {

TR_DEV f4 tonemap_pixel(f4 col, int op, float exposure, float gamma, int grid, uint x, uint y) {
    f3 c;
    if (op <= 1) c = F3(col) * exposure;
    else if (op == 2) {
        c = min3(max3(F3(col) * exposure, F3(0)), F3(1000));
        c = max3(F3(0.0f), c - 0.004f);
        f3 q = (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
        c = F3(powf(q.x, 2.2f), powf(q.y, 2.2f), powf(q.z, 2.2f));
    } else if (op == 3) {
        c = min3(max3(F3(col) * exposure, F3(0)), F3(1000));
        c = c / (F3(1.0f) + c);
    } else {
        c = min3(max3(F3(col) * exposure, F3(0)), F3(1000));
        float lum = rgb_to_luminance(c);
        float new_lum = lum / (1.0f + lum);
        c = c / fmax2(lum, 1e-4f) * new_lum;
    }
    if (gamma != 1.0f) { float ig = 1.0f / gamma; c = F3(powf(c.x, ig), powf(c.y, ig), powf(c.z, ig)); }
    if (grid != 0) {
        int gx = (int)(x / (uint)grid) & 1, gy = (int)(y / (uint)grid) & 1;
        f3 ac = (gx ^ gy) == 0 ? F3(0.4f) : F3(0.6f);
        c = mix3(ac, c, col.w);
    }
    return F4(c, col.w);
}

}  // namespace
}  // namespace tr
