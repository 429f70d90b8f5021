// Texture sampling for ray-tracing stages: implicit-LOD texture() has no derivatives there, so it is
// base level, bilinear, repeat addressing (src/sampler_table.cc:8-17), on un-normalised RGBA8 texels or - a 16-bit PNG, which the
// reference keeps as R16G16B16A16Unorm (src/gltf.cc:548-556) - RGBA16 texels (TextureInfo::format).
#pragma once
#include "common.h"

namespace tr {

TR_DEV int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

TR_DEV f4 fetch_rgba8(const uint8_t* base, int w, int x, int y) {
    uint p = *reinterpret_cast<const uint*>(base + ((size_t)y * (size_t)w + (size_t)x) * 4);   // one dword per texel
    return F4((float)(p & 0xFF), (float)((p >> 8) & 0xFF), (float)((p >> 16) & 0xFF), (float)(p >> 24)) * (1.0f / 255.0f);
}

TR_DEV f4 fetch_rgba16(const uint8_t* base, int w, int x, int y) {
    const uint2 p = *reinterpret_cast<const uint2*>(base + ((size_t)y * (size_t)w + (size_t)x) * 8);   // two dwords per texel
    return F4((float)(p.x & 0xFFFF), (float)(p.x >> 16), (float)(p.y & 0xFFFF), (float)(p.y >> 16)) * (1.0f / 65535.0f);
}

TR_DEV f4 bilerp(f4 c00, f4 c10, f4 c01, f4 c11, float fx, float fy) {
    f4 top = c00 * (1.0f - fx) + c10 * fx;
    f4 bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

TR_DEV f4 sample_texture(const SceneView& sv, int tex_id, f2 uv) {
    const TextureInfo ti = sv.tex_infos[tex_id];
    int w = (int)ti.width, h = (int)ti.height;
    // a texture unit returns a texel for any coordinate; non-finite ones (the uv of a light sample on a triangle seen
    // edge-on, whose weight is zero anyway) are defined as 0 here so that 0 * texel stays 0
    if (!(fabsf(uv.x) < __builtin_huge_valf())) uv.x = 0.0f;
    if (!(fabsf(uv.y) < __builtin_huge_valf())) uv.y = 0.0f;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat((int)fx0, w), y0 = wrap_repeat((int)fy0, h);
    int x1 = wrap_repeat((int)fx0 + 1, w), y1 = wrap_repeat((int)fy0 + 1, h);
    const uint8_t* base = sv.texels + (size_t)ti.texel_offset * 4;
    if (ti.format == TEXTURE_FORMAT_RGBA16)
        return bilerp(fetch_rgba16(base, w, x0, y0), fetch_rgba16(base, w, x1, y0), fetch_rgba16(base, w, x0, y1), fetch_rgba16(base, w, x1, y1), fx, fy);
    return bilerp(fetch_rgba8(base, w, x0, y0), fetch_rgba8(base, w, x1, y0), fetch_rgba8(base, w, x0, y1),
                  fetch_rgba8(base, w, x1, y1), fx, fy);
}

TR_DEV f4 sample_envmap(const SceneView& sv, f2 uv) {
    int w = (int)sv.env_w, h = (int)sv.env_h;
    float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = wrap_repeat((int)fx0, w), y0 = wrap_repeat((int)fy0, h);
    int x1 = wrap_repeat((int)fx0 + 1, w), y1 = wrap_repeat((int)fy0 + 1, h);
    const f4* e = sv.envmap;
    return bilerp(e[(size_t)y0 * w + x0], e[(size_t)y0 * w + x1], e[(size_t)y1 * w + x0], e[(size_t)y1 * w + x1], fx, fy);
}

}  // namespace tr
