// tauray_envmap.hh - `--envmap=file.hdr|file.exr` for the C++ host layer (src/options.hh:125): a Radiance .hdr or OpenEXR file becomes the lat-long
// environment map of a tr::scene_data, with the alias table environment_map::generate_alias_table builds for importance sampling
// (src/environment_map.cc:39-140, importance = shader/alias_table_importance.comp:16-28 evaluated on the host).
// Same results as the Python mirror (tauray_amd/hdr.py, tauray_amd/scene.py build_alias_table), operation by operation:
// tests/test_cpp_host.py::test_cpp_envmap_matches_python.
#ifndef TAURAY_ENVMAP_HH
#define TAURAY_ENVMAP_HH
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "tauray_hip.hh"
#include "tauray_exr.hh"

namespace tr
{

// texture::load_from_file -> stbi_loadf (src/texture.cc:453-461) + the alpha channel it appends: RGBA32F, row 0 = the top row of
// the file; rgb = mantissa * 2^(exponent - 136), 0 for exponent 0.  Flat and run-length-encoded scanlines, -Y +X only (stb_image).
inline std::vector<float> load_hdr(const std::string& path, uint32_t& width, uint32_t& height)
{
    std::ifstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("Failed to open " + path);
    std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto line = [&]() {
        std::string l;
        while(pos < raw.size() && raw[pos] != '\n') l.push_back((char)raw[pos++]);
        if(pos >= raw.size()) throw std::runtime_error(path + ": truncated .hdr header");
        ++pos;
        return l;
    };
    const std::string magic = line();
    if(magic.compare(0, 10, "#?RADIANCE") != 0 && magic.compare(0, 6, "#?RGBE") != 0) throw std::runtime_error(path + ": not a Radiance .hdr file");
    bool format_ok = false;
    for(std::string l = line(); !l.empty(); l = line()) if(l == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    if(!format_ok) throw std::runtime_error(path + ": unsupported .hdr format");
    {
        std::istringstream res(line());
        std::string ys, xs; long h = 0, w = 0;
        res >> ys >> h >> xs >> w;
        if(ys != "-Y" || xs != "+X" || h <= 0 || w <= 0) throw std::runtime_error(path + ": unsupported .hdr data layout");
        width = (uint32_t)w; height = (uint32_t)h;
    }
    const size_t w = width, h = height;
    // a resolution line that promises more than the data can hold: a run code repeats a byte 127 times at best
    if(w > (1u << 24) || h > (1u << 24) || w * h * 4 > (raw.size() - pos) * 64 + 1024) throw std::runtime_error(path + ": the .hdr resolution does not fit the file");
    std::vector<uint8_t> rgbe(w * h * 4);
    bool flat = w < 8 || w >= 32768;
    auto need = [&](size_t n) { if(pos + n > raw.size()) throw std::runtime_error(path + ": truncated .hdr data"); };
    for(size_t y = 0; y < h; ++y)
    {
        if(!flat)
        {
            need(4);
            if(!(raw[pos] == 2 && raw[pos + 1] == 2 && !(raw[pos + 2] & 0x80)))
            {
                if(y != 0) throw std::runtime_error(path + ": corrupt .hdr scanline");
                flat = true;      // the first scanline decides, like stb_image
            }
        }
        if(flat) { need(4 * w); std::memcpy(&rgbe[y * w * 4], &raw[pos], 4 * w); pos += 4 * w; continue; }
        if(((size_t)raw[pos + 2] << 8 | raw[pos + 3]) != w) throw std::runtime_error(path + ": invalid decoded scanline length");
        pos += 4;
        for(int c = 0; c < 4; ++c)
            for(size_t x = 0; x < w;)
            {
                need(2);
                size_t count = raw[pos++];
                if(count > 128)
                {
                    count -= 128;
                    if(x + count > w) throw std::runtime_error(path + ": corrupt .hdr run");
                    const uint8_t v = raw[pos++];
                    for(size_t k = 0; k < count; ++k) rgbe[(y * w + x + k) * 4 + (size_t)c] = v;
                }
                else
                {
                    need(count);
                    if(count == 0 || x + count > w) throw std::runtime_error(path + ": corrupt .hdr run");
                    for(size_t k = 0; k < count; ++k) rgbe[(y * w + x + k) * 4 + (size_t)c] = raw[pos++];
                }
                x += count;
            }
    }
    std::vector<float> out(w * h * 4);
    for(size_t i = 0; i < w * h; ++i)
    {
        const int e = rgbe[i * 4 + 3];
        const float scale = e != 0 ? std::ldexp(1.0f, e - 136) : 0.0f;
        // "16-bit floats for hdr images" (src/texture.cc:498-500 -> :50-66): clamped to +-65000, rounded to half.  Exact for the
        // 8-bit mantissas of RGBE except beyond the clamp (a sun disc) and below 2^-24.
        for(int c = 0; c < 3; ++c)
            out[i * 4 + (size_t)c] = exr::half_to_float(exr::float_to_half(std::min(std::max((float)rgbe[i * 4 + (size_t)c] * scale, -65000.0f), 65000.0f)));
        out[i * 4 + 3] = 1.0f;
    }
    return out;
}

// environment_map::generate_alias_table (src/environment_map.cc:39-140): 16-byte entries {alias_id, probability (32-bit fixed
// point), pdf, alias_pdf} (shader/alias_table.glsl:7-13).  Float arithmetic in float, the trigonometry evaluated in double and
// rounded, the average from one running double sum - exactly as tauray_amd/scene.py build_alias_table.
inline std::vector<uint8_t> build_alias_table(const float* rgba, uint32_t w, uint32_t h)
{
    struct entry { uint32_t alias_id, probability; float pdf, alias_pdf; };
    const size_t n = (size_t)w * h;
    const float pi_f = (float)3.14159265358979323846, two_pi_f = (float)(2.0 * 3.14159265358979323846);
    std::vector<float> importance(n), solid(h), denom(h);
    for(uint32_t y = 0; y < h; ++y)
    {
        const float y0 = (float)y / (float)h, y1 = ((float)y + 1.0f) / (float)h;
        const float c0 = (float)std::cos((double)(pi_f * y0)), c1 = (float)std::cos((double)(pi_f * y1));
        solid[y] = two_pi_f * (c0 - c1) / (float)w;
        const float sin_theta = (float)std::sin((double)(((float)y + 0.5f) / (float)h * pi_f));
        denom[y] = (float)(2.0 * 3.14159265358979323846 * 3.14159265358979323846) * sin_theta;
    }
    double total = 0;
    for(size_t i = 0; i < n; ++i)
    {
        const float lum = rgba[i * 4] * 0.2126f + rgba[i * 4 + 1] * 0.7152f + rgba[i * 4 + 2] * 0.0722f;
        importance[i] = lum * solid[i / w];
        total += (double)importance[i];
    }
    const float inv_average = total > 0 ? (float)(1.0 / (total / (double)n)) : 0.0f;
    for(float& v: importance) v = v * inv_average;
    std::vector<entry> table(n);
    for(size_t i = 0; i < n; ++i) table[i] = entry{(uint32_t)i, 0xFFFFFFFFu, 0.0f, 0.0f};
    auto fixed32 = [](float v) {
        const double r = std::ldexp((double)v, 32);
        return (uint32_t)(r < 0.0 ? 0.0 : (r > 4294967295.0 ? 4294967295.0 : r));
    };
    size_t i = 0, j = 0;
    while(i < n && importance[i] > 1.0f) ++i;
    while(j < n && importance[j] <= 1.0f) ++j;
    float weight = j < n ? importance[j] : 0.0f;
    while(j < n)
    {
        if(weight > 1.0f)
        {
            if(i >= n) break;
            table[i].probability = fixed32(importance[i]);
            table[i].alias_id = (uint32_t)j;
            weight = (weight + importance[i]) - 1.0f;
            ++i;
            while(i < n && importance[i] > 1.0f) ++i;
        }
        else
        {
            table[j].probability = fixed32(weight);
            const size_t old_j = j;
            ++j;
            while(j < n && importance[j] <= 1.0f) ++j;
            if(j < n)
            {
                table[old_j].alias_id = (uint32_t)j;
                weight = (weight + importance[j]) - 1.0f;
            }
        }
    }
    for(size_t k = 0; k < n; ++k)
    {
        table[k].pdf = importance[k] / denom[k / w];
        const size_t a = table[k].alias_id;
        table[k].alias_pdf = importance[a] / denom[a / w];
    }
    const uint8_t* p = reinterpret_cast<const uint8_t*>(table.data());
    return std::vector<uint8_t>(p, p + n * sizeof(entry));
}

// environment_map(dev, path) with the default factor (1, 1, 1) and the lat-long projection (src/tauray.cc:198-201)
inline void set_envmap(scene_data& s, const std::string& path)
{
    uint32_t w = 0, h = 0;
    std::vector<float> px;
    if(path.size() >= 4 && path.compare(path.size() - 4, 4, ".exr") == 0)
    {
        // texture::load_from_file, the `.exr` branch (src/texture.cc:409-429): floats as they are, alpha 1 appended to three channels
        int n = 0;
        const std::vector<float> f = exr::load_exr(path, w, h, n);
        if(n != 3 && n != 4) throw std::runtime_error(path + ": an environment map needs three or four channels");
        px.resize(size_t(w) * h * 4);
        for(size_t i = 0; i < size_t(w) * h; ++i)
            for(int c = 0; c < 4; ++c) px[4 * i + c] = c < n ? f[size_t(n) * i + c] : 1.0f;
    }
    else px = load_hdr(path, w, h);
    // an infinite or NaN texel (an .exr written with half overflow, say) makes every entry of the alias table NaN and every sample of
    // the frame with it; the reference renders that, this loader says so
    for(float v: px) if(!std::isfinite(v)) throw std::runtime_error(path + ": the environment map holds non-finite texels");
    const uint8_t* p = reinterpret_cast<const uint8_t*>(px.data());
    s.envmap.assign(p, p + px.size() * 4);
    s.envmap_width = w; s.envmap_height = h;
    s.alias_table = build_alias_table(px.data(), w, h);
    s.environment_factor[0] = s.environment_factor[1] = s.environment_factor[2] = 1.0f;
    s.environment_factor[3] = 1.0f;      // vec4(envmap->get_factor(), 1) (src/scene_stage.cc:1345)
}

}
#endif
