// tauray_image.hh - texture files -> RGBA8, for both hosts (the C++ loader includes it, the Python mirror calls it through
// trhip_image_decode of include/trhip.h, so the two flatten a scene to the same bytes).
//
// The reference decodes glTF images with stb_image through tinygltf (src/gltf.cc:520-576; external/stb_image.h is vendored there
// and is not used here): PNG in every colour type and bit depth, interlaced or not, and JPEG.  This header reads
//   * PNG: grey, grey + alpha, RGB, RGBA and palette images of 1 / 2 / 4 / 8 / 16 bits per sample, tRNS transparency, Adam7
//     interlacing.  16-bit samples are kept (decoded::rgba16) and also rounded to 8 bits, v * 255 / 65535 (the reference keeps them as R16G16B16A16Unorm
//     textures: at most half an 8-bit step apart, the texel store here is RGBA8);
//   * JPEG: baseline, extended-sequential and progressive Huffman files (SOF0 / SOF1 / SOF2, 8 bits per sample; scans of all or of
//     single components), grey or YCbCr (or RGB when an Adobe marker says so), any sampling factors, restart intervals.  Chroma planes subsampled by two are interpolated linearly
//     (the triangle filter libjpeg calls "fancy upsampling"), the inverse DCT is evaluated in floating point.  JPEG leaves both
//     to the decoder, so decoders agree to a few levels, not to the bit.  Arithmetic-coded, lossless and hierarchical files are refused.
// Row 0 of the result is the top row of the file.  Needs zlib for PNG (TAURAY_HIP_WITH_ZLIB).
#ifndef TAURAY_IMAGE_HH
#define TAURAY_IMAGE_HH
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef TAURAY_HIP_WITH_ZLIB
#include <zlib.h>
#endif

namespace tr
{
namespace image
{

// rgba: 8 bits per sample, always.  bits = 16 (a PNG of 16 bits per sample): rgba16 holds the same image at its own precision - what the
// reference keeps as R16G16B16A16Unorm (src/gltf.cc:548-556; stb_image expands grey / RGB to RGBA with alpha 65535) - and rgba its
// rounding to 8 bits for callers that want bytes.
struct decoded { uint32_t w = 0, h = 0; int channels_in_file = 0; int bits = 8; std::vector<uint8_t> rgba; std::vector<uint16_t> rgba16; };

//---------------------------------------------------------------------------------------------------------------------
// PNG
inline decoded decode_png(const uint8_t* data, size_t size)
{
#ifndef TAURAY_HIP_WITH_ZLIB
    (void)data; (void)size;
    throw std::runtime_error("image: PNG needs a build with TAURAY_HIP_WITH_ZLIB");
#else
    auto be32 = [&](size_t o) { return (uint32_t(data[o]) << 24) | (uint32_t(data[o + 1]) << 16) | (uint32_t(data[o + 2]) << 8) | uint32_t(data[o + 3]); };
    size_t pos = 8;
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, ctype = -1, interlace = 0;
    decoded out;
    while(pos + 12 <= size)
    {
        const uint32_t len = be32(pos);
        const char* tag = reinterpret_cast<const char*>(data + pos + 4);
        const uint8_t* body = data + pos + 8;
        if(pos + 12 + size_t(len) > size) throw std::runtime_error("image: truncated PNG");
        if(!std::strncmp(tag, "IHDR", 4) && len >= 13) { out.w = be32(pos + 8); out.h = be32(pos + 12); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if(!std::strncmp(tag, "PLTE", 4)) plte.assign(body, body + len);
        else if(!std::strncmp(tag, "tRNS", 4)) trns.assign(body, body + len);
        else if(!std::strncmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if(!std::strncmp(tag, "IEND", 4)) break;
        pos += 12 + size_t(len);
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (depth == 8) || (depth == 16 && ctype != 3) || ((depth == 1 || depth == 2 || depth == 4) && (ctype == 0 || ctype == 3));
    if(ch == 0 || !depth_ok || out.w == 0 || out.h == 0 || interlace > 1) throw std::runtime_error("image: unsupported PNG header");
    if(ctype == 3 && plte.size() < 3) throw std::runtime_error("image: palette PNG without a palette");
    const size_t w = out.w, h = out.h;
    const int bits_per_pixel = ch * depth;
    const size_t bpp = (size_t)std::max(1, bits_per_pixel / 8);           // filter distance in bytes
    auto row_bytes = [&](size_t pw) { return (pw * (size_t)bits_per_pixel + 7) / 8; };
    // passes: {x0, y0, dx, dy}; a non-interlaced image is one pass over everything
    static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    size_t total = 0;
    const int n_pass = interlace ? 7 : 1;
    size_t pw[7], ph[7];
    for(int p = 0; p < n_pass; ++p)
    {
        pw[p] = interlace ? (w + (size_t)adam7[p][2] - 1 - (size_t)adam7[p][0]) / (size_t)adam7[p][2] : w;
        ph[p] = interlace ? (h + (size_t)adam7[p][3] - 1 - (size_t)adam7[p][1]) / (size_t)adam7[p][3] : h;
        if(interlace && ((size_t)adam7[p][0] >= w || (size_t)adam7[p][1] >= h)) pw[p] = ph[p] = 0;
        if(pw[p] && ph[p]) total += (row_bytes(pw[p]) + 1) * ph[p];
    }
    // A header that promises more than its data can hold (a flipped bit in the width is enough) must not cost gigabytes and minutes
    // before the inflate fails: deflate expands by at most 1032 : 1, and stb_image's cap on a side (1 << 24) bounds the arithmetic.
    if(w > (1u << 24) || h > (1u << 24) || total > idat.size() * 1032 + 1024) throw std::runtime_error("image: PNG dimensions do not fit its data");
    std::vector<uint8_t> raw(total);
    uLongf raw_len = (uLongf)raw.size();
    if(uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) throw std::runtime_error("image: PNG inflate failed");
    // samples of one pixel as 16-bit values (8-bit and smaller ones scaled up at the end)
    std::vector<uint16_t> px(w * h * (size_t)ch);
    size_t off = 0;
    std::vector<uint8_t> cur, prev;
    for(int p = 0; p < n_pass; ++p)
    {
        if(!pw[p] || !ph[p]) continue;
        const size_t rb = row_bytes(pw[p]);
        prev.assign(rb, 0);
        cur.resize(rb);
        for(size_t y = 0; y < ph[p]; ++y)
        {
            const uint8_t ft = raw[off];
            const uint8_t* line = raw.data() + off + 1;
            off += rb + 1;
            for(size_t x = 0; x < rb; ++x)
            {
                const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
                int pred;
                switch(ft)
                {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: { const int q = a + b - c, pa = std::abs(q - a), pb = std::abs(q - b), pc = std::abs(q - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: throw std::runtime_error("image: bad PNG filter");
                }
                cur[x] = (uint8_t)((line[x] + pred) & 255);
            }
            const size_t oy = interlace ? (size_t)adam7[p][1] + y * (size_t)adam7[p][3] : y;
            for(size_t x = 0; x < pw[p]; ++x)
            {
                const size_t ox = interlace ? (size_t)adam7[p][0] + x * (size_t)adam7[p][2] : x;
                uint16_t* d = px.data() + (oy * w + ox) * (size_t)ch;
                for(int k = 0; k < ch; ++k)
                {
                    const size_t s = x * (size_t)ch + (size_t)k;
                    if(depth == 16) d[k] = (uint16_t)((cur[2 * s] << 8) | cur[2 * s + 1]);
                    else if(depth == 8) d[k] = cur[s];
                    else { const size_t bit = s * (size_t)depth; d[k] = (uint16_t)((cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1 << depth) - 1)); }
                }
            }
            std::swap(cur, prev);
        }
    }
    const int maxv = (1 << depth) - 1;
    auto to8 = [&](uint16_t v) -> uint8_t { return depth == 8 ? (uint8_t)v : (uint8_t)((uint32_t(v) * 255u + (uint32_t)maxv / 2u) / (uint32_t)maxv); };
    out.rgba.resize(w * h * 4);
    out.channels_in_file = ctype == 3 ? (trns.empty() ? 3 : 4) : ch + ((ctype == 0 || ctype == 2) && !trns.empty() ? 1 : 0);
    if(depth == 16)
    {   // the samples as they are (ctype 3 has no 16-bit form)
        out.bits = 16;
        out.rgba16.resize(w * h * 4);
        for(size_t i = 0; i < w * h; ++i)
        {
            const uint16_t* s = px.data() + i * (size_t)ch;
            uint16_t* d = out.rgba16.data() + i * 4;
            if(ch <= 2)
            {
                d[0] = d[1] = d[2] = s[0];
                d[3] = ch == 2 ? s[1] : 65535;
                if(ch == 1 && trns.size() >= 2 && s[0] == (uint16_t)((trns[0] << 8) | trns[1])) d[3] = 0;
            }
            else
            {
                d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
                d[3] = ch == 4 ? s[3] : 65535;
                if(ch == 3 && trns.size() >= 6 && s[0] == (uint16_t)((trns[0] << 8) | trns[1]) && s[1] == (uint16_t)((trns[2] << 8) | trns[3]) &&
                   s[2] == (uint16_t)((trns[4] << 8) | trns[5])) d[3] = 0;
            }
        }
    }
    for(size_t i = 0; i < w * h; ++i)
    {
        const uint16_t* s = px.data() + i * (size_t)ch;
        uint8_t* d = out.rgba.data() + i * 4;
        if(ctype == 3)
        {
            const size_t idx = s[0];
            if(3 * idx + 2 >= plte.size()) throw std::runtime_error("image: PNG palette index out of range");
            d[0] = plte[3 * idx]; d[1] = plte[3 * idx + 1]; d[2] = plte[3 * idx + 2];
            d[3] = idx < trns.size() ? trns[idx] : 255;
        }
        else if(ch <= 2)
        {
            d[0] = d[1] = d[2] = to8(s[0]);
            d[3] = ch == 2 ? to8(s[1]) : 255;
            if(ch == 1 && trns.size() >= 2 && s[0] == (uint16_t)((trns[0] << 8) | trns[1])) d[3] = 0;
        }
        else
        {
            d[0] = to8(s[0]); d[1] = to8(s[1]); d[2] = to8(s[2]);
            d[3] = ch == 4 ? to8(s[3]) : 255;
            if(ch == 3 && trns.size() >= 6 && s[0] == (uint16_t)((trns[0] << 8) | trns[1]) && s[1] == (uint16_t)((trns[2] << 8) | trns[3]) &&
               s[2] == (uint16_t)((trns[4] << 8) | trns[5])) d[3] = 0;
        }
    }
    return out;
#endif
}

//---------------------------------------------------------------------------------------------------------------------
// JPEG (baseline / extended sequential / progressive, Huffman, 8 bit)
namespace jpeg_detail
{
struct huffman
{
    // canonical code of `length` bits -> symbol: codes of one length are consecutive (ITU T.81 annex C)
    int mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    bool defined = false;
    void build(const uint8_t counts[16], const uint8_t* symbols, int n)
    {
        std::memcpy(values, symbols, (size_t)n);
        int code = 0, k = 0;
        for(int len = 1; len <= 16; ++len)
        {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        defined = true;
    }
};

struct bit_reader
{
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0;
    bool hit_marker = false;
    int bit()
    {
        if(n == 0)
        {
            uint8_t b = 0;
            if(!hit_marker && p < end)
            {
                b = *p++;
                if(b == 0xFF)
                {
                    if(p < end && *p == 0x00) ++p;              // stuffed zero
                    else { hit_marker = true; --p; b = 0; }       // a marker: the entropy-coded segment ends here, pad with zeros
                }
            }
            acc = b; n = 8;
        }
        --n;
        return (int)((acc >> n) & 1u);
    }
    int bits(int count) { int v = 0; while(count--) v = (v << 1) | bit(); return v; }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int decode_symbol(bit_reader& br, const huffman& h)
{
    int code = 0;
    for(int len = 1; len <= 16; ++len)
    {
        code = (code << 1) | br.bit();
        if(h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
    }
    throw std::runtime_error("image: bad Huffman code in JPEG");
}
inline int extend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }

static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// 8x8 inverse DCT, separable, in double precision; result = samples + 128, clamped
inline void idct8x8(const float* coef, uint8_t* out, size_t stride)
{
    static double c[8][8];
    static bool init = false;
    if(!init)
    {
        for(int x = 0; x < 8; ++x) for(int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    double tmp[64];
    for(int v = 0; v < 8; ++v)
        for(int x = 0; x < 8; ++x)
        {
            double s = 0;
            for(int u = 0; u < 8; ++u) s += c[x][u] * coef[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    for(int y = 0; y < 8; ++y)
        for(int x = 0; x < 8; ++x)
        {
            double s = 0;
            for(int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
            const long r = std::lround(s + 128.0);
            out[(size_t)y * stride + (size_t)x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}
}  // namespace jpeg_detail

inline decoded decode_jpeg(const uint8_t* data, size_t size)
{
    using namespace jpeg_detail;
    // bw x bh: blocks of the component in whole MCUs (what interleaved scans walk); sbw x sbh: the blocks that cover the image
    // (what a scan of this component alone walks, T.81 A.2.3)
    struct component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; size_t bw = 0, bh = 0, sbw = 0, sbh = 0, pw = 0, ph = 0;
                       std::vector<int16_t> coef; std::vector<uint8_t> plane; };
    uint16_t qt[4][64] = {};
    bool qt_defined[4] = {false, false, false, false};
    huffman dc[4], ac[4];
    std::vector<component> comps;
    decoded out;
    int restart_interval = 0, adobe_transform = -1;
    int hmax = 1, vmax = 1;
    bool progressive = false;
    size_t mcux = 0, mcuy = 0;
    size_t pos = 2;
    auto be16 = [&](size_t o) { if(o + 2 > size) throw std::runtime_error("image: truncated JPEG"); return (int)((data[o] << 8) | data[o + 1]); };
    bool decoded_scan = false;
    while(pos + 4 <= size)
    {
        if(data[pos] != 0xFF) { ++pos; continue; }
        const uint8_t m = data[pos + 1];
        if(m == 0xFF) { ++pos; continue; }
        pos += 2;
        if(m == 0x00 || m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;      // stuffed bytes and restart markers left over from a scan
        if(m == 0xD9) break;
        const int len = be16(pos);
        if(len < 2 || pos + (size_t)len > size) throw std::runtime_error("image: truncated JPEG segment");
        const uint8_t* seg = data + pos + 2;
        const int n = len - 2;
        if(m == 0xDB)
        {
            for(int o = 0; o < n;)
            {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if(tq > 3 || o + 1 + 64 * (pq ? 2 : 1) > n) throw std::runtime_error("image: bad JPEG quantisation table");
                for(int i = 0; i < 64; ++i) qt[tq][zigzag[i]] = pq ? (uint16_t)((seg[o + 1 + 2 * i] << 8) | seg[o + 2 + 2 * i]) : seg[o + 1 + i];
                qt_defined[tq] = true;
                o += 1 + 64 * (pq ? 2 : 1);
            }
        }
        else if(m == 0xC4)
        {
            for(int o = 0; o + 17 <= n;)
            {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                int total = 0;
                for(int i = 0; i < 16; ++i) total += seg[o + 1 + i];
                if(th > 3 || tc > 1 || total > 256 || o + 17 + total > n) throw std::runtime_error("image: bad JPEG Huffman table");
                (tc ? ac : dc)[th].build(seg + o + 1, seg + o + 17, total);
                o += 17 + total;
            }
        }
        else if(m == 0xC0 || m == 0xC1 || m == 0xC2)
        {
            if(!comps.empty()) throw std::runtime_error("image: JPEG with more than one frame");
            if(n < 6 || seg[0] != 8) throw std::runtime_error("image: JPEG with other than 8 bits per sample");
            progressive = m == 0xC2;
            out.h = (uint32_t)((seg[1] << 8) | seg[2]); out.w = (uint32_t)((seg[3] << 8) | seg[4]);
            const int nc = seg[5];
            if((nc != 1 && nc != 3) || n < 6 + 3 * nc || out.w == 0 || out.h == 0) throw std::runtime_error("image: JPEG with " + std::to_string(nc) + " components");
            comps.resize((size_t)nc);
            for(int i = 0; i < nc; ++i)
            {
                comps[(size_t)i].id = seg[6 + 3 * i]; comps[(size_t)i].h = seg[7 + 3 * i] >> 4; comps[(size_t)i].v = seg[7 + 3 * i] & 15; comps[(size_t)i].tq = seg[8 + 3 * i] & 3;
                if(comps[(size_t)i].h < 1 || comps[(size_t)i].h > 4 || comps[(size_t)i].v < 1 || comps[(size_t)i].v > 4) throw std::runtime_error("image: bad JPEG sampling factors");
                hmax = std::max(hmax, comps[(size_t)i].h); vmax = std::max(vmax, comps[(size_t)i].v);
            }
            mcux = (out.w + 8 * (size_t)hmax - 1) / (8 * (size_t)hmax); mcuy = (out.h + 8 * (size_t)vmax - 1) / (8 * (size_t)vmax);
            // every MCU costs the entropy-coded data at least a bit: a frame header that promises more MCUs than the file has bits is
            // damaged, and taking it at its word would allocate and transform gigabytes of coefficients
            if(mcux * mcuy > size * 8) throw std::runtime_error("image: JPEG dimensions do not fit its data");
            for(component& c: comps)
            {
                c.bw = mcux * (size_t)c.h; c.bh = mcuy * (size_t)c.v;
                c.sbw = ((out.w * (size_t)c.h + (size_t)hmax - 1) / (size_t)hmax + 7) / 8;
                c.sbh = ((out.h * (size_t)c.v + (size_t)vmax - 1) / (size_t)vmax + 7) / 8;
                c.coef.assign(c.bw * c.bh * 64, 0);
            }
        }
        else if((m >= 0xC3 && m <= 0xCF && m != 0xC8 && m != 0xCC))
            throw std::runtime_error("image: unsupported JPEG coding process (lossless, hierarchical or arithmetic)");
        else if(m == 0xDD && n >= 2) restart_interval = (seg[0] << 8) | seg[1];
        else if(m == 0xEE && n >= 12 && !std::memcmp(seg, "Adobe", 5)) adobe_transform = seg[11];
        else if(m == 0xDA)
        {
            if(comps.empty()) throw std::runtime_error("image: JPEG scan before the frame header");
            const int ns = n >= 1 ? seg[0] : 0;
            if(ns < 1 || ns > (int)comps.size() || n < 1 + 2 * ns + 3) throw std::runtime_error("image: bad JPEG scan header");
            component* sc[4] = {nullptr, nullptr, nullptr, nullptr};
            const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
            if(!progressive ? (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) : (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || Al > 13 || Ah > 13))
                throw std::runtime_error("image: bad JPEG scan parameters");
            for(int i = 0; i < ns; ++i)
            {
                for(component& k: comps) if(k.id == seg[1 + 2 * i]) sc[i] = &k;
                if(!sc[i]) throw std::runtime_error("image: JPEG scan names an unknown component");
                component* c = sc[i];
                c->td = seg[2 + 2 * i] >> 4; c->ta = seg[2 + 2 * i] & 15;
                const bool need_dc = Ss == 0 && Ah == 0, need_ac = Se > 0;
                if(c->td > 3 || c->ta > 3 || (need_dc && !dc[c->td].defined) || (need_ac && !ac[c->ta].defined)) throw std::runtime_error("image: JPEG scan uses an undefined table");
                c->pred = 0;
            }
            bit_reader br{data + pos + (size_t)len, data + size};
            int eobrun = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            // one 8x8 block of this scan (T.81 F.2.2 sequential; G.2 progressive: DC first / refinement, AC first / refinement with
            // end-of-band runs)
            auto refine = [&](int16_t& v) { if(br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1)); };
            auto block = [&](component& c, int16_t* b) {
                if(!progressive)
                {
                    const int t = decode_symbol(br, dc[c.td]);
                    if(t > 15) throw std::runtime_error("image: JPEG DC difference category out of range");   // T.81 F.1.2.1: SSSS <= 11 (15 at 12 bits); a DHT symbol is any byte
                    c.pred += extend(br.bits(t), t);
                    b[0] = (int16_t)c.pred;
                    for(int k = 1; k < 64;)
                    {
                        const int rs = decode_symbol(br, ac[c.ta]);
                        const int r = rs >> 4, s = rs & 15;
                        if(s == 0) { if(r == 15) { k += 16; continue; } break; }
                        k += r;
                        if(k > 63) throw std::runtime_error("image: JPEG coefficient index out of range");
                        b[zigzag[k]] = (int16_t)extend(br.bits(s), s);
                        ++k;
                    }
                }
                else if(Ss == 0)
                {
                    if(Ah == 0)
                    {
                        const int t = decode_symbol(br, dc[c.td]);
                        if(t > 15) throw std::runtime_error("image: JPEG DC difference category out of range");
                        c.pred += extend(br.bits(t), t);
                        b[0] = (int16_t)(c.pred * p1);
                    }
                    else if(br.bit()) b[0] = (int16_t)(b[0] | p1);
                }
                else if(Ah == 0)
                {
                    if(eobrun > 0) { --eobrun; return; }
                    for(int k = Ss; k <= Se;)
                    {
                        const int rs = decode_symbol(br, ac[c.ta]);
                        const int r = rs >> 4, s = rs & 15;
                        if(s == 0)
                        {
                            if(r < 15) { eobrun = (1 << r) - 1; if(r) eobrun += br.bits(r); break; }
                            k += 16;
                            continue;
                        }
                        k += r;
                        if(k > Se) throw std::runtime_error("image: JPEG coefficient index out of range");
                        b[zigzag[k]] = (int16_t)(extend(br.bits(s), s) * p1);
                        ++k;
                    }
                }
                else
                {
                    int k = Ss;
                    if(eobrun == 0)
                    {
                        for(; k <= Se; ++k)
                        {
                            const int rs = decode_symbol(br, ac[c.ta]);
                            int r = rs >> 4;
                            const int s = rs & 15;
                            int value = 0;
                            if(s)
                            {
                                if(s != 1) throw std::runtime_error("image: bad JPEG refinement code");
                                value = br.bit() ? p1 : m1;
                            }
                            else if(r != 15)
                            {
                                eobrun = 1 << r;
                                if(r) eobrun += br.bits(r);
                                break;
                            }
                            // over the coefficients that are non-zero already (one correction bit each) and r zero ones
                            for(; k <= Se; ++k)
                            {
                                int16_t& v = b[zigzag[k]];
                                if(v != 0) refine(v);
                                else if(--r < 0) break;
                            }
                            if(s && k <= Se) b[zigzag[k]] = (int16_t)value;
                        }
                    }
                    if(eobrun > 0)
                    {
                        for(; k <= Se; ++k) { int16_t& v = b[zigzag[k]]; if(v != 0) refine(v); }
                        --eobrun;
                    }
                }
            };
            const bool interleaved = ns > 1;
            const size_t nx = interleaved ? mcux : sc[0]->sbw, ny = interleaved ? mcuy : sc[0]->sbh;
            size_t mcu_count = 0;
            for(size_t my = 0; my < ny; ++my)
                for(size_t mx = 0; mx < nx; ++mx)
                {
                    if(restart_interval && mcu_count && mcu_count % (size_t)restart_interval == 0)
                    {
                        // skip to the RSTn marker, reset the predictors and the end-of-band run
                        const uint8_t* q = br.p;
                        while(q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                        br.p = q + 2 <= br.end ? q + 2 : br.end;
                        br.reset();
                        for(int i = 0; i < ns; ++i) sc[i]->pred = 0;
                        eobrun = 0;
                    }
                    if(interleaved)
                    {
                        for(int i = 0; i < ns; ++i)
                            for(int by = 0; by < sc[i]->v; ++by)
                                for(int bx = 0; bx < sc[i]->h; ++bx)
                                    block(*sc[i], sc[i]->coef.data() + ((my * (size_t)sc[i]->v + (size_t)by) * sc[i]->bw + mx * (size_t)sc[i]->h + (size_t)bx) * 64);
                    }
                    else block(*sc[0], sc[0]->coef.data() + (my * sc[0]->bw + mx) * 64);
                    ++mcu_count;
                }
            decoded_scan = true;
            pos = (size_t)(br.p - data);      // markers and stuffed bytes the reader stopped in front of are skipped above
            continue;
        }
        pos += (size_t)len;
    }
    if(!decoded_scan) throw std::runtime_error("image: JPEG without image data");
    // coefficients -> samples
    for(component& c: comps)
    {
        if(!qt_defined[c.tq]) throw std::runtime_error("image: JPEG component without a quantisation table");
        c.pw = c.bw * 8; c.ph = c.bh * 8;
        c.plane.assign(c.pw * c.ph, 0);
        float blk[64];
        for(size_t by = 0; by < c.bh; ++by)
            for(size_t bx = 0; bx < c.bw; ++bx)
            {
                const int16_t* q = c.coef.data() + (by * c.bw + bx) * 64;
                for(int i = 0; i < 64; ++i) blk[i] = (float)((int)q[i] * (int)qt[c.tq][i]);
                idct8x8(blk, c.plane.data() + by * 8 * c.pw + bx * 8, c.pw);
            }
        std::vector<int16_t>().swap(c.coef);
    }
    // upsample every component to full resolution
    const size_t W = out.w, H = out.h;
    std::vector<std::vector<uint8_t>> full(comps.size());
    for(size_t ci = 0; ci < comps.size(); ++ci)
    {
        const component& c = comps[ci];
        const int fx = hmax / c.h, fy = vmax / c.v;
        if(hmax % c.h || vmax % c.v) throw std::runtime_error("image: fractional JPEG sampling ratios are not read");
        // the part of the plane that belongs to the image
        const size_t cw = (W * (size_t)c.h + (size_t)hmax - 1) / (size_t)hmax, chh = (H * (size_t)c.v + (size_t)vmax - 1) / (size_t)vmax;
        std::vector<uint8_t>& f = full[ci];
        f.resize(W * H);
        auto at = [&](long x, long y) -> int { x = x < 0 ? 0 : (x >= (long)cw ? (long)cw - 1 : x); y = y < 0 ? 0 : (y >= (long)chh ? (long)chh - 1 : y); return c.plane[(size_t)y * c.pw + (size_t)x]; };
        for(size_t y = 0; y < H; ++y)
            for(size_t x = 0; x < W; ++x)
            {
                int v;
                if(fx == 1 && fy == 1) v = at((long)x, (long)y);
                else if((fx == 1 || fx == 2) && (fy == 1 || fy == 2))
                {
                    // triangle filter: the nearer source sample weighs 3, the farther 1 (per axis that is halved)
                    const long sx = fx == 2 ? (long)(x >> 1) : (long)x, sy = fy == 2 ? (long)(y >> 1) : (long)y;
                    const long nx = fx == 2 ? ((x & 1) ? sx + 1 : sx - 1) : sx, ny = fy == 2 ? ((y & 1) ? sy + 1 : sy - 1) : sy;
                    if(fx == 2 && fy == 2) v = (9 * at(sx, sy) + 3 * at(nx, sy) + 3 * at(sx, ny) + at(nx, ny) + 8) >> 4;
                    else if(fx == 2) v = (3 * at(sx, sy) + at(nx, sy) + 2) >> 2;
                    else v = (3 * at(sx, sy) + at(sx, ny) + 2) >> 2;
                }
                else v = at((long)(x / (size_t)fx), (long)(y / (size_t)fy));
                f[y * W + x] = (uint8_t)v;
            }
    }
    out.channels_in_file = (int)comps.size();
    out.rgba.resize(W * H * 4);
    const bool ycc = comps.size() == 3 && adobe_transform != 0;
    auto clamp8 = [](double v) { const long r = std::lround(v); return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r)); };
    for(size_t i = 0; i < W * H; ++i)
    {
        uint8_t* d = out.rgba.data() + 4 * i;
        if(comps.size() == 1) d[0] = d[1] = d[2] = full[0][i];
        else if(ycc)
        {
            const double Y = full[0][i], cb = full[1][i] - 128.0, cr = full[2][i] - 128.0;
            d[0] = clamp8(Y + 1.402 * cr); d[1] = clamp8(Y - 0.344136 * cb - 0.714136 * cr); d[2] = clamp8(Y + 1.772 * cb);
        }
        else { d[0] = full[0][i]; d[1] = full[1][i]; d[2] = full[2][i]; }
        d[3] = 255;
    }
    return out;
}

// Any supported file by its signature.
inline decoded decode(const uint8_t* data, size_t size)
{
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if(size >= 8 && !std::memcmp(data, png_sig, 8)) return decode_png(data, size);
    if(size >= 3 && data[0] == 0xFF && data[1] == 0xD8 && data[2] == 0xFF) return decode_jpeg(data, size);
    throw std::runtime_error("image: neither a PNG nor a JPEG file");
}

}  // namespace image
}  // namespace tr
#endif
