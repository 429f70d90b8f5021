// tauray_exr.hh - OpenEXR files for the host layers, written against the file format (OpenEXR "Technical Introduction" /
// "OpenEXR File Layout" and the published description of the PIZ codec), not against a library:
//
//   tr::exr::read        single-part scanline or tiled (one level) images; pixel types UINT / HALF / FLOAT; compression NONE,
//                        RLE, ZIPS, ZIP, PIZ.  What texture::load_from_file's read_exr hands to the reference (src/texture.cc:70-163,
//                        via tinyexr): `.exr` environment maps and textures.  Multi-part, deep, sub-sampled channels and the lossy
//                        codecs (PXR24, B44, DWA) are refused with a message.
//   tr::exr::load_exr    read + the channel mapping of read_exr: R / G / B / A by the first letter of the name, file order when a
//                        channel is called anything else; at most four channels, interleaved floats (src/texture.cc:101-158).
//   tr::exr::encode      the scanline files headless writes (src/headless.cc:355-412): channels [A,] B, G, R, half or float, any
//                        of the five codecs above; PIZ is the reference's default (src/headless.hh:56).
//
// PIZ in short: the 16-bit words of a block of 32 scanlines, channel by channel (a float counts as two words per pixel); a
// bitmap of the values that occur maps them to a dense range; a two-dimensional Haar-like wavelet (14-bit or 16-bit modular
// arithmetic depending on the range) replaces pairs by average and difference on every scale; the result is Huffman-coded with
// canonical codes (lengths <= 58 bits, stored run-length packed in 6-bit fields) and one extra symbol for runs of a repeated word.
// The decoder is pinned by the reference's own golden images (tests/golden/ref_piz_*.exr, written by Tauray with tinyexr), the
// encoder by round trips and - in the build container - by tinyexr reading its files (tests/test_images.py).
#ifndef TAURAY_EXR_HH
#define TAURAY_EXR_HH
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>
#ifdef TAURAY_HIP_WITH_ZLIB
#include <zlib.h>
#endif

namespace tr
{
namespace exr
{

enum { COMP_NONE = 0, COMP_RLE = 1, COMP_ZIPS = 2, COMP_ZIP = 3, COMP_PIZ = 4 };
enum { PIXEL_UINT = 0, PIXEL_HALF = 1, PIXEL_FLOAT = 2 };

inline float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
    if(exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else if(exp == 0)
    {
        if(man == 0) bits = sign;
        else
        {
            int e = -1;
            do { man <<= 1; ++e; } while(!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3FFu) << 13;
        }
    }
    else bits = sign | (exp + 127 - 15) << 23 | man << 13;
    float f; std::memcpy(&f, &bits, 4);
    return f;
}

inline uint16_t float_to_half(float f)      // round to nearest even, overflow to infinity
{
    uint32_t x; std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, man = x & 0x7FFFFFu;
    int32_t exp = (int32_t)((x >> 23) & 0xFF);
    if(exp == 255) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    exp = exp - 127 + 15;
    if(exp >= 31) return (uint16_t)(sign | 0x7C00u);
    if(exp <= 0)
    {
        if(exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        int shift = 14 - exp;
        uint32_t h = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if(rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13), rem = man & 0x1FFFu;
    if(rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

//----------------------------------------------------------------------------------------------------------------------
// The byte predictor of the RLE and ZIP codecs: bytes at even positions first, then the odd ones; every byte but the first
// replaced by its difference to the one before, biased by 128.
inline void predictor_forward(const uint8_t* raw, size_t n, std::vector<uint8_t>& t)
{
    const size_t half = (n + 1) / 2;
    t.resize(n);
    for(size_t i = 0; i < n; ++i) t[(i & 1) ? half + i / 2 : i / 2] = raw[i];
    for(size_t i = n; i-- > 1;) t[i] = (uint8_t)(t[i] - t[i - 1] + 128);
}
inline void predictor_inverse(std::vector<uint8_t>& t, uint8_t* raw)
{
    const size_t n = t.size(), half = (n + 1) / 2;
    for(size_t i = 1; i < n; ++i) t[i] = (uint8_t)(t[i - 1] + t[i] - 128);
    for(size_t i = 0; i < n; ++i) raw[i] = t[(i & 1) ? half + i / 2 : i / 2];
}

// RLE: a count byte c as signed char: c < 0 -> -c literal bytes follow, c >= 0 -> the next byte c + 1 times.  Runs of 3..128.
inline void rle_pack(const std::vector<uint8_t>& t, std::vector<uint8_t>& out)
{
    out.clear();
    const size_t n = t.size();
    size_t i = 0;
    while(i < n)
    {
        size_t run = 1;
        while(i + run < n && t[i + run] == t[i] && run < 128) ++run;
        if(run >= 3)
        {
            out.push_back((uint8_t)(run - 1)); out.push_back(t[i]);
            i += run;
            continue;
        }
        // literals up to the next run of three (or 127 bytes)
        size_t j = i;
        while(j < n && j - i < 127 && !(j + 2 < n && t[j] == t[j + 1] && t[j] == t[j + 2])) ++j;
        out.push_back((uint8_t)(-(int)(j - i)));
        out.insert(out.end(), t.begin() + (ptrdiff_t)i, t.begin() + (ptrdiff_t)j);
        i = j;
    }
}
inline bool rle_unpack(const uint8_t* in, size_t n_in, std::vector<uint8_t>& t, size_t n_out)
{
    t.clear(); t.reserve(n_out);
    size_t i = 0;
    while(i < n_in)
    {
        const int c = (int8_t)in[i++];
        if(c < 0)
        {
            const size_t k = (size_t)(-c);
            if(i + k > n_in || t.size() + k > n_out) return false;
            t.insert(t.end(), in + i, in + i + k);
            i += k;
        }
        else
        {
            if(i >= n_in || t.size() + (size_t)c + 1 > n_out) return false;
            t.insert(t.end(), (size_t)c + 1, in[i++]);
        }
    }
    return t.size() == n_out;
}

//----------------------------------------------------------------------------------------------------------------------
// Huffman coder of the PIZ codec.  Symbols are 16-bit words plus one run symbol (the largest symbol in use + 1).  A code is
// at most 58 bits; codes are canonical: within a length symbols in increasing order get consecutive values, and the first
// value of length l is ceil-halved from the end of length l + 1 (longest codes start at 0).
struct huffman
{
    static constexpr int MAX_LEN = 58;
    static constexpr uint32_t SYMBOLS = 65537;

    // first code value of every length from the number of codes per length
    static void first_codes(const uint64_t count[MAX_LEN + 1], uint64_t first[MAX_LEN + 1])
    {
        uint64_t c = 0;
        for(int l = MAX_LEN; l >= 1; --l)
        {
            first[l] = c;
            c = (c + count[l]) >> 1;
        }
        first[0] = 0;
    }

    struct bit_writer
    {
        std::vector<uint8_t>& out;
        uint64_t acc = 0; int n = 0; uint64_t total = 0;
        explicit bit_writer(std::vector<uint8_t>& o): out(o) {}
        void put(int bits, uint64_t v)         // MSB first; fewer than 8 bits are pending between calls
        {
            if(bits > 32) { put(bits - 32, v >> 32); put(32, v & 0xFFFFFFFFull); return; }
            if(bits == 0) return;
            acc = (acc << bits) | (v & ((1ull << bits) - 1ull));
            n += bits; total += (uint64_t)bits;
            while(n >= 8) { out.push_back((uint8_t)(acc >> (n - 8))); n -= 8; }
        }
        void flush() { if(n > 0) { out.push_back((uint8_t)(acc << (8 - n))); n = 0; } }
    };
    struct bit_reader
    {
        const uint8_t* p; size_t bytes; uint64_t pos = 0;
        bit_reader(const uint8_t* d, size_t n): p(d), bytes(n) {}
        uint32_t get(int bits)                 // bits <= 32, MSB first; past the end: zeros
        {
            uint32_t v = 0;
            for(int i = 0; i < bits; ++i, ++pos)
            {
                const size_t b = (size_t)(pos >> 3);
                const uint32_t bit = b < bytes ? (p[b] >> (7 - (pos & 7))) & 1u : 0u;
                v = (v << 1) | bit;
            }
            return v;
        }
    };

    // code lengths from frequencies (every symbol with freq > 0 gets one); false if a code would be longer than MAX_LEN
    static bool lengths_from_frequencies(const std::vector<uint64_t>& freq, std::vector<uint8_t>& len)
    {
        const size_t n = freq.size();
        len.assign(n, 0);
        struct node { uint64_t f; uint32_t id; };
        struct cmp { bool operator()(const node& a, const node& b) const { return a.f > b.f || (a.f == b.f && a.id > b.id); } };
        std::priority_queue<node, std::vector<node>, cmp> heap;
        std::vector<int32_t> parent;
        parent.reserve(2 * n);
        std::vector<uint32_t> leaf_of;
        for(size_t s = 0; s < n; ++s)
            if(freq[s]) { heap.push({freq[s], (uint32_t)parent.size()}); parent.push_back(-1); leaf_of.push_back((uint32_t)s); }
        const size_t leaves = parent.size();
        if(leaves == 0) return true;
        if(leaves == 1) { len[leaf_of[0]] = 1; return true; }
        while(heap.size() > 1)
        {
            const node a = heap.top(); heap.pop();
            const node b = heap.top(); heap.pop();
            const uint32_t id = (uint32_t)parent.size();
            parent.push_back(-1);
            parent[a.id] = (int32_t)id; parent[b.id] = (int32_t)id;
            heap.push({a.f + b.f, id});
        }
        // depth of every node: parents have larger ids than children
        std::vector<uint8_t> depth(parent.size(), 0);
        for(size_t i = parent.size() - 1; i-- > 0;)
        {
            const int d = depth[(size_t)parent[i]] + 1;
            if(d > MAX_LEN) return false;
            depth[i] = (uint8_t)d;
        }
        for(size_t i = 0; i < leaves; ++i) len[leaf_of[i]] = depth[i];
        return true;
    }

    // `words` -> the compressed form: 20-byte header (first symbol, last symbol = the run symbol, table bytes, data bits, 0),
    // the packed length table, the bit stream.  false: not codable (caller stores the block raw).
    static bool compress(const uint16_t* words, size_t n, std::vector<uint8_t>& out)
    {
        out.clear();
        if(n == 0) return true;
        std::vector<uint64_t> freq(SYMBOLS, 0);
        for(size_t i = 0; i < n; ++i) freq[words[i]]++;
        uint32_t im = 0, iM = 0;
        while(!freq[im]) ++im;
        for(uint32_t s = im; s < 65536; ++s) if(freq[s]) iM = s;
        ++iM;                          // the run symbol
        freq[iM] = 1;
        std::vector<uint8_t> len;
        if(!lengths_from_frequencies(freq, len)) return false;
        uint64_t count[MAX_LEN + 1] = {0}, first[MAX_LEN + 1];
        for(uint32_t s = im; s <= iM; ++s) count[len[s]]++;
        count[0] = 0;
        first_codes(count, first);
        std::vector<uint64_t> code(SYMBOLS, 0);
        for(uint32_t s = im; s <= iM; ++s) if(len[s]) code[s] = first[len[s]]++;

        out.resize(20, 0);
        // ---- the table: 6 bits per symbol from im to iM; 0..58 = length; 59..62 = 2..5 zero lengths; 63 + 8 bits = 6..261 zero lengths
        {
            bit_writer w(out);
            for(uint32_t s = im; s <= iM; ++s)
            {
                if(len[s] == 0)
                {
                    uint32_t zeros = 1;
                    while(s + zeros <= iM && len[s + zeros] == 0 && zeros < 261) ++zeros;
                    if(zeros >= 2)
                    {
                        if(zeros >= 6) { w.put(6, 63); w.put(8, zeros - 6); }
                        else w.put(6, 59 + zeros - 2);
                        s += zeros - 1;
                        continue;
                    }
                }
                w.put(6, len[s]);
            }
            w.flush();
        }
        const uint32_t table_bytes = (uint32_t)(out.size() - 20);
        // ---- the data: a word, or word + run symbol + 8-bit repeat count where that is shorter than repeating the word
        bit_writer w(out);
        auto send = [&](uint16_t s, uint32_t repeats) {
            const uint64_t ls = len[s], lr = len[iM];
            if(ls + lr + 8 < ls * repeats) { w.put((int)ls, code[s]); w.put((int)lr, code[iM]); w.put(8, repeats); }
            else for(uint32_t k = 0; k <= repeats; ++k) w.put((int)ls, code[s]);
        };
        uint16_t s = words[0];
        uint32_t repeats = 0;
        for(size_t i = 1; i < n; ++i)
        {
            if(words[i] == s && repeats < 255) { ++repeats; continue; }
            send(s, repeats);
            s = words[i]; repeats = 0;
        }
        send(s, repeats);
        const uint64_t n_bits = w.total;
        w.flush();
        if(n_bits > 0xFFFFFFFFull) return false;
        const uint32_t hdr[5] = {im, iM, table_bytes, (uint32_t)n_bits, 0u};
        std::memcpy(out.data(), hdr, 20);
        return true;
    }

    static bool decompress(const uint8_t* in, size_t n_in, uint16_t* words, size_t n)
    {
        if(n == 0) return true;
        if(n_in < 20) return false;
        uint32_t hdr[5];
        std::memcpy(hdr, in, 20);
        const uint32_t im = hdr[0], iM = hdr[1];
        const uint64_t n_bits = hdr[3];
        if(im >= SYMBOLS || iM >= SYMBOLS || im > iM) return false;
        if((n_bits + 7) / 8 > n_in - 20) return false;
        std::vector<uint8_t> len(SYMBOLS, 0);
        bit_reader tr(in + 20, n_in - 20);
        for(uint32_t s = im; s <= iM; ++s)
        {
            if((tr.pos >> 3) > n_in - 20) return false;
            const uint32_t l = tr.get(6);
            if(l == 63 || l >= 59)
            {
                const uint32_t zeros = l == 63 ? tr.get(8) + 6 : l - 59 + 2;
                if(s + zeros > iM + 1) return false;
                s += zeros - 1;
            }
            else len[s] = (uint8_t)l;
        }
        const size_t table_bytes = (size_t)((tr.pos + 7) >> 3);
        uint64_t count[MAX_LEN + 1] = {0}, first[MAX_LEN + 1];
        for(uint32_t s = im; s <= iM; ++s) count[len[s]]++;
        count[0] = 0;
        first_codes(count, first);
        // symbols of a length in increasing order
        std::vector<uint32_t> start(MAX_LEN + 2, 0), sorted;
        for(int l = 1; l <= MAX_LEN; ++l) start[l + 1] = start[l] + (uint32_t)count[l];
        sorted.resize(start[MAX_LEN + 1]);
        {
            std::vector<uint32_t> fill(start.begin(), start.end());
            for(uint32_t s = im; s <= iM; ++s) if(len[s]) sorted[fill[len[s]]++] = s;
        }
        // fast table for codes of up to FAST bits: entry = symbol << 6 | length
        constexpr int FAST = 12;
        std::vector<uint32_t> fast(1u << FAST, 0);
        for(int l = 1; l <= FAST; ++l)
            for(uint64_t k = 0; k < count[l]; ++k)
            {
                const uint64_t c = first[l] + k;
                if(c >> l) return false;
                const uint32_t sym = sorted[start[l] + (uint32_t)k];
                const uint32_t base = (uint32_t)(c << (FAST - l));
                for(uint32_t j = 0; j < (1u << (FAST - l)); ++j) fast[base + j] = sym << 6 | (uint32_t)l;
            }
        const uint8_t* data = in + 20 + table_bytes;
        const size_t data_bytes = n_in - 20 - table_bytes;
        if((n_bits + 7) / 8 > data_bytes) return false;
        uint64_t pos = 0;
        auto byte = [&](size_t i) -> uint64_t { return i < data_bytes ? data[i] : 0; };
        auto peek = [&](uint64_t at, int bits) -> uint64_t {     // the next `bits` (<= 58) bits, MSB first; zeros past the end
            const size_t b = (size_t)(at >> 3);
            uint64_t hi = 0;
            for(size_t i = 0; i < 8; ++i) hi = (hi << 8) | byte(b + i);
            const int sh = (int)(at & 7);
            const uint64_t window = sh ? (hi << sh) | (byte(b + 8) >> (8 - sh)) : hi;
            return window >> (64 - bits);
        };
        size_t o = 0;
        while(pos < n_bits)
        {
            uint32_t sym; int l;
            const uint32_t e = fast[(size_t)peek(pos, FAST)];
            if(e) { sym = e >> 6; l = (int)(e & 63u); }
            else
            {
                const uint64_t w = peek(pos, MAX_LEN);
                l = 0; sym = 0;
                bool found = false;
                for(int k = FAST + 1; k <= MAX_LEN; ++k)
                {
                    if(!count[k]) continue;
                    const uint64_t c = w >> (MAX_LEN - k);
                    if(c >= first[k] && c - first[k] < count[k]) { sym = sorted[start[k] + (uint32_t)(c - first[k])]; l = k; found = true; break; }
                }
                if(!found) return false;
            }
            if(pos + (uint64_t)l > n_bits) return false;
            pos += (uint64_t)l;
            if(sym == iM)
            {
                if(o == 0 || pos + 8 > n_bits) return false;
                const uint32_t repeats = (uint32_t)peek(pos, 8);
                pos += 8;
                if(o + repeats > n) return false;
                const uint16_t v = words[o - 1];
                for(uint32_t k = 0; k < repeats; ++k) words[o++] = v;
            }
            else
            {
                if(o >= n) return false;
                words[o++] = (uint16_t)sym;
            }
        }
        return o == n;
    }
};

//----------------------------------------------------------------------------------------------------------------------
// The wavelet of the PIZ codec on one plane of 16-bit words (nx x ny, strides ox / oy in words).  `max_value` < 2^14: pairs
// become (average, difference) in 16-bit signed arithmetic; otherwise the 16-bit modular variant.
struct wavelet
{
    static void enc14(uint16_t a, uint16_t b, uint16_t& l, uint16_t& h)
    {
        const int as = (int16_t)a, bs = (int16_t)b;
        l = (uint16_t)(int16_t)((as + bs) >> 1);
        h = (uint16_t)(int16_t)(as - bs);
    }
    static void dec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
    {
        const int ls = (int16_t)l, hs = (int16_t)h;
        const int ai = ls + (hs & 1) + (hs >> 1);
        a = (uint16_t)(int16_t)ai;
        b = (uint16_t)(int16_t)(ai - hs);
    }
    static void enc16(uint16_t a, uint16_t b, uint16_t& l, uint16_t& h)
    {
        const int ao = ((int)a + 0x8000) & 0xFFFF;
        int m = (ao + (int)b) >> 1;
        int d = ao - (int)b;
        if(d < 0) m = (m + 0x8000) & 0xFFFF;
        d &= 0xFFFF;
        l = (uint16_t)m; h = (uint16_t)d;
    }
    static void dec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b)
    {
        const int m = l, d = h;
        const int bb = (m - (d >> 1)) & 0xFFFF;
        const int aa = (d + bb - 0x8000) & 0xFFFF;
        b = (uint16_t)bb; a = (uint16_t)aa;
    }

    template <bool FORWARD>
    static void level(uint16_t* in, int nx, int ox, int ny, int oy, int p, int p2, bool w14)
    {
        auto pair = [&](uint16_t x, uint16_t y, uint16_t& u, uint16_t& v) {
            if(FORWARD) { if(w14) enc14(x, y, u, v); else enc16(x, y, u, v); }
            else { if(w14) dec14(x, y, u, v); else dec16(x, y, u, v); }
        };
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t* py = in;
        uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        for(; py <= ey; py += oy2)
        {
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for(; px <= ex; px += ox2)
            {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                uint16_t i00, i01, i10, i11;
                if(FORWARD)
                {
                    pair(*px, *p01, i00, i01); pair(*p10, *p11, i10, i11);
                    pair(i00, i10, *px, *p10); pair(i01, i11, *p01, *p11);
                }
                else
                {
                    pair(*px, *p10, i00, i10); pair(*p01, *p11, i01, i11);
                    pair(i00, i01, *px, *p01); pair(i10, i11, *p10, *p11);
                }
            }
            if(nx & p)       // a last column without a right-hand neighbour
            {
                uint16_t* p10 = px + oy1;
                uint16_t i00;
                pair(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if(ny & p)           // a last row without a neighbour below
        {
            uint16_t* px = py;
            uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for(; px <= ex; px += ox2)
            {
                uint16_t* p01 = px + ox1;
                uint16_t i00;
                pair(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
    }
    static void encode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t max_value)
    {
        const bool w14 = max_value < (1 << 14);
        const int n = std::min(nx, ny);
        for(int p = 1, p2 = 2; p2 <= n; p = p2, p2 <<= 1) level<true>(in, nx, ox, ny, oy, p, p2, w14);
    }
    static void decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t max_value)
    {
        const bool w14 = max_value < (1 << 14);
        const int n = std::min(nx, ny);
        int p = 1;
        while(p <= n) p <<= 1;
        p >>= 1;
        int p2 = p;
        p >>= 1;
        for(; p >= 1; p2 = p, p >>= 1) level<false>(in, nx, ox, ny, oy, p, p2, w14);
    }
};

// One PIZ block.  `raw`: the block as it would be stored uncompressed (scanline after scanline, within a scanline channel
// after channel, little-endian words); words_per_pixel[c] = 1 for HALF, 2 for UINT / FLOAT.  false = store raw.
inline bool piz_compress(const uint8_t* raw, size_t n_raw, int nx, int ny, const std::vector<int>& words_per_pixel, std::vector<uint8_t>& out)
{
    const size_t n_words = n_raw / 2;
    std::vector<uint16_t> tmp(n_words);
    std::vector<size_t> chan_start(words_per_pixel.size() + 1, 0);
    for(size_t c = 0; c < words_per_pixel.size(); ++c) chan_start[c + 1] = chan_start[c] + (size_t)nx * ny * (size_t)words_per_pixel[c];
    if(chan_start.back() != n_words) throw std::runtime_error("EXR: PIZ block size does not match its channels");
    {
        const uint8_t* p = raw;
        for(int y = 0; y < ny; ++y)
            for(size_t c = 0; c < words_per_pixel.size(); ++c)
            {
                const size_t k = (size_t)nx * (size_t)words_per_pixel[c];
                std::memcpy(tmp.data() + chan_start[c] + (size_t)y * k, p, k * 2);
                p += k * 2;
            }
    }
    std::vector<uint8_t> bitmap(8192, 0);
    for(size_t i = 0; i < n_words; ++i) bitmap[tmp[i] >> 3] |= (uint8_t)(1u << (tmp[i] & 7));
    bitmap[0] &= (uint8_t)~1u;         // zero is always part of the range, never stored
    int min_nz = 8191, max_nz = 0;
    for(int i = 0; i < 8192; ++i) if(bitmap[i]) { min_nz = std::min(min_nz, i); max_nz = std::max(max_nz, i); }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for(uint32_t i = 0; i < 65536; ++i) if(i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[i] = (uint16_t)k++;
    const uint16_t max_value = (uint16_t)(k - 1);
    for(size_t i = 0; i < n_words; ++i) tmp[i] = lut[tmp[i]];
    for(size_t c = 0; c < words_per_pixel.size(); ++c)
        for(int j = 0; j < words_per_pixel[c]; ++j)
            wavelet::encode(tmp.data() + chan_start[c] + j, nx, words_per_pixel[c], ny, nx * words_per_pixel[c], max_value);
    std::vector<uint8_t> huf;
    if(!huffman::compress(tmp.data(), n_words, huf)) return false;
    out.clear();
    const uint16_t mm[2] = {(uint16_t)min_nz, (uint16_t)max_nz};
    out.insert(out.end(), (const uint8_t*)mm, (const uint8_t*)mm + 4);
    if(min_nz <= max_nz) out.insert(out.end(), bitmap.begin() + min_nz, bitmap.begin() + max_nz + 1);
    const int32_t len = (int32_t)huf.size();
    out.insert(out.end(), (const uint8_t*)&len, (const uint8_t*)&len + 4);
    out.insert(out.end(), huf.begin(), huf.end());
    return out.size() < n_raw;
}

inline bool piz_decompress(const uint8_t* in, size_t n_in, uint8_t* raw, size_t n_raw, int nx, int ny, const std::vector<int>& words_per_pixel)
{
    const size_t n_words = n_raw / 2;
    std::vector<size_t> chan_start(words_per_pixel.size() + 1, 0);
    for(size_t c = 0; c < words_per_pixel.size(); ++c) chan_start[c + 1] = chan_start[c] + (size_t)nx * ny * (size_t)words_per_pixel[c];
    if(chan_start.back() != n_words || n_in < 4) return false;
    uint16_t mm[2];
    std::memcpy(mm, in, 4);
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if(mm[0] <= mm[1])
    {
        if(mm[1] >= 8192 || pos + (size_t)(mm[1] - mm[0] + 1) > n_in) return false;
        std::memcpy(bitmap.data() + mm[0], in + pos, (size_t)(mm[1] - mm[0] + 1));
        pos += (size_t)(mm[1] - mm[0] + 1);
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for(uint32_t i = 0; i < 65536; ++i) if(i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if(pos + 4 > n_in) return false;
    int32_t len;
    std::memcpy(&len, in + pos, 4);
    pos += 4;
    if(len < 0 || pos + (size_t)len > n_in) return false;
    std::vector<uint16_t> tmp(n_words);
    if(!huffman::decompress(in + pos, (size_t)len, tmp.data(), n_words)) return false;
    for(size_t c = 0; c < words_per_pixel.size(); ++c)
        for(int j = 0; j < words_per_pixel[c]; ++j)
            wavelet::decode(tmp.data() + chan_start[c] + j, nx, words_per_pixel[c], ny, nx * words_per_pixel[c], max_value);
    for(size_t i = 0; i < n_words; ++i) tmp[i] = lut[tmp[i]];
    uint8_t* p = raw;
    for(int y = 0; y < ny; ++y)
        for(size_t c = 0; c < words_per_pixel.size(); ++c)
        {
            const size_t n = (size_t)nx * (size_t)words_per_pixel[c];
            std::memcpy(p, tmp.data() + chan_start[c] + (size_t)y * n, n * 2);
            p += n * 2;
        }
    return true;
}

inline int lines_per_block(int compression)
{
    switch(compression)
    {
    case COMP_NONE: case COMP_RLE: case COMP_ZIPS: return 1;
    case COMP_ZIP: return 16;
    case COMP_PIZ: return 32;
    default: return 0;
    }
}

// a block of nx x ny pixels -> its stored form (never larger than raw: blocks that do not shrink are stored as they are)
inline void compress_block(int compression, const std::vector<uint8_t>& raw, int nx, int ny, const std::vector<int>& words_per_pixel, std::vector<uint8_t>& out)
{
    out.clear();
    std::vector<uint8_t> t;
    switch(compression)
    {
    case COMP_NONE: out = raw; return;
    case COMP_RLE:
        predictor_forward(raw.data(), raw.size(), t);
        rle_pack(t, out);
        break;
    case COMP_ZIPS: case COMP_ZIP:
    {
#ifdef TAURAY_HIP_WITH_ZLIB
        predictor_forward(raw.data(), raw.size(), t);
        uLongf len = compressBound((uLong)t.size());
        out.resize(len);
        if(compress(out.data(), &len, t.data(), (uLong)t.size()) != Z_OK) throw std::runtime_error("EXR: zlib compress failed");
        out.resize(len);
        break;
#else
        throw std::runtime_error("EXR zip compression needs a build with TAURAY_HIP_WITH_ZLIB");
#endif
    }
    case COMP_PIZ:
        if(!piz_compress(raw.data(), raw.size(), nx, ny, words_per_pixel, out)) out.clear();
        break;
    default: throw std::runtime_error("EXR: unsupported compression " + std::to_string(compression));
    }
    if(out.empty() || out.size() >= raw.size()) out = raw;
}

inline void decompress_block(int compression, const uint8_t* in, size_t n_in, std::vector<uint8_t>& raw, int nx, int ny, const std::vector<int>& words_per_pixel)
{
    if(n_in == raw.size()) { std::memcpy(raw.data(), in, n_in); return; }      // stored as it is
    if(n_in > raw.size()) throw std::runtime_error("EXR: a block is larger than its pixels");
    std::vector<uint8_t> t;
    switch(compression)
    {
    case COMP_RLE:
        if(!rle_unpack(in, n_in, t, raw.size())) throw std::runtime_error("EXR: corrupt RLE block");
        predictor_inverse(t, raw.data());
        return;
    case COMP_ZIPS: case COMP_ZIP:
    {
#ifdef TAURAY_HIP_WITH_ZLIB
        t.resize(raw.size());
        uLongf len = (uLongf)t.size();
        if(uncompress(t.data(), &len, in, (uLong)n_in) != Z_OK || len != t.size()) throw std::runtime_error("EXR: corrupt ZIP block");
        predictor_inverse(t, raw.data());
        return;
#else
        throw std::runtime_error("EXR zip compression needs a build with TAURAY_HIP_WITH_ZLIB");
#endif
    }
    case COMP_PIZ:
        if(!piz_decompress(in, n_in, raw.data(), raw.size(), nx, ny, words_per_pixel)) throw std::runtime_error("EXR: corrupt PIZ block");
        return;
    default: throw std::runtime_error("EXR: a short block in an uncompressed file");
    }
}

//----------------------------------------------------------------------------------------------------------------------
struct channel { std::string name; int pixel_type = PIXEL_HALF; };
struct image
{
    int width = 0, height = 0;
    int compression = 0;
    bool tiled = false;
    std::vector<channel> channels;              // file order (alphabetical)
    std::vector<std::vector<float>> planes;     // per channel, width * height, row 0 = top of the data window
};

inline image read(const uint8_t* d, size_t n)
{
    size_t pos = 0;
    auto need = [&](size_t k) { if(pos + k > n) throw std::runtime_error("EXR: truncated file"); };
    auto i32 = [&]() { need(4); int32_t v; std::memcpy(&v, d + pos, 4); pos += 4; return v; };
    auto str = [&]() { std::string s; while(true) { need(1); const char c = (char)d[pos++]; if(!c) break; s.push_back(c); if(s.size() > 255) throw std::runtime_error("EXR: bad string"); } return s; };
    if(i32() != 20000630) throw std::runtime_error("EXR: not an OpenEXR file");
    const uint32_t version = (uint32_t)i32();
    if((version & 0xFFu) != 2) throw std::runtime_error("EXR: unsupported file version");
    if(version & 0x1800u) throw std::runtime_error("EXR: deep and multi-part files are not supported");     // read_exr returns nullptr for these (src/texture.cc:81)
    image img;
    img.tiled = (version & 0x200u) != 0;
    int32_t dw[4] = {0, 0, -1, -1};
    int32_t tile_w = 0, tile_h = 0, tile_mode = 0;
    bool have_channels = false, have_dw = false, have_comp = false;
    int line_order = 0;
    while(true)
    {
        const std::string name = str();
        if(name.empty()) break;
        const std::string type = str();
        const int32_t size = i32();
        if(size < 0) throw std::runtime_error("EXR: bad attribute size");
        need((size_t)size);
        const uint8_t* a = d + pos;
        if(name == "channels")
        {
            size_t q = 0;
            while(q < (size_t)size && a[q])
            {
                channel c;
                while(q < (size_t)size && a[q]) c.name.push_back((char)a[q++]);
                ++q;
                if(q + 16 > (size_t)size) throw std::runtime_error("EXR: bad channel list");
                int32_t f[4];
                std::memcpy(&f[0], a + q, 4); std::memcpy(&f[2], a + q + 8, 4); std::memcpy(&f[3], a + q + 12, 4);
                q += 16;
                c.pixel_type = f[0];
                if(c.pixel_type < 0 || c.pixel_type > 2) throw std::runtime_error("EXR: unknown pixel type");
                if(f[2] != 1 || f[3] != 1) throw std::runtime_error("EXR: sub-sampled channels are not supported");
                img.channels.push_back(c);
            }
            have_channels = true;
        }
        else if(name == "compression" && size >= 1) { img.compression = a[0]; have_comp = true; }
        else if(name == "dataWindow" && size >= 16) { std::memcpy(dw, a, 16); have_dw = true; }
        else if(name == "lineOrder" && size >= 1) line_order = a[0];
        else if(name == "tiles" && size >= 9) { std::memcpy(&tile_w, a, 4); std::memcpy(&tile_h, a + 4, 4); tile_mode = a[8]; }
        pos += (size_t)size;
    }
    (void)line_order;       // every chunk carries its own position
    if(!have_channels || !have_dw || !have_comp || img.channels.empty()) throw std::runtime_error("EXR: a required attribute is missing");
    if(lines_per_block(img.compression) == 0) throw std::runtime_error("EXR: compression " + std::to_string(img.compression) + " (PXR24 / B44 / DWA) is not supported");
    img.width = dw[2] - dw[0] + 1; img.height = dw[3] - dw[1] + 1;
    if(img.width <= 0 || img.height <= 0 || (int64_t)img.width * img.height > (int64_t)1 << 31) throw std::runtime_error("EXR: bad data window");
    std::vector<int> wpp;
    size_t pixel_bytes = 0;
    for(const channel& c : img.channels) { wpp.push_back(c.pixel_type == PIXEL_HALF ? 1 : 2); pixel_bytes += c.pixel_type == PIXEL_HALF ? 2 : 4; }
    // A data window that promises more than the file can hold (one damaged byte is enough) must not cost gigabytes before a chunk fails:
    // none of the codecs expands by more than about a thousand to one (deflate 1032 : 1; PIZ's run code repeats a word 256 times)
    if((uint64_t)img.width * (uint64_t)img.height * pixel_bytes > (uint64_t)n * 2048 + 65536) throw std::runtime_error("EXR: the data window does not fit the file");
    img.planes.assign(img.channels.size(), std::vector<float>((size_t)img.width * img.height, 0.0f));
    std::vector<uint8_t> raw;
    // a decoded block -> planes
    auto scatter = [&](int x0, int y0, int nx, int ny) {
        const uint8_t* p = raw.data();
        for(int y = 0; y < ny; ++y)
            for(size_t c = 0; c < img.channels.size(); ++c)
            {
                float* dst = img.planes[c].data() + (size_t)(y0 + y) * img.width + x0;
                switch(img.channels[c].pixel_type)
                {
                case PIXEL_HALF: for(int x = 0; x < nx; ++x) { uint16_t h; std::memcpy(&h, p, 2); p += 2; dst[x] = half_to_float(h); } break;
                case PIXEL_FLOAT: std::memcpy(dst, p, (size_t)nx * 4); p += (size_t)nx * 4; break;
                default: for(int x = 0; x < nx; ++x) { uint32_t u; std::memcpy(&u, p, 4); p += 4; dst[x] = (float)u; } break;
                }
            }
    };
    auto u64 = [&](size_t at) { if(at + 8 > n) throw std::runtime_error("EXR: truncated offset table"); uint64_t v; std::memcpy(&v, d + at, 8); return v; };
    if(!img.tiled)
    {
        const int lines = lines_per_block(img.compression);
        const int blocks = (img.height + lines - 1) / lines;
        const size_t table = pos;
        for(int b = 0; b < blocks; ++b)
        {
            size_t at = (size_t)u64(table + 8 * (size_t)b);
            if(at > n || n - at < 8) throw std::runtime_error("EXR: bad chunk offset");
            int32_t y, size;
            std::memcpy(&y, d + at, 4); std::memcpy(&size, d + at + 4, 4);
            at += 8;
            const int row = y - dw[1];
            if(row < 0 || row >= img.height || size < 0 || (size_t)size > n - at) throw std::runtime_error("EXR: bad chunk");
            const int ny = std::min(lines, img.height - row);
            raw.resize(pixel_bytes * (size_t)img.width * (size_t)ny);
            decompress_block(img.compression, d + at, (size_t)size, raw, img.width, ny, wpp);
            scatter(0, row, img.width, ny);
        }
    }
    else
    {
        if(tile_w <= 0 || tile_h <= 0) throw std::runtime_error("EXR: bad tile description");
        // level (0, 0) tiles come first in the offset table for every level mode
        (void)tile_mode;
        const int tx_n = (img.width + tile_w - 1) / tile_w, ty_n = (img.height + tile_h - 1) / tile_h;
        const size_t table = pos;
        for(int t = 0; t < tx_n * ty_n; ++t)
        {
            size_t at = (size_t)u64(table + 8 * (size_t)t);
            if(at > n || n - at < 20) throw std::runtime_error("EXR: bad tile offset");
            int32_t h[5];
            std::memcpy(h, d + at, 20);
            at += 20;
            if(h[2] != 0 || h[3] != 0) continue;
            if(h[0] < 0 || h[0] >= tx_n || h[1] < 0 || h[1] >= ty_n || h[4] < 0 || (size_t)h[4] > n - at) throw std::runtime_error("EXR: bad tile");
            const int x0 = h[0] * tile_w, y0 = h[1] * tile_h;
            const int nx = std::min(tile_w, img.width - x0), ny = std::min(tile_h, img.height - y0);
            raw.resize(pixel_bytes * (size_t)nx * (size_t)ny);
            decompress_block(img.compression, d + at, (size_t)h[4], raw, nx, ny, wpp);
            scatter(x0, y0, nx, ny);
        }
    }
    return img;
}

inline std::vector<uint8_t> read_file(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("Failed to open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// read_exr (src/texture.cc:70-163): up to four channels, interleaved; R, G, B, A picked by the first letter of the channel name,
// or - as soon as one channel is called something else - the channels in file order.  (Files whose names break the R/G/B/A
// pattern only partly make the reference read through an unset index; those are refused here.)
inline std::vector<float> interleave_like_read_exr(const image& img, int& channel_count)
{
    const int nc = (int)img.channels.size();
    channel_count = std::min(nc, 4);
    int cid[4] = {-1, -1, -1, -1};
    bool rgba = true;
    for(int c = 0; c < nc; ++c)
    {
        const char k = img.channels[(size_t)c].name.empty() ? '\0' : img.channels[(size_t)c].name[0];
        if(k == 'R') cid[0] = c;
        else if(k == 'G') cid[1] = c;
        else if(k == 'B') cid[2] = c;
        else if(k == 'A') cid[3] = c;
        else rgba = false;
    }
    if(!rgba) for(int c = 0; c < std::min(nc, 4); ++c) cid[c] = c;
    for(int c = 0; c < channel_count; ++c)
        if(cid[c] < 0) throw std::runtime_error("EXR: channel names are not R, G, B, A in order (the reference reads an unset index for these)");
    const size_t px = (size_t)img.width * img.height;
    std::vector<float> out(px * (size_t)channel_count);
    for(int c = 0; c < channel_count; ++c)
    {
        const float* src = img.planes[(size_t)cid[c]].data();
        for(size_t i = 0; i < px; ++i) out[i * (size_t)channel_count + (size_t)c] = src[i];
    }
    return out;
}

inline std::vector<float> load_exr(const std::string& path, uint32_t& width, uint32_t& height, int& channel_count)
{
    const std::vector<uint8_t> bytes = read_file(path);
    const image img = read(bytes.data(), bytes.size());
    width = (uint32_t)img.width; height = (uint32_t)img.height;
    return interleave_like_read_exr(img, channel_count);
}

// The file headless::save_image writes (src/headless.cc:355-412): RGBA32F pixels in, channels [A,] B, G, R out.
inline std::vector<uint8_t> encode(const float* rgba, uint32_t width, uint32_t height, bool alpha, bool half, int compression)
{
    const int lines = lines_per_block(compression);
    if(lines == 0) throw std::runtime_error("EXR: unsupported compression " + std::to_string(compression));
    const int nch = alpha ? 4 : 3;
    const char* names[4] = {"A", "B", "G", "R"};          // alphabetical = file order
    const int src[4] = {3, 2, 1, 0};
    const int first = alpha ? 0 : 1;
    std::vector<uint8_t> out;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; out.insert(out.end(), b, b + n); };
    auto put_i32 = [&](int32_t v) { put(&v, 4); };
    auto put_str = [&](const char* s) { put(s, std::strlen(s) + 1); };
    auto attr = [&](const char* name, const char* type, int32_t size) { put_str(name); put_str(type); put_i32(size); };
    put_i32(20000630); put_i32(2);
    attr("channels", "chlist", nch * (2 + 16) + 1);
    for(int c = first; c < 4; ++c)
    {
        put_str(names[c]);
        put_i32(half ? PIXEL_HALF : PIXEL_FLOAT); uint8_t plinear[4] = {0, 0, 0, 0}; put(plinear, 4); put_i32(1); put_i32(1);
    }
    uint8_t zero = 0; put(&zero, 1);
    const uint8_t code = (uint8_t)compression;
    attr("compression", "compression", 1); put(&code, 1);
    int32_t box[4] = {0, 0, (int32_t)width - 1, (int32_t)height - 1};
    attr("dataWindow", "box2i", 16); put(box, 16);
    attr("displayWindow", "box2i", 16); put(box, 16);
    attr("lineOrder", "lineOrder", 1); put(&zero, 1);
    float one = 1.0f, origin[2] = {0, 0};
    attr("pixelAspectRatio", "float", 4); put(&one, 4);
    attr("screenWindowCenter", "v2f", 8); put(origin, 8);
    attr("screenWindowWidth", "float", 4); put(&one, 4);
    put(&zero, 1);
    const uint32_t n_blocks = (height + (uint32_t)lines - 1) / (uint32_t)lines;
    const size_t table_pos = out.size();
    out.resize(out.size() + 8 * size_t(n_blocks));
    const std::vector<int> wpp((size_t)nch, half ? 1 : 2);
    std::vector<uint8_t> raw, packed;
    for(uint32_t b = 0; b < n_blocks; ++b)
    {
        const uint32_t y0 = b * (uint32_t)lines, y1 = std::min(height, y0 + (uint32_t)lines);
        raw.clear();
        raw.reserve((size_t)(y1 - y0) * width * (size_t)nch * (half ? 2 : 4));
        for(uint32_t y = y0; y < y1; ++y)
            for(int c = first; c < 4; ++c)
                for(uint32_t x = 0; x < width; ++x)
                {
                    float v = rgba[(size_t(y) * width + x) * 4 + (size_t)src[c]];
                    if(half) { uint16_t h = float_to_half(v); raw.insert(raw.end(), (uint8_t*)&h, (uint8_t*)&h + 2); }
                    else raw.insert(raw.end(), (uint8_t*)&v, (uint8_t*)&v + 4);
                }
        compress_block(compression, raw, (int)width, (int)(y1 - y0), wpp, packed);
        uint64_t off = out.size();
        std::memcpy(out.data() + table_pos + 8 * size_t(b), &off, 8);
        put_i32((int32_t)y0); put_i32((int32_t)packed.size());
        put(packed.data(), packed.size());
    }
    return out;
}

}
}
#endif
