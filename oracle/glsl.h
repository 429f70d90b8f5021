// TEST INFRASTRUCTURE - not part of the shipped product.
// Minimal GLSL-semantics vector/matrix layer for the CPU oracle: only what the
// restated shaders use.  Column-major matrices, GLSL built-in semantics
// (mix, step, reflect, refract, bitfieldReverse, findLSB/findMSB, ...).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace gl {

typedef uint32_t uint;

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct uvec2 { uint x, y; };
struct uvec3 { uint x, y, z; };
struct uvec4 { uint x, y, z, w; };
struct ivec2 { int x, y; };
struct ivec3 { int x, y, z; };

inline vec2 V2(float a) { return {a, a}; }
inline vec2 V2(float a, float b) { return {a, b}; }
inline vec3 V3(float a) { return {a, a, a}; }
inline vec3 V3(float a, float b, float c) { return {a, b, c}; }
inline vec3 V3(const vec4& v) { return {v.x, v.y, v.z}; }
inline vec4 V4(float a) { return {a, a, a, a}; }
inline vec4 V4(float a, float b, float c, float d) { return {a, b, c, d}; }
inline vec4 V4(const vec3& v, float w) { return {v.x, v.y, v.z, w}; }

// --- vec2
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator/(vec2 a, vec2 b) { return {a.x / b.x, a.y / b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(float s, vec2 a) { return {a.x * s, a.y * s}; }
inline vec2 operator/(vec2 a, float s) { return {a.x / s, a.y / s}; }
inline vec2 operator+(vec2 a, float s) { return {a.x + s, a.y + s}; }
inline vec2 operator-(vec2 a, float s) { return {a.x - s, a.y - s}; }
inline vec2 operator-(float s, vec2 a) { return {s - a.x, s - a.y}; }
inline vec2 operator-(vec2 a) { return {-a.x, -a.y}; }
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }

// --- vec3
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(vec3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline float length(vec2 a) { return sqrtf(dot(a, a)); }
// GLSL normalize(0) is undefined; drivers yield NaN (x * inversesqrt(0)).
inline vec3 normalize(vec3 a) { float l = length(a); return {a.x / l, a.y / l, a.z / l}; }
inline float idx(const vec3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// --- vec4
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(float s, vec4 a) { return a * s; }
inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
inline vec4& operator+=(vec4& a, vec4 b) { a = a + b; return a; }
inline float dot(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// --- uvec
inline uvec4 operator+(uvec4 a, uvec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline uvec4 operator*(uvec4 a, uvec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline uvec4 operator*(uvec4 a, uint s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline uvec4 operator+(uvec4 a, uint s) { return {a.x + s, a.y + s, a.z + s, a.w + s}; }
inline uvec4 operator^(uvec4 a, uvec4 b) { return {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
inline uvec4 operator|(uvec4 a, uint s) { return {a.x | s, a.y | s, a.z | s, a.w | s}; }
inline uvec4 operator>>(uvec4 a, uint s) { return {a.x >> s, a.y >> s, a.z >> s, a.w >> s}; }
inline uvec2 operator+(uvec2 a, uvec2 b) { return {a.x + b.x, a.y + b.y}; }
inline uvec2 operator*(uvec2 a, uint s) { return {a.x * s, a.y * s}; }
inline uvec2 operator+(uvec2 a, uint s) { return {a.x + s, a.y + s}; }
inline uvec2 operator^(uvec2 a, uvec2 b) { return {a.x ^ b.x, a.y ^ b.y}; }
inline uvec2 operator>>(uvec2 a, uint s) { return {a.x >> s, a.y >> s}; }
inline vec4 to_float(uvec4 a) { return {(float)a.x, (float)a.y, (float)a.z, (float)a.w}; }

// --- matrices (column-major)
struct mat3 { vec3 c[3]; };
struct mat4 { vec4 c[4]; };
inline mat3 M3(vec3 a, vec3 b, vec3 c) { return {{a, b, c}}; }
inline mat3 M3(const mat4& m) { return {{V3(m.c[0]), V3(m.c[1]), V3(m.c[2])}}; }
inline vec3 operator*(const mat3& m, vec3 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z; }
inline vec3 operator*(vec3 v, const mat3& m) { return {dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])}; }
inline vec4 operator*(const mat4& m, vec4 v) { return m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3] * v.w; }
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = a * b.c[i];
    return r;
}

// --- scalar built-ins
inline float min(float a, float b) { return b < a ? b : a; }
inline float max(float a, float b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
inline int clamp(int x, int lo, int hi) { return min(max(x, lo), hi); }
inline uint clamp(uint x, uint lo, uint hi) { return min(max(x, lo), hi); }
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
inline vec4 mix(vec4 a, vec4 b, float t) { return a * (1.0f - t) + b * t; }
inline float step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
inline float sign(float x) { return x > 0 ? 1.0f : (x < 0 ? -1.0f : 0.0f); }
inline float fract(float x) { return x - floorf(x); }
inline float inversesqrt(float x) { return 1.0f / sqrtf(x); }
inline vec3 max(vec3 a, vec3 b) { return {max(a.x, b.x), max(a.y, b.y), max(a.z, b.z)}; }
inline vec3 min(vec3 a, vec3 b) { return {min(a.x, b.x), min(a.y, b.y), min(a.z, b.z)}; }
inline vec3 clamp(vec3 a, vec3 lo, vec3 hi) { return min(max(a, lo), hi); }
inline vec3 abs3(vec3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
inline vec3 pow3(vec3 a, vec3 b) { return {powf(a.x, b.x), powf(a.y, b.y), powf(a.z, b.z)}; }
inline bool any_nan(vec3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}
inline uint bitfieldReverse(uint v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
inline uvec4 bitfieldReverse(uvec4 v) {
    return {bitfieldReverse(v.x), bitfieldReverse(v.y), bitfieldReverse(v.z), bitfieldReverse(v.w)};
}
inline int findLSB(uint v) { return v == 0 ? -1 : __builtin_ctz(v); }
inline int findMSB(uint v) { return v == 0 ? -1 : 31 - __builtin_clz(v); }

// packHalf2x16 / unpackHalf2x16 (round-to-nearest-even float -> half)
inline uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t man = x & 0x7FFFFFu;
    int32_t exp = (int32_t)((x >> 23) & 0xFF);
    if (exp == 255) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    exp = exp - 127 + 15;
    if (exp >= 31) return (uint16_t)(sign | 0x7C00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        int shift = 14 - exp;
        uint32_t h = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}
inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FFu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7F800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}
inline uint packHalf2x16(vec2 v) { return (uint)float_to_half(v.x) | ((uint)float_to_half(v.y) << 16); }
inline vec2 unpackHalf2x16(uint p) { return {half_to_float((uint16_t)(p & 0xFFFF)), half_to_float((uint16_t)(p >> 16))}; }

}  // namespace gl
