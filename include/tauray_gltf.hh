// tauray_gltf.hh - GLB (binary glTF 2.0) loader for the C++ host layer: file -> tr::scene_data, the flattened scene
// trhip_scene_upload takes (SURVEY.md Appendix A arrays).
//
// What the reference's loader does for the subset the path tracer needs: create_material (src/gltf.cc:199-280), nodes / lights /
// cameras (:330-505), meshes (:510-798), mesh::calculate_normals / calculate_tangents (src/mesh.cc:113-185), followed by the
// instance flattening of scene_stage (src/scene_stage.cc:664-819: one instance per (model, vertex group) in node traversal
// order) and the host-side packing of src/camera.cc:323-478, src/light.cc and src/scene_stage.cc:17-111,1066-1354.
// Extensions: KHR_lights_punctual, KHR_materials_transmission, KHR_materials_ior, KHR_materials_emissive_strength, TR_data.
// Skins: JOINTS_0 / WEIGHTS_0 (renormalised) + inverse bind matrices; skinned meshes sit at the origin (src/gltf.cc:722-731,
// 777-784) and scene_data::skinned carries the file's rest pose for scene_stage::set_scene.  Animation clips (translation / rotation /
// scale channels, LINEAR / STEP / CUBICSPLINE; src/gltf.cc:167-190,580-627) are kept in scene_data::animation and played by
// tr::scene_animator below (src/animation.{hh,cc,tcc}, src/scene.cc:213-244).  Textures: PNG of any colour type, bit depth and
// interlacing and baseline JPEG (include/tauray_image.hh), embedded or behind a relative / data: uri (src/gltf.cc:532-576); besides
// .glb containers - the only form the reference opens - .gltf text files with their buffers.  Not read: morph targets,
// progressive JPEG.  Same scope and same results as the Python mirror tauray_amd/gltf.py
// (tests/test_cpp_host.py::test_cpp_glb_loader_matches_python_loader).
#ifndef TAURAY_GLTF_HH
#define TAURAY_GLTF_HH
#include <array>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>

#include "tauray_hip.hh"
#include "tauray_image.hh"

namespace tr
{
namespace gltf_detail
{

//---------------------------------------------------------------------------------------------------------------------
// JSON (the subset glTF uses; numbers as double)
struct json
{
    enum kind_t { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<json> arr;
    std::vector<std::pair<std::string, json>> obj;

    const json* find(const std::string& key) const
    {
        if(kind != OBJ) return nullptr;
        for(const auto& kv: obj) if(kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string& key) const { return find(key) != nullptr; }
    const json& at(const std::string& key) const
    {
        const json* j = find(key);
        if(!j) throw std::runtime_error("glTF: missing key " + key);
        return *j;
    }
    const json& at(size_t i) const
    {
        if(kind != ARR || i >= arr.size()) throw std::runtime_error("glTF: index out of range");
        return arr[i];
    }
    double number(const std::string& key, double fallback) const { const json* j = find(key); return j && j->kind == NUM ? j->num : fallback; }
    int integer(const std::string& key, int fallback) const { const json* j = find(key); return j && j->kind == NUM ? (int)j->num : fallback; }
    size_t size() const { return kind == ARR ? arr.size() : 0; }
};

class json_parser
{
public:
    json_parser(const char* p, size_t n): p(p), end(p + n) {}
    json parse() { json v = value(); ws(); return v; }
private:
    const char *p, *end;
    int depth = 0;      // nesting of the value being parsed: a file of nothing but '[' must end in a message, not in the end of the stack
    struct nest { int& d; explicit nest(int& d): d(d) { ++d; } ~nest() { --d; } };
    [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("glTF JSON: ") + what); }
    void ws() { while(p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
    json value()
    {
        ws();
        if(p >= end) fail("unexpected end");
        const nest level(depth);
        if(depth > 200) fail("nested too deeply");
        json v;
        if(*p == '{')
        {
            v.kind = json::OBJ; ++p; ws();
            if(p < end && *p == '}') { ++p; return v; }
            while(true)
            {
                ws();
                std::string k = string();
                ws();
                if(p >= end || *p != ':') fail("expected ':'");
                ++p;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if(p < end && *p == ',') { ++p; continue; }
                if(p < end && *p == '}') { ++p; return v; }
                fail("expected ',' or '}'");
            }
        }
        if(*p == '[')
        {
            v.kind = json::ARR; ++p; ws();
            if(p < end && *p == ']') { ++p; return v; }
            while(true)
            {
                v.arr.push_back(value());
                ws();
                if(p < end && *p == ',') { ++p; continue; }
                if(p < end && *p == ']') { ++p; return v; }
                fail("expected ',' or ']'");
            }
        }
        if(*p == '"') { v.kind = json::STR; v.str = string(); return v; }
        if(end - p >= 4 && !std::strncmp(p, "true", 4)) { p += 4; v.kind = json::BOOL; v.b = true; return v; }
        if(end - p >= 5 && !std::strncmp(p, "false", 5)) { p += 5; v.kind = json::BOOL; return v; }
        if(end - p >= 4 && !std::strncmp(p, "null", 4)) { p += 4; return v; }
        // number: strtod parses what Python's float() parses (correctly rounded decimal -> double)
        std::string tmp;
        while(p < end && (std::isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) tmp.push_back(*p++);
        if(tmp.empty()) fail("unexpected character");
        v.kind = json::NUM;
        v.num = std::strtod(tmp.c_str(), nullptr);
        return v;
    }
    std::string string()
    {
        if(p >= end || *p != '"') fail("expected string");
        ++p;
        std::string s;
        while(p < end && *p != '"')
        {
            if(*p == '\\')
            {
                if(++p >= end) fail("bad escape");
                switch(*p)
                {
                case 'n': s.push_back('\n'); break;
                case 't': s.push_back('\t'); break;
                case 'r': s.push_back('\r'); break;
                case 'b': s.push_back('\b'); break;
                case 'f': s.push_back('\f'); break;
                case 'u': if(end - p < 5) fail("bad \\u escape"); s.push_back('?'); p += 4; break;   // names only: not needed exactly
                default: s.push_back(*p);
                }
                ++p;
            }
            else s.push_back(*p++);
        }
        if(p >= end) fail("unterminated string");
        ++p;
        return s;
    }
};

//---------------------------------------------------------------------------------------------------------------------
// 4x4 double matrices, mathematical (row, column) indexing; stored to the GPU structs column-major like glm
struct mat4d
{
    double m[4][4];
    static mat4d identity() { mat4d r{}; for(int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
};
inline mat4d mul(const mat4d& a, const mat4d& b)
{
    mat4d r{};
    for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j)
    {
        double s = 0;
        for(int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
        r.m[i][j] = s;
    }
    return r;
}
inline mat4d transpose(const mat4d& a) { mat4d r{}; for(int i = 0; i < 4; ++i) for(int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
// LU with partial pivoting, then the inverse column by column (what LAPACK's getrf / getri amount to for a 4x4)
inline mat4d inverse(const mat4d& a)
{
    double lu[4][4];
    int piv[4] = {0, 1, 2, 3};
    std::memcpy(lu, a.m, sizeof(lu));
    for(int k = 0; k < 4; ++k)
    {
        int best = k;
        for(int i = k + 1; i < 4; ++i) if(std::fabs(lu[i][k]) > std::fabs(lu[best][k])) best = i;
        if(best != k) { for(int j = 0; j < 4; ++j) std::swap(lu[k][j], lu[best][j]); std::swap(piv[k], piv[best]); }
        for(int i = k + 1; i < 4; ++i)
        {
            lu[i][k] /= lu[k][k];
            for(int j = k + 1; j < 4; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
        }
    }
    mat4d r{};
    for(int c = 0; c < 4; ++c)
    {
        double y[4];
        for(int i = 0; i < 4; ++i)
        {
            double s = piv[i] == c ? 1.0 : 0.0;
            for(int j = 0; j < i; ++j) s -= lu[i][j] * y[j];
            y[i] = s;
        }
        for(int i = 3; i >= 0; --i)
        {
            double s = y[i];
            for(int j = i + 1; j < 4; ++j) s -= lu[i][j] * r.m[j][c];
            r.m[i][c] = s / lu[i][i];
        }
    }
    return r;
}
inline void to_glm(const mat4d& a, float* out16) { for(int c = 0; c < 4; ++c) for(int r = 0; r < 4; ++r) out16[c * 4 + r] = (float)a.m[r][c]; }

// transformable::get_transform (src/transformable.cc:203-212)
inline mat4d trs_matrix(const double t[3], const double q[4], const double s[3])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double rot[3][3] = {
        {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
        {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
        {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    mat4d m = mat4d::identity();
    for(int i = 0; i < 3; ++i) for(int j = 0; j < 3; ++j) m.m[i][j] = rot[i][j] * s[j];
    for(int i = 0; i < 3; ++i) m.m[i][3] = t[i];
    return m;
}

//---------------------------------------------------------------------------------------------------------------------
// GPU-side PODs
#pragma pack(push, 4)
struct vertex { float pos[3], normal[3], uv[2], tangent[4]; };
struct material { float albedo_factor[4], metallic_roughness_factor[4], emission_factor[4], transmittance, ior, normal_factor; uint32_t flags; int32_t albedo_tex, mr_tex, normal_tex, emission_tex; };
struct instance { int32_t light_base_id, sh_grid_index; uint32_t pad; float shadow_terminator_mul; float model[16], model_normal[16], model_prev[16]; material mat; };
struct directional_light { float color[3]; int32_t shadow_map_index; float dir[3]; float dir_cutoff; };
struct point_light { float color[3], dir[3], pos[3], radius, dir_cutoff, dir_falloff, cutoff_radius, spot_radius; int32_t shadow_map_index, padding; };
struct camera_data { float view[16], view_inverse[16], view_proj[16], proj_inverse[16], origin[4], dof_params[4], projection_info[4], pan[4]; };

//---------------------------------------------------------------------------------------------------------------------
// A camera as the file describes it + camera::write_uniform_buffer (src/camera.cc:431-478) with the aspect ratio
// set_camera_params forces (src/tauray.cc:68-110)
struct gltf_camera { mat4d transform; bool perspective; double fov, aspect, near, far; double ortho[6]; double pan[2] = {0, 0}; };

inline camera_data pack_camera(gltf_camera& c, double aspect)
{
    camera_data cd{};
    mat4d proj{};
    double info[4];
    if(c.perspective)
    {
        c.aspect = aspect;
        const double t = std::tan((c.fov * (3.14159265358979323846 / 180.0)) / 2.0);
        proj.m[0][0] = 1.0 / (c.aspect * t); proj.m[1][1] = 1.0 / t; proj.m[3][2] = -1.0;
        const double w = 2 * std::tan((c.fov * (3.14159265358979323846 / 180.0)) / 2.0), z = w * c.aspect;
        if(std::isinf(c.far)) { proj.m[2][2] = -1.0; proj.m[2][3] = -2.0 * c.near; info[0] = -c.near; info[1] = -1.0; }
        else
        {
            proj.m[2][2] = -(c.far + c.near) / (c.far - c.near); proj.m[2][3] = -(2.0 * c.far * c.near) / (c.far - c.near);
            info[0] = c.near * c.far / (c.near - c.far); info[1] = (c.near + c.far) / (c.near - c.far);
        }
        info[2] = z; info[3] = w;
        cd.dof_params[0] = 1.0f;
        proj.m[0][2] = c.pan[0]; proj.m[1][2] = c.pan[1];      // camera::set_pan (src/camera.cc:375-381): the off-axis shift of a light-field view
        cd.pan[0] = (float)c.pan[0]; cd.pan[1] = (float)c.pan[1];
    }
    else
    {
        double l = c.ortho[0], r = c.ortho[1], b = c.ortho[2], t = c.ortho[3];
        const double n = c.ortho[4], f = c.ortho[5];
        const double yr = (r - l) / aspect, yc = (b + t) * 0.5;      // camera::set_aspect (src/camera.cc:166-186)
        b = yc - yr * 0.5; t = yc + yr * 0.5;
        proj = mat4d::identity();
        proj.m[0][0] = 2 / (r - l); proj.m[1][1] = 2 / (t - b); proj.m[2][2] = -2 / (f - n);
        proj.m[0][3] = -(r + l) / (r - l); proj.m[1][3] = -(t + b) / (t - b); proj.m[2][3] = -(f + n) / (f - n);
        info[0] = f - n; info[1] = -f; info[2] = r - l; info[3] = t - b;
    }
    const mat4d view = inverse(c.transform);
    to_glm(view, cd.view);
    to_glm(c.transform, cd.view_inverse);
    to_glm(mul(proj, view), cd.view_proj);
    to_glm(inverse(proj), cd.proj_inverse);
    for(int k = 0; k < 4; ++k) cd.origin[k] = (float)c.transform.m[k][3];
    for(int k = 0; k < 4; ++k) cd.projection_info[k] = (float)info[k];
    return cd;
}

struct mesh_span { uint32_t vertex_offset, vertex_count, index_offset, triangle_count; };
struct texture_info { uint32_t width, height, texel_offset, format; };      // offset in 4-byte words; format 0 = RGBA8, 1 = RGBA16 (include/trhip.h)
#pragma pack(pop)
static_assert(sizeof(vertex) == 48 && sizeof(material) == 80 && sizeof(instance) == 288 && sizeof(camera_data) == 320, "layout");
static_assert(sizeof(directional_light) == 32 && sizeof(point_light) == 64, "layout");

// a glTF `uri`: a data: URI (base64) or a file next to the scene file (percent-encoded relative path)
inline std::vector<uint8_t> read_uri(const std::string& uri, const std::string& base_dir)
{
    auto unquote = [](const std::string& u) {
        std::string out;
        for(size_t i = 0; i < u.size(); ++i)
        {
            if(u[i] == '%' && i + 2 < u.size() && std::isxdigit((unsigned char)u[i + 1]) && std::isxdigit((unsigned char)u[i + 2]))
            { out.push_back((char)std::stoi(u.substr(i + 1, 2), nullptr, 16)); i += 2; }
            else out.push_back(u[i]);
        }
        return out;
    };
    if(uri.compare(0, 5, "data:") == 0)
    {
        const size_t comma = uri.find(',');
        if(comma == std::string::npos) throw std::runtime_error("glTF: malformed data: uri");
        const std::string head = uri.substr(0, comma);
        if(head.size() < 7 || head.compare(head.size() - 7, 7, ";base64") != 0) { const std::string t = unquote(uri.substr(comma + 1)); return std::vector<uint8_t>(t.begin(), t.end()); }
        std::vector<uint8_t> out;
        uint32_t acc = 0; int bits = 0;
        for(size_t i = comma + 1; i < uri.size(); ++i)
        {
            const char c = uri[i];
            int v;
            if(c >= 'A' && c <= 'Z') v = c - 'A'; else if(c >= 'a' && c <= 'z') v = c - 'a' + 26; else if(c >= '0' && c <= '9') v = c - '0' + 52;
            else if(c == '+' || c == '-') v = 62; else if(c == '/' || c == '_') v = 63; else continue;      // '=' padding and whitespace
            acc = (acc << 6) | (uint32_t)v; bits += 6;
            if(bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
        }
        return out;
    }
    const std::string path = (base_dir.empty() ? std::string() : base_dir + "/") + unquote(uri);
    std::ifstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("Failed to open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// The JSON document and the buffers of a .glb container (what the reference opens: LoadBinaryFromFile, src/gltf.cc:527) or of a
// .gltf text file with its external / data: buffers.
struct glb_file
{
    json doc;
    std::vector<std::vector<uint8_t>> buffers;
    std::string dir;

    explicit glb_file(const std::string& path)
    {
        std::ifstream f(path, std::ios::binary);
        if(!f) throw std::runtime_error("Failed to open " + path);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const size_t slash = path.find_last_of('/');
        dir = slash == std::string::npos ? std::string() : path.substr(0, slash);
        std::vector<uint8_t> glb_bin;
        bool have_bin = false;
        if(d.size() >= 12 && std::memcmp(d.data(), "glTF", 4) == 0)
        {
            size_t off = 12;
            bool have_json = false;
            while(off + 8 <= d.size())
            {
                uint32_t clen, ctype;
                std::memcpy(&clen, d.data() + off, 4); std::memcpy(&ctype, d.data() + off + 4, 4);
                if(off + 8 + clen > d.size()) throw std::runtime_error(path + " is truncated");
                if(ctype == 0x4E4F534A) { doc = json_parser(reinterpret_cast<const char*>(d.data() + off + 8), clen).parse(); have_json = true; }
                else if(ctype == 0x004E4942) { glb_bin.assign(d.begin() + off + 8, d.begin() + off + 8 + clen); have_bin = true; }
                off += 8 + size_t(clen);
            }
            if(!have_json) throw std::runtime_error(path + " has no JSON chunk");
        }
        else
        {
            size_t i = 0;
            while(i < d.size() && std::isspace(d[i])) ++i;
            if(i >= d.size() || d[i] != '{') throw std::runtime_error(path + " is not a GLB file");
            doc = json_parser(reinterpret_cast<const char*>(d.data()), d.size()).parse();
        }
        if(doc.has("buffers"))
        {
            size_t i = 0;
            for(const json& b: doc.at("buffers").arr)
            {
                if(b.has("uri")) buffers.push_back(read_uri(b.at("uri").str, dir));
                else if(i == 0 && have_bin) buffers.push_back(std::move(glb_bin));
                else throw std::runtime_error("glTF: buffer " + std::to_string(i) + " has neither a uri nor a GLB chunk");
                ++i;
            }
        }
    }

    const std::vector<uint8_t>& buffer_of(const json& buffer_view) const
    {
        const size_t b = (size_t)buffer_view.number("buffer", 0);
        if(b >= buffers.size()) throw std::runtime_error("glTF: bufferView names a missing buffer");
        return buffers[b];
    }
    // the file behind an `images` entry: embedded (bufferView) or a uri (src/gltf.cc:532-576)
    std::vector<uint8_t> image(const json& img) const
    {
        if(img.has("bufferView"))
        {
            const json& bv = doc.at("bufferViews").at((size_t)img.integer("bufferView", 0));
            const std::vector<uint8_t>& bin = buffer_of(bv);
            const size_t o = (size_t)bv.number("byteOffset", 0), n = (size_t)bv.number("byteLength", 0);
            if(o + n > bin.size()) throw std::runtime_error("glTF: image exceeds the buffer");
            return std::vector<uint8_t>(bin.begin() + (long)o, bin.begin() + (long)(o + n));
        }
        if(img.has("uri")) return read_uri(img.at("uri").str, dir);
        throw std::runtime_error("glTF: image without bufferView or uri");
    }

    // accessor as rows of doubles (exact for every component type glTF has)
    std::vector<double> accessor(int index, int& components, size_t& count) const
    {
        if(index < 0 || (size_t)index >= doc.at("accessors").size()) throw std::runtime_error("glTF: accessor index out of range");
        const json& a = doc.at("accessors").at((size_t)index);
        const int view = a.integer("bufferView", -1);
        if(view < 0 || (size_t)view >= doc.at("bufferViews").size()) throw std::runtime_error("glTF: bufferView index out of range");
        const json& bv = doc.at("bufferViews").at((size_t)view);
        const std::vector<uint8_t>& bin = buffer_of(bv);
        const int ct = a.integer("componentType", 0);
        const std::string& type = a.at("type").str;
        components = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
        const int sz = ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : ct == 5125 || ct == 5126 ? 4 : 0;
        if(!components || !sz) throw std::runtime_error("glTF: unsupported accessor type");
        // every number below comes from the file: checked as doubles before any of them becomes a size_t (a negative or huge double
        // cast to size_t is undefined), and the extent of the accessor is compared without a product that could wrap
        const double d_count = a.number("count", 0), d_vo = bv.number("byteOffset", 0), d_ao = a.number("byteOffset", 0),
                     d_len = bv.number("byteLength", 0), d_stride = bv.number("byteStride", 0);
        const double lim = (double)bin.size();
        if(!(d_count >= 0 && d_vo >= 0 && d_ao >= 0 && d_len >= 0 && d_stride >= 0)) throw std::runtime_error("glTF: negative count, offset or stride");
        if(d_count > lim || d_vo > lim || d_ao > lim || d_len > lim) throw std::runtime_error("glTF: accessor exceeds its buffer");
        const size_t elem = size_t(sz) * (size_t)components;
        // byteStride: 4 ... 252 (glTF 2.0 section 5.11) and at least one element, 0 = tightly packed
        if(d_stride != 0 && (d_stride < (double)elem || d_stride > 252)) throw std::runtime_error("glTF: byteStride out of range");
        count = (size_t)d_count;
        const size_t base = (size_t)d_vo + (size_t)d_ao;
        const size_t stride = d_stride != 0 ? (size_t)d_stride : elem;
        // an accessor lives inside its bufferView, a bufferView inside its buffer (glTF 2.0 section 3.6.2)
        const size_t view_len = (size_t)d_len, in_view = (size_t)d_ao;
        if((size_t)d_vo > bin.size() || view_len > bin.size() - (size_t)d_vo) throw std::runtime_error("glTF: bufferView exceeds the buffer");
        if(count && (in_view > view_len || view_len - in_view < elem || count - 1 > (view_len - in_view - elem) / stride))
            throw std::runtime_error("glTF: accessor exceeds its bufferView");
        std::vector<double> out(count * components);
        for(size_t i = 0; i < count; ++i)
            for(int c = 0; c < components; ++c)
            {
                const uint8_t* p = bin.data() + base + i * stride + size_t(c) * sz;
                double v = 0;
                switch(ct)
                {
                case 5120: v = *reinterpret_cast<const int8_t*>(p); break;
                case 5121: v = *p; break;
                case 5122: { int16_t t; std::memcpy(&t, p, 2); v = t; break; }
                case 5123: { uint16_t t; std::memcpy(&t, p, 2); v = t; break; }
                case 5125: { uint32_t t; std::memcpy(&t, p, 4); v = t; break; }
                default: { float t; std::memcpy(&t, p, 4); v = t; break; }
                }
                out[i * components + c] = v;
            }
        return out;
    }
};

// create_material (src/gltf.cc:199-280)
inline material create_material(const glb_file& g, const json& mat)
{
    static const json empty;
    const json& pbr = mat.has("pbrMetallicRoughness") ? mat.at("pbrMetallicRoughness") : empty;
    auto tex_source = [&](const json* info) -> int32_t {
        if(!info || info->integer("index", -1) < 0) return -1;
        return g.doc.at("textures").at((size_t)info->integer("index", 0)).integer("source", -1);
    };
    double albedo[4] = {1, 1, 1, 1}, emission[3] = {0, 0, 0};
    if(const json* f = pbr.find("baseColorFactor")) for(size_t i = 0; i < 4 && i < f->size(); ++i) albedo[i] = f->at(i).num;
    if(const json* f = mat.find("emissiveFactor")) for(size_t i = 0; i < 3 && i < f->size(); ++i) emission[i] = f->at(i).num;
    double transmittance = 0.0, ior = 1.45;
    const json& ext = mat.has("extensions") ? mat.at("extensions") : empty;
    bool discard_tr_emission = false;
    if(const json* es = ext.find("KHR_materials_emissive_strength")) if(es->has("emissiveStrength"))
    {
        const double k = es->at("emissiveStrength").num;
        for(double& e: emission) e = e * k;
        discard_tr_emission = true;
    }
    const json* pbr_ext = pbr.find("extensions");
    if(const json* tr = pbr_ext ? pbr_ext->find("TR_data") : nullptr)
    {
        if(tr->has("transmission")) transmittance = tr->at("transmission").num;
        if(tr->has("ior")) ior = tr->at("ior").num;
        if(!discard_tr_emission && tr->has("emission")) for(size_t i = 0; i < 3; ++i) emission[i] = tr->at("emission").at(i).num;
    }
    if(const json* t = ext.find("KHR_materials_transmission")) if(t->has("transmissionFactor")) transmittance = t->at("transmissionFactor").num;
    if(const json* t = ext.find("KHR_materials_ior")) if(t->has("ior")) ior = t->at("ior").num;
    material m{};
    for(int i = 0; i < 4; ++i) m.albedo_factor[i] = (float)albedo[i];
    m.metallic_roughness_factor[0] = (float)pbr.number("metallicFactor", 1.0);
    m.metallic_roughness_factor[1] = (float)pbr.number("roughnessFactor", 1.0);
    for(int i = 0; i < 3; ++i) m.emission_factor[i] = (float)emission[i];
    m.transmittance = (float)transmittance; m.ior = (float)ior; m.normal_factor = 1.0f;
    const json* ds = mat.find("doubleSided");
    m.flags = ds && ds->kind == json::BOOL && ds->b ? 1u : 0u;
    m.albedo_tex = tex_source(pbr.find("baseColorTexture"));
    m.mr_tex = tex_source(pbr.find("metallicRoughnessTexture"));
    m.normal_tex = tex_source(mat.find("normalTexture"));
    m.emission_tex = tex_source(mat.find("emissiveTexture"));
    return m;
}

struct vertex_group { material mat; std::vector<vertex> vertices; std::vector<uint32_t> indices; std::vector<trhip_skin> skins; };

// mesh::calculate_normals (src/mesh.cc:113-143)
inline void calculate_normals(std::vector<vertex>& v, const std::vector<uint32_t>& idx)
{
    for(vertex& x: v) x.normal[0] = x.normal[1] = x.normal[2] = 0;
    for(size_t i = 0; i + 2 < idx.size(); i += 3)
    {
        vertex &v0 = v[idx[i]], &v1 = v[idx[i + 1]], &v2 = v[idx[i + 2]];
        const float d0[3] = {v1.pos[0] - v0.pos[0], v1.pos[1] - v0.pos[1], v1.pos[2] - v0.pos[2]};
        const float d1[3] = {v2.pos[0] - v0.pos[0], v2.pos[1] - v0.pos[1], v2.pos[2] - v0.pos[2]};
        float hn[3] = {d0[1] * d1[2] - d1[1] * d0[2], d0[2] * d1[0] - d1[2] * d0[0], d0[0] * d1[1] - d1[0] * d0[1]};
        const float len = std::sqrt(hn[0] * hn[0] + hn[1] * hn[1] + hn[2] * hn[2]);
        if(len > 1e-6f) for(float& c: hn) c /= len;
        for(vertex* p: {&v0, &v1, &v2}) for(int k = 0; k < 3; ++k) p->normal[k] += hn[k];
    }
    for(vertex& x: v)
    {
        const float len = std::sqrt(x.normal[0] * x.normal[0] + x.normal[1] * x.normal[1] + x.normal[2] * x.normal[2]);
        if(len > 1e-6f) for(float& c: x.normal) c /= len;
    }
}

// mesh::calculate_tangents (src/mesh.cc:145-185); only the first vertex of a triangle accumulates, as in the reference
inline void calculate_tangents(std::vector<vertex>& v, const std::vector<uint32_t>& idx)
{
    // glm::normalize: v * inversesqrt(dot(v, v)), the reciprocal square root as 1 / sqrt in float - not a division by the length
    auto normalize3 = [](float* a) { const float inv = 1.0f / std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); a[0] *= inv; a[1] *= inv; a[2] *= inv; };
    std::vector<std::array<float, 4>> acc(v.size(), std::array<float, 4>{0, 0, 0, 0});
    for(size_t i = 0; i + 2 < idx.size(); i += 3)
    {
        const vertex &v0 = v[idx[i]], &v1 = v[idx[i + 1]], &v2 = v[idx[i + 2]];
        const float d0[3] = {v1.pos[0] - v0.pos[0], v1.pos[1] - v0.pos[1], v1.pos[2] - v0.pos[2]};
        const float d1[3] = {v2.pos[0] - v0.pos[0], v2.pos[1] - v0.pos[1], v2.pos[2] - v0.pos[2]};
        float hn[3] = {d0[1] * d1[2] - d1[1] * d0[2], d0[2] * d1[0] - d1[2] * d0[0], d0[0] * d1[1] - d1[0] * d0[1]};
        const float len = std::sqrt(hn[0] * hn[0] + hn[1] * hn[1] + hn[2] * hn[2]);
        if(len > 1e-6f) for(float& c: hn) c /= len;
        const float uv0[2] = {v1.uv[0] - v0.uv[0], v1.uv[1] - v0.uv[1]}, uv1[2] = {v2.uv[0] - v0.uv[0], v2.uv[1] - v0.uv[1]};
        float ht[3], hb[3];
        for(int k = 0; k < 3; ++k) { ht[k] = uv1[1] * d0[k] - uv0[1] * d1[k]; hb[k] = uv1[0] * d1[k] - uv0[0] * d0[k]; }
        normalize3(ht); normalize3(hb);
        const float cr[3] = {hn[1] * ht[2] - ht[1] * hn[2], hn[2] * ht[0] - ht[2] * hn[0], hn[0] * ht[1] - ht[0] * hn[1]};
        const float sign = (cr[0] * hb[0] + cr[1] * hb[1] + cr[2] * hb[2]) < 0 ? -1.0f : 1.0f;
        std::array<float, 4>& a = acc[idx[i]];
        a[0] += ht[0]; a[1] += ht[1]; a[2] += ht[2]; a[3] += sign;
    }
    for(size_t i = 0; i < v.size(); ++i)
    {
        const float* n = v[i].normal;
        const float d = n[0] * acc[i][0] + n[1] * acc[i][1] + n[2] * acc[i][2];
        float t[3] = {acc[i][0] - n[0] * d, acc[i][1] - n[1] * d, acc[i][2] - n[2] * d};
        normalize3(t);
        v[i].tangent[0] = t[0]; v[i].tangent[1] = t[1]; v[i].tangent[2] = t[2];
        v[i].tangent[3] = acc[i][3] < 0 ? -1.0f : 1.0f;
    }
}

}   // namespace gltf_detail

//---------------------------------------------------------------------------------------------------------------------
// Animation clips (src/animation.{hh,cc,tcc}): what load_glb keeps of a file so that tr::scene_animator can play it.
// Mirrors tauray_amd/animation.py operation by operation (same rounding of timestamps, float ratio, double interpolation).
struct gltf_animation
{
    enum interpolation { LINEAR = 0, STEP, CUBICSPLINE };
    // std::vector<animation::sample<T>>: microsecond ticks (src/gltf.cc:167-190), three or four components per key
    struct track
    {
        interpolation interp = LINEAR;
        int width = 0;                                   // 3 (position, scaling) or 4 (orientation); 0 = no such track
        std::vector<int64_t> timestamps;
        std::vector<std::array<double, 4>> data, in_tangent, out_tangent;

        // animation::interpolate (src/animation.tcc:39-77)
        std::array<double, 4> sample(int64_t time, bool quaternion) const
        {
            const size_t i = (size_t)(std::upper_bound(timestamps.begin(), timestamps.end(), time) - timestamps.begin());
            if(i == timestamps.size()) return data.back();
            if(i == 0) return data.front();
            const float frame_ticks = (float)(timestamps[i] - timestamps[i - 1]);
            const double ratio = (double)((float)(time - timestamps[i - 1]) / frame_ticks);
            std::array<double, 4> r{};
            if(interp == STEP) return data[i - 1];
            if(interp == CUBICSPLINE && !in_tangent.empty())
            {   // cubic_spline (src/math.tcc:24-35): coefficients in float, tangents scaled by the interval in seconds
                const double scale = (double)(frame_ticks * 0.000001f);
                const float t = (float)ratio, t2 = t * t, t3 = t2 * t, tmp = 2.0f * t3 - 3.0f * t2;
                const double c1 = (double)(tmp + 1.0f), c2 = (double)(t3 - 2.0f * t2 + t), c3 = (double)(-tmp), c4 = (double)(t3 - t2);
                for(int k = 0; k < width; ++k)
                    r[(size_t)k] = c1 * data[i - 1][(size_t)k] + c2 * (out_tangent[i - 1][(size_t)k] * scale) + c3 * data[i][(size_t)k] + c4 * (in_tangent[i][(size_t)k] * scale);
                return r;
            }
            const std::array<double, 4>& a = data[i - 1];
            std::array<double, 4> b = data[i];
            if(quaternion)
            {   // glm::slerp: the shorter arc, linear when the two nearly coincide
                double cos_theta = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
                if(cos_theta < 0) { for(double& v: b) v = -v; cos_theta = -cos_theta; }
                if(!(cos_theta > 1.0 - (double)std::numeric_limits<float>::epsilon()))
                {
                    const double angle = std::acos(cos_theta), sa = std::sin((1.0 - ratio) * angle), sb = std::sin(ratio * angle), sn = std::sin(angle);
                    for(int k = 0; k < 4; ++k) r[(size_t)k] = (sa * a[(size_t)k] + sb * b[(size_t)k]) / sn;
                    return r;
                }
            }
            for(int k = 0; k < width; ++k) r[(size_t)k] = a[(size_t)k] * (1.0 - ratio) + b[(size_t)k] * ratio;
            return r;
        }
    };
    // tr::animation: one named clip of one node
    struct clip
    {
        track position, scaling, orientation;
        int64_t loop_time() const
        {
            int64_t t = 0;
            for(const track* tr: {&position, &scaling, &orientation}) if(tr->width && !tr->timestamps.empty()) t = std::max(t, tr->timestamps.back());
            return t;
        }
    };
    struct node
    {
        int parent = -1;
        std::vector<int> children;
        bool has_trs = true;
        double translation[3] = {0, 0, 0}, rotation[4] = {0, 0, 0, 1}, scale[3] = {1, 1, 1};
        gltf_detail::mat4d matrix = gltf_detail::mat4d::identity();
        std::vector<uint32_t> instances, cameras;       // rigid instances / cameras placed by this node's global transform
        // punctual lights on this node: kind, index in the scene's point (point lights, then spotlights) / directional array
        struct light { bool directional; uint32_t index; };
        std::vector<light> lights;
        gltf_detail::mat4d local() const { return has_trs ? gltf_detail::trs_matrix(translation, rotation, scale) : matrix; }
    };
    struct skin { std::vector<int> joint_nodes; std::vector<gltf_detail::mat4d> inverse_bind; };      // parallel to scene_data::skinned

    std::map<int, node> nodes;
    std::vector<int> roots;
    std::map<int, std::map<std::string, clip>> clips;     // node -> animation_pool (std::map: alphabetical)
    std::vector<gltf_detail::gltf_camera> cameras;
    std::vector<skin> skins;
    double aspect = 1.0;
};

struct glb_load_options
{
    float aspect_ratio = 0;               // 0: width / height (set_camera_params, src/tauray.cc:68-110)
    bool force_single_sided = false, force_double_sided = false;
    bool gather_emissive_triangles = true;   // scene_stage::options::gather_emissive_triangles (src/tauray.cc:384)
};

// load_gltf + scene flattening for a .glb file.
inline scene_data load_glb(const std::string& path, uint32_t width, uint32_t height, const glb_load_options& lo = {})
{
    using namespace gltf_detail;
    constexpr double PI = 3.14159265358979323846;
    const glb_file g(path);
    const json& j = g.doc;
    static const json empty;
    auto list = [&](const char* key) -> const json& { return j.has(key) ? j.at(key) : empty; };

    // The reference loads images flipped (stbi flag) and flips them back (src/gltf.cc:525,557): row 0 = top row of the file.
    // A PNG of 16 bits per sample stays RGBA16, as the reference keeps it (R16G16B16A16Unorm, src/gltf.cc:548-556).
    struct texture { uint32_t w, h; bool wide; std::vector<uint8_t> rgba; std::vector<uint16_t> rgba16; };
    std::vector<texture> textures;
    for(const json& img: list("images").arr)
    {
        const std::vector<uint8_t> file = g.image(img);      // PNG or JPEG by signature
        image::decoded d = image::decode(file.data(), file.size());
        texture t;
        t.w = d.w; t.h = d.h; t.wide = d.bits == 16; t.rgba = std::move(d.rgba); t.rgba16 = std::move(d.rgba16);
        textures.push_back(std::move(t));
    }

    // meshes -> vertex groups
    std::vector<std::vector<vertex_group>> models;
    for(const json& mesh: list("meshes").arr)
    {
        std::vector<vertex_group> groups;
        for(const json& p: mesh.at("primitives").arr)
        {
            vertex_group vg;
            if(p.has("material"))
            {
                const int mi = p.integer("material", -1);
                if(mi < 0 || (size_t)mi >= list("materials").size()) throw std::runtime_error("glTF: material index out of range");
                vg.mat = create_material(g, j.at("materials").at((size_t)mi));
                if(lo.force_single_sided && vg.mat.transmittance == 0) vg.mat.flags &= ~1u;
                if(lo.force_double_sided) vg.mat.flags |= 1u;
            }
            else
            {
                vg.mat = material{};
                for(float& c: vg.mat.albedo_factor) c = 1;
                vg.mat.metallic_roughness_factor[1] = 1; vg.mat.ior = 1.45f; vg.mat.normal_factor = 1;
                vg.mat.albedo_tex = vg.mat.mr_tex = vg.mat.normal_tex = vg.mat.emission_tex = -1;
            }
            const json& at = p.at("attributes");
            int nc; size_t count;
            const std::vector<double> pos = g.accessor(at.integer("POSITION", -1), nc, count);
            vg.vertices.assign(count, vertex{});
            for(size_t i = 0; i < count; ++i) for(int k = 0; k < 3; ++k) vg.vertices[i].pos[k] = (float)pos[i * nc + k];
            if(nc < 3) throw std::runtime_error("glTF: POSITION needs three components");
            auto attribute = [&](const char* name, int want, int& c) {      // an attribute has a value for every vertex and enough components
                size_t n;
                std::vector<double> a = g.accessor(at.integer(name, -1), c, n);
                if(n < count || c < want) throw std::runtime_error(std::string("glTF: attribute ") + name + " is shorter than POSITION");
                return a;
            };
            if(at.has("NORMAL")) { int c; auto a = attribute("NORMAL", 3, c); for(size_t i = 0; i < count; ++i) for(int k = 0; k < 3; ++k) vg.vertices[i].normal[k] = (float)a[i * c + k]; }
            if(at.has("TEXCOORD_0")) { int c; auto a = attribute("TEXCOORD_0", 2, c); for(size_t i = 0; i < count; ++i) for(int k = 0; k < 2; ++k) vg.vertices[i].uv[k] = (float)a[i * c + k]; }
            if(at.has("TANGENT")) { int c; auto a = attribute("TANGENT", 4, c); for(size_t i = 0; i < count; ++i) for(int k = 0; k < 4; ++k) vg.vertices[i].tangent[k] = (float)a[i * c + k]; }
            if(p.has("indices"))
            {   // unsigned integers (glTF 2.0 section 3.7.2.1)
                const int ia = p.integer("indices", -1);
                if(ia < 0 || (size_t)ia >= list("accessors").size()) throw std::runtime_error("glTF: accessor index out of range");
                const int ict = j.at("accessors").at((size_t)ia).integer("componentType", 0);
                if(ict != 5121 && ict != 5123 && ict != 5125) throw std::runtime_error("glTF: indices must be unsigned integers");
            }
            if(p.has("indices")) { int c; size_t n; auto a = g.accessor(p.integer("indices", 0), c, n); vg.indices.resize(n * c); for(size_t i = 0; i < a.size(); ++i) vg.indices[i] = (uint32_t)a[i]; }
            else { vg.indices.resize(count); for(size_t i = 0; i < count; ++i) vg.indices[i] = (uint32_t)i; }
            for(uint32_t ix: vg.indices) if(ix >= count) throw std::runtime_error("glTF: index out of range");
            if(!at.has("NORMAL")) calculate_normals(vg.vertices, vg.indices);
            if(!at.has("TANGENT")) calculate_tangents(vg.vertices, vg.indices);
            if(at.has("JOINTS_0"))
            {   // mesh::skin_data, weights renormalised (src/gltf.cc:722-731)
                int c; size_t n;
                const std::vector<double> jt = g.accessor(at.integer("JOINTS_0", 0), c, n);
                vg.skins.assign(count, trhip_skin{});
                for(size_t i = 0; i < count; ++i) for(int k = 0; k < 4; ++k) vg.skins[i].joints[k] = (uint32_t)jt[i * c + k];
                if(at.has("WEIGHTS_0"))
                {
                    int wc; size_t wn;
                    const std::vector<double> wt = g.accessor(at.integer("WEIGHTS_0", 0), wc, wn);
                    for(size_t i = 0; i < count; ++i)
                    {
                        const float w[4] = {(float)wt[i * wc], (float)wt[i * wc + 1], (float)wt[i * wc + 2], (float)wt[i * wc + 3]};
                        const float sum = ((w[0] + w[1]) + w[2]) + w[3];
                        for(int k = 0; k < 4; ++k) vg.skins[i].weights[k] = w[k] / sum;
                    }
                }
            }
            groups.push_back(std::move(vg));
        }
        models.push_back(std::move(groups));
    }

    scene_data s;
    std::vector<instance> instances;
    std::vector<mesh_span> spans;
    std::vector<vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<point_light> point_lights, spot_lights;
    std::vector<directional_light> dir_lights;
    std::vector<gltf_camera> cameras;
    double light_angle = 0, light_radius = 0;
    std::map<int, mat4d> node_globals;
    struct pending_skin { uint32_t instance; int skin; const std::vector<trhip_skin>* skins; };
    std::vector<pending_skin> skinned_pending;

    auto make_point_light = [&](const double color[3], const double pos[3], double radius) {
        point_light p{};
        for(int k = 0; k < 3; ++k) { p.color[k] = (float)color[k]; p.pos[k] = (float)pos[k]; }
        p.radius = (float)radius;
        const double cutoff_brightness = 5.0 / 256.0;
        p.cutoff_radius = (float)std::sqrt(std::max(color[0], std::max(color[1], color[2])) / cutoff_brightness);
        p.spot_radius = -1.0f;
        p.shadow_map_index = -1;
        return p;
    };

    auto anim = std::make_shared<gltf_animation>();
    std::vector<char> visited(list("nodes").size(), 0);
    std::function<void(int, const mat4d&, int)> visit = [&](int node_index, const mat4d& parent, int parent_index) {
        if(node_index < 0 || (size_t)node_index >= visited.size()) throw std::runtime_error("glTF: node index out of range");
        if(visited[(size_t)node_index]) throw std::runtime_error("glTF: node " + std::to_string(node_index) + " is reached twice: the node hierarchy must be a forest");
        visited[(size_t)node_index] = 1;
        const json& node = j.at("nodes").at((size_t)node_index);
        gltf_animation::node& rec = anim->nodes[node_index];
        rec = gltf_animation::node{};
        rec.parent = parent_index;
        if(const json* ch = node.find("children")) for(const json& c: ch->arr) rec.children.push_back((int)c.num);
        const json* ext = node.find("extensions");
        const json* tr = ext ? ext->find("TR_data") : nullptr;
        if(tr) if(const json* l = tr->find("light"))
        {
            if(l->has("angle")) light_angle = l->at("angle").num;
            if(l->has("radius")) light_radius = l->at("radius").num;
        }
        mat4d local;
        if(const json* m = node.find("matrix"))
        {   // column-major in the file
            for(int c = 0; c < 4; ++c) for(int r = 0; r < 4; ++r) local.m[r][c] = m->at(size_t(c * 4 + r)).num;
            rec.has_trs = false; rec.matrix = local;
        }
        else
        {
            double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, sc[3] = {1, 1, 1};
            if(const json* a = node.find("translation")) for(int k = 0; k < 3; ++k) t[k] = a->at((size_t)k).num;
            if(const json* a = node.find("rotation")) for(int k = 0; k < 4; ++k) q[k] = a->at((size_t)k).num;
            if(const json* a = node.find("scale")) for(int k = 0; k < 3; ++k) sc[k] = a->at((size_t)k).num;
            local = trs_matrix(t, q, sc);
            std::memcpy(rec.translation, t, sizeof(t)); std::memcpy(rec.rotation, q, sizeof(q)); std::memcpy(rec.scale, sc, sizeof(sc));
        }
        const mat4d glob = mul(parent, local);
        node_globals[node_index] = glob;

        if(node.has("mesh"))
        {
            double sto = 0;
            if(tr) if(const json* m = tr->find("mesh")) sto = m->number("shadow_terminator_offset", 0.0);
            const int mesh_index = node.integer("mesh", -1);
            if(mesh_index < 0 || (size_t)mesh_index >= models.size()) throw std::runtime_error("glTF: mesh index out of range");
            for(const vertex_group& vg: models[(size_t)mesh_index])
            {   // one INSTANCE record (src/scene_stage.cc:1085-1114)
                instance in{};
                in.light_base_id = -1; in.sh_grid_index = -1;
                in.shadow_terminator_mul = (float)(1.0 / (1.0 - 0.5 * sto));
                // glTF places skinned meshes at the origin; the loader enforces it (src/gltf.cc:777-784)
                const bool skinned = node.has("skin") && !vg.skins.empty();
                const mat4d model = skinned ? mat4d::identity() : glob;
                if(skinned) skinned_pending.push_back(pending_skin{(uint32_t)instances.size(), node.integer("skin", 0), &vg.skins});
                else anim->nodes[node_index].instances.push_back((uint32_t)instances.size());
                to_glm(model, in.model);
                to_glm(transpose(inverse(model)), in.model_normal);
                to_glm(model, in.model_prev);
                in.mat = vg.mat;
                instances.push_back(in);
                spans.push_back(mesh_span{(uint32_t)vertices.size(), (uint32_t)vg.vertices.size(), (uint32_t)indices.size(), (uint32_t)(vg.indices.size() / 3)});
                vertices.insert(vertices.end(), vg.vertices.begin(), vg.vertices.end());
                indices.insert(indices.end(), vg.indices.begin(), vg.indices.end());
            }
        }
        if(node.has("camera"))
        {
            const int camera_index = node.integer("camera", -1);
            if(camera_index < 0 || (size_t)camera_index >= list("cameras").size()) throw std::runtime_error("glTF: camera index out of range");
            const json& c = j.at("cameras").at((size_t)camera_index);
            gltf_camera cam{};
            cam.transform = glob;
            if(c.at("type").str == "perspective")
            {
                const json& pp = c.at("perspective");
                cam.perspective = true;
                cam.fov = pp.at("yfov").num * (180.0 / PI);
                cam.aspect = pp.number("aspectRatio", 1.0);
                cam.near = pp.at("znear").num;
                cam.far = pp.has("zfar") ? pp.at("zfar").num : std::numeric_limits<double>::infinity();
            }
            else
            {
                const json& o = c.at("orthographic");
                cam.perspective = false;
                const double xm = o.at("xmag").num, ym = o.at("ymag").num;
                const double v[6] = {-0.5 * xm, 0.5 * xm, -0.5 * ym, 0.5 * ym, o.at("znear").num, o.at("zfar").num};
                std::memcpy(cam.ortho, v, sizeof(v));
            }
            anim->nodes[node_index].cameras.push_back((uint32_t)cameras.size());
            cameras.push_back(cam);
        }
        if(const json* kl = ext ? ext->find("KHR_lights_punctual") : nullptr)
        {
            const json* all_ext = j.find("extensions");
            const json* punctual = all_ext ? all_ext->find("KHR_lights_punctual") : nullptr;
            const json* light_list = punctual ? punctual->find("lights") : nullptr;
            const int light_index = kl->integer("light", -1);
            if(!light_list || light_index < 0 || (size_t)light_index >= light_list->size()) throw std::runtime_error("glTF: light index out of range");
            const json& l = light_list->arr[(size_t)light_index];
            double color[3] = {1, 1, 1};
            if(const json* c = l.find("color")) for(int k = 0; k < 3; ++k) color[k] = c->at((size_t)k).num;
            const double intensity = l.number("intensity", 1.0);
            for(double& c: color) c = c * intensity;
            // get_global_direction: normalize(global orientation * (0, 0, -1)); orientation = the rotation part, columns normalised
            double direction[3], position[3];
            {
                double coln[3];
                for(int c = 0; c < 3; ++c) coln[c] = std::sqrt(glob.m[0][c] * glob.m[0][c] + glob.m[1][c] * glob.m[1][c] + glob.m[2][c] * glob.m[2][c]);
                for(int r = 0; r < 3; ++r)
                    direction[r] = (glob.m[r][0] / coln[0]) * 0.0 + (glob.m[r][1] / coln[1]) * 0.0 + (glob.m[r][2] / coln[2]) * -1.0;
                for(int r = 0; r < 3; ++r) position[r] = glob.m[r][3];
            }
            auto normalized = [](const double d[3], float out[3]) {
                const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                for(int k = 0; k < 3; ++k) out[k] = (float)(d[k] / n);
            };
            const std::string& type = l.at("type").str;
            if(type == "directional")
            {   // directional_light_entry (src/scene_stage.cc:46-62); TR_data.light.angle is in radians in the file
                directional_light d{};
                for(int k = 0; k < 3; ++k) d.color[k] = (float)color[k];
                d.shadow_map_index = -1;
                normalized(direction, d.dir);
                const double angle_deg = light_angle * (180.0 / PI);
                d.dir_cutoff = (float)std::cos(angle_deg * (PI / 180.0));
                anim->nodes[node_index].lights.push_back({true, (uint32_t)dir_lights.size()});
                dir_lights.push_back(d);
            }
            else if(type == "point" || type == "spot")
            {   // point_light_entry (src/scene_stage.cc:64-98): radiant intensity = power / 4 pi
                double c4[3];
                for(int k = 0; k < 3; ++k) c4[k] = color[k] / (4 * PI);
                point_light p = make_point_light(c4, position, light_radius);
                if(type == "spot")
                {
                    const json& sp = l.at("spot");
                    const double outer = sp.number("outerConeAngle", PI / 4) * (180.0 / PI), inner = sp.number("innerConeAngle", 0.0) * (180.0 / PI);
                    // spotlight::set_inner_angle (src/light.cc:103-112) with ratio 4 / 255
                    double falloff = 1.0;
                    if(inner > 0)
                    {
                        const double ci = std::cos(inner * (PI / 180.0)), co = std::cos(outer * (PI / 180.0));
                        falloff = std::log(4 / 255.0) / std::log(std::max(1.0 - ci, 0.0) / (1.0 - co));
                    }
                    normalized(direction, p.dir);
                    p.dir_cutoff = (float)std::cos(outer * (PI / 180.0));
                    p.dir_falloff = (float)falloff;
                    p.spot_radius = (float)(double(p.cutoff_radius) * std::tan(outer * (PI / 180.0)));
                    anim->nodes[node_index].lights.push_back({false, 0x80000000u | (uint32_t)spot_lights.size()});     // index fixed up below
                    spot_lights.push_back(p);
                }
                else
                {
                    anim->nodes[node_index].lights.push_back({false, (uint32_t)point_lights.size()});
                    point_lights.push_back(p);
                }
            }
        }
        if(const json* ch = node.find("children")) for(const json& c: ch->arr) visit((int)c.num, glob, node_index);
    };
    for(const json& sc: list("scenes").arr)
        if(const json* nodes = sc.find("nodes")) for(const json& n: nodes->arr) { anim->roots.push_back((int)n.num); visit((int)n.num, mat4d::identity(), -1); }

    // animation clips per node (src/gltf.cc:580-627): a channel fills the track its target path names in the target node's pool
    // entry of the clip's name; timestamps become microsecond ticks (read_animation_accessors, :167-190)
    for(const json& an: list("animations").arr)
        for(const json& chan: an.at("channels").arr)
        {
            const json& target = chan.at("target");
            if(!target.has("node")) continue;
            const std::string& path_name = target.at("path").str;
            const int width = path_name == "rotation" ? 4 : 3;
            if(path_name != "translation" && path_name != "rotation" && path_name != "scale") continue;      // morph-target weights
            const json& sampler = an.at("samplers").at((size_t)chan.integer("sampler", 0));
            gltf_animation::clip& cl = anim->clips[target.integer("node", 0)][an.has("name") ? an.at("name").str : std::string()];
            gltf_animation::track& tr = path_name == "translation" ? cl.position : (path_name == "rotation" ? cl.orientation : cl.scaling);
            tr = gltf_animation::track{};
            tr.width = width;
            const std::string interp = sampler.has("interpolation") ? sampler.at("interpolation").str : std::string("LINEAR");
            tr.interp = interp == "STEP" ? gltf_animation::STEP : (interp == "CUBICSPLINE" ? gltf_animation::CUBICSPLINE : gltf_animation::LINEAR);
            int comps; size_t count;
            const std::vector<double> times = g.accessor(sampler.integer("input", 0), comps, count);
            const size_t n = count;
            for(size_t k = 0; k < n; ++k) tr.timestamps.push_back((int64_t)std::floor((double)((float)times[k] * 1000000.0f) + 0.5));
            const std::vector<double> values = g.accessor(sampler.integer("output", 0), comps, count);
            auto key = [&](size_t index) { std::array<double, 4> v{}; for(int c = 0; c < width; ++c) v[(size_t)c] = (double)(float)values[index * (size_t)width + (size_t)c]; return v; };
            const bool tangents = count >= 3 * n;
            for(size_t k = 0; k < n; ++k)
            {
                if(tangents) { tr.in_tangent.push_back(key(3 * k)); tr.data.push_back(key(3 * k + 1)); tr.out_tangent.push_back(key(3 * k + 2)); }
                else tr.data.push_back(key(k));
            }
        }

    // light_base_id (src/scene_stage.cc:1069-1075)
    uint32_t tri_light_count = 0;
    for(size_t i = 0; i < instances.size(); ++i)
    {
        const float* e = instances[i].mat.emission_factor;
        if(lo.gather_emissive_triangles && (e[0] != 0 || e[1] != 0 || e[2] != 0)) { instances[i].light_base_id = (int32_t)tri_light_count; tri_light_count += spans[i].triangle_count; }
    }

    // cameras: set_camera_params forces the aspect ratio (src/tauray.cc:68-110), camera::write_uniform_buffer packs (src/camera.cc:431-478)
    const double aspect = lo.aspect_ratio > 0 ? (double)lo.aspect_ratio : double(width) / double(height);
    std::vector<camera_data> cams;
    for(gltf_camera& c: cameras) cams.push_back(pack_camera(c, aspect));
    anim->cameras = cameras; anim->aspect = aspect;
    if(!cameras.empty()) s.projection = cameras[0].perspective ? 0u : 1u;

    // point lights first, then spotlights (src/scene_stage.cc:1287-1317)
    for(auto& n: anim->nodes) for(auto& l: n.second.lights) if(!l.directional && (l.index & 0x80000000u)) l.index = (l.index & 0x7FFFFFFFu) + (uint32_t)point_lights.size();
    point_lights.insert(point_lights.end(), spot_lights.begin(), spot_lights.end());

    // texture table + material::potentially_transparent (src/material.cc:7-11, check_opaque src/gltf.cc:54-66)
    std::vector<texture_info> infos;
    std::vector<bool> opaque;
    for(const texture& t: textures)
    {
        infos.push_back(texture_info{t.w, t.h, (uint32_t)(s.texels.size() / 4), t.wide ? 1u : 0u});
        if(t.wide)
        {
            const uint8_t* b = reinterpret_cast<const uint8_t*>(t.rgba16.data());
            s.texels.insert(s.texels.end(), b, b + t.rgba16.size() * 2);
            opaque.push_back(false);      // check_opaque looks at 8-bit images only
            continue;
        }
        s.texels.insert(s.texels.end(), t.rgba.begin(), t.rgba.end());
        bool op = true;
        for(size_t i = 3; i < t.rgba.size(); i += 4) if(t.rgba[i] != 255) { op = false; break; }
        opaque.push_back(op);
    }
    for(const instance& in: instances)
    {
        bool tr = in.mat.transmittance > 0 || in.mat.albedo_factor[3] < 1.0f;
        if(in.mat.albedo_tex >= 0 && (size_t)in.mat.albedo_tex < opaque.size() && !opaque[(size_t)in.mat.albedo_tex]) tr = true;
        s.non_opaque.push_back(tr ? 1 : 0);
    }

    auto bytes = [](const auto& v, std::vector<uint8_t>& out) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(v.data());
        out.assign(p, p + v.size() * sizeof(v[0]));
    };
    bytes(instances, s.instances); bytes(spans, s.spans); bytes(vertices, s.vertices); bytes(indices, s.indices);
    bytes(point_lights, s.point_lights); bytes(dir_lights, s.directional_lights); bytes(infos, s.texture_infos); bytes(cams, s.cameras);
    s.gather_emissive_triangles = tri_light_count > 0 ? 1u : 0u;
    // skins: per joint the global transform of its node in the rest pose times the inverse bind matrix (model::update_joints)
    for(const pending_skin& ps: skinned_pending)
    {
        const json& sk = j.at("skins").at((size_t)ps.skin);
        const json& joints = sk.at("joints");
        std::vector<double> ibm;
        if(sk.has("inverseBindMatrices")) { int c; size_t n; ibm = g.accessor(sk.integer("inverseBindMatrices", 0), c, n); }
        scene_data::skinned_mesh out;
        out.instance = ps.instance;
        out.skins = *ps.skins;
        anim->skins.emplace_back();
        for(size_t k = 0; k < joints.size(); ++k)
        {
            mat4d inv_bind = mat4d::identity();
            if(ibm.size() >= (k + 1) * 16) for(int c = 0; c < 4; ++c) for(int r = 0; r < 4; ++r) inv_bind.m[r][c] = ibm[k * 16 + size_t(c) * 4 + r];   // column-major in the file
            anim->skins.back().joint_nodes.push_back((int)joints.at(k).num);
            anim->skins.back().inverse_bind.push_back(inv_bind);
            const auto it = node_globals.find((int)joints.at(k).num);
            const mat4d jt = mul(it != node_globals.end() ? it->second : mat4d::identity(), inv_bind);
            float m[16];
            to_glm(jt, m);
            out.joint_transforms.insert(out.joint_transforms.end(), m, m + 16);
        }
        s.skinned.push_back(std::move(out));
    }
    s.animation = anim;
    return s;
}

//---------------------------------------------------------------------------------------------------------------------
// generate_cameras (src/tauray.cc:680-727): the light-field camera grid of `--camera-grid=w,h,x,y` - grid_w x grid_h copies of
// the scene's first camera on a plane through it (spacing dx, dy; rolled by roll_deg about the view axis; shifted by `offset`),
// each panned so that the views coincide at `recentering_distance` (camera::set_pan).  Replaces scene_data::cameras by the grid
// (row by row from the top, like the reference's camera indices); returns the number of viewports.  For scenes load_glb made.
inline uint32_t generate_cameras(scene_data& s, int grid_w, int grid_h, double dx, double dy, double recentering_distance = 5.0, double roll_deg = 0.0,
                                 const double offset[3] = nullptr)
{
    using namespace gltf_detail;
    if(!s.animation || s.animation->cameras.empty()) throw std::runtime_error("generate_cameras: the scene has no camera to build the grid around");
    if(grid_w < 1 || grid_h < 1) throw std::runtime_error("generate_cameras: empty grid");
    const gltf_camera parent = s.animation->cameras[0];
    if(!parent.perspective) throw std::runtime_error("generate_cameras: the grid pans perspective cameras");
    constexpr double PI = 3.14159265358979323846;
    const double aspect = s.animation->aspect;
    const double width = (grid_w - 1) * dx, height = (grid_h - 1) * dy;
    const double tv = std::tan(parent.fov * (PI / 180.0) * 0.5), th = aspect * tv;      // tan(vfov / 2), tan(hfov / 2)
    const double c = std::cos(roll_deg * (PI / 180.0)), sn = std::sin(roll_deg * (PI / 180.0));
    std::vector<gltf_camera> grid;
    std::vector<camera_data> packed;
    for(int y = 0; y < grid_h; ++y)
        for(int x = 0; x < grid_w; ++x)
        {
            const double gx = -width * 0.5 + x * dx, gy = height * 0.5 - y * dy;
            const double gp[3] = {c * gx - sn * gy, sn * gx + c * gy, 0.0};
            gltf_camera sub = parent;
            sub.pan[0] = -gp[0] / (th * recentering_distance);
            sub.pan[1] = -gp[1] / (tv * recentering_distance);
            mat4d local = mat4d::identity();
            for(int k = 0; k < 3; ++k) local.m[k][3] = gp[k] + (offset ? offset[k] : 0.0);
            sub.transform = mul(parent.transform, local);
            packed.push_back(pack_camera(sub, aspect));
            grid.push_back(sub);
        }
    s.animation->cameras = grid;
    s.cameras.resize(packed.size() * sizeof(camera_data));
    std::memcpy(s.cameras.data(), packed.data(), s.cameras.size());
    s.previous_cameras.clear();
    return (uint32_t)packed.size();
}

//---------------------------------------------------------------------------------------------------------------------
// play(scene, name, loop, fallback) / update(scene, dt) / is_playing(scene) of src/scene.cc:213-244 with the controller of
// src/animation.tcc:79-205 (a queue of one clip), over a scene load_glb produced: `tauray --animation[=name] --framerate F`.
// update() rewrites scene_data::instances (model, model_normal; model_prev = last frame's model), ::cameras (previous_cameras =
// last frame's), the joint matrices of ::skinned and the light records; scene_stage::apply / rt_renderer::update_scene send them
// to the device.
// Punctual lights on moving nodes get a new position / direction in scene_data::point_lights / directional_lights.
class scene_animator
{
public:
    explicit scene_animator(scene_data& s): scene(&s), anim(s.animation)
    {
        if(anim) for(const auto& p: anim->clips) controllers[p.first] = controller{};
    }

    void play(const std::string& name = "", bool loop = false)
    {
        if(!anim) return;
        for(auto& p: controllers)
        {
            controller& c = p.second;
            const auto& pool = anim->clips.at(p.first);
            c.timer = 0;
            auto it = pool.find(name);
            c.current = it != pool.end() ? &it->second : (name.empty() && !pool.empty() ? &pool.begin()->second : nullptr);   // use_fallback = no name given
            c.loop_time = c.current ? c.current->loop_time() : 0;
            c.playing = c.loop_time != 0;
            c.loop = loop;
        }
    }

    bool is_playing() const { for(const auto& p: controllers) if(p.second.playing) return true; return false; }

    // dt in microsecond ticks: 0 for the first frame, round(1e6 / framerate) afterwards (src/tauray.cc:1052,1090)
    void update(int64_t dt)
    {
        using namespace gltf_detail;
        if(!anim) return;
        scene->previous_cameras = scene->cameras;
        for(auto& p: controllers)
        {
            controller& c = p.second;
            gltf_animation::node& node = anim->nodes.at(p.first);
            if(!c.playing || !node.has_trs) continue;
            c.timer += dt;
            if(c.loop) c.timer %= c.loop_time;
            else if(c.timer >= c.loop_time) { c.playing = false; c.loop_time = 0; c.timer = 0; continue; }   // the node keeps its last pose
            // animation::apply (src/animation.cc:40-53)
            if(c.current->position.width) { const auto v = c.current->position.sample(c.timer, false); for(int k = 0; k < 3; ++k) node.translation[k] = v[(size_t)k]; }
            if(c.current->scaling.width) { const auto v = c.current->scaling.sample(c.timer, false); for(int k = 0; k < 3; ++k) node.scale[k] = v[(size_t)k]; }
            if(c.current->orientation.width)
            {
                auto q = c.current->orientation.sample(c.timer, true);
                if(c.current->orientation.interp == gltf_animation::CUBICSPLINE)
                {
                    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                    for(double& v: q) v = v / n;
                }
                for(int k = 0; k < 4; ++k) node.rotation[k] = q[(size_t)k];
            }
        }
        instance* inst = reinterpret_cast<instance*>(scene->instances.data());
        const uint32_t n_inst = scene->instance_count();
        for(uint32_t i = 0; i < n_inst; ++i) std::memcpy(inst[i].model_prev, inst[i].model, sizeof(inst[i].model));
        std::map<int, mat4d> globals;
        std::function<void(int, const mat4d&)> visit = [&](int n, const mat4d& parent) {
            const gltf_animation::node& node = anim->nodes.at(n);
            const mat4d glob = mul(parent, node.local());
            globals[n] = glob;
            for(uint32_t i: node.instances) { to_glm(glob, inst[i].model); to_glm(transpose(inverse(glob)), inst[i].model_normal); }
            for(uint32_t ci: node.cameras) anim->cameras[ci].transform = glob;
            for(const gltf_animation::node::light& l: node.lights)
            {   // get_global_direction / get_global_position as in load_glb; colours, radii and cone angles do not move
                double coln[3], direction[3];
                for(int c = 0; c < 3; ++c) coln[c] = std::sqrt(glob.m[0][c] * glob.m[0][c] + glob.m[1][c] * glob.m[1][c] + glob.m[2][c] * glob.m[2][c]);
                for(int r = 0; r < 3; ++r) direction[r] = (glob.m[r][0] / coln[0]) * 0.0 + (glob.m[r][1] / coln[1]) * 0.0 + (glob.m[r][2] / coln[2]) * -1.0;
                const double dn = std::sqrt(direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2]);
                if(l.directional)
                {
                    directional_light& d = reinterpret_cast<directional_light*>(scene->directional_lights.data())[l.index];
                    for(int k = 0; k < 3; ++k) d.dir[k] = (float)(direction[k] / dn);
                }
                else
                {
                    point_light& pl = reinterpret_cast<point_light*>(scene->point_lights.data())[l.index];
                    for(int k = 0; k < 3; ++k) pl.pos[k] = (float)glob.m[k][3];
                    if(pl.spot_radius >= 0) for(int k = 0; k < 3; ++k) pl.dir[k] = (float)(direction[k] / dn);
                }
            }
            for(int ch: node.children) visit(ch, glob);
        };
        for(int r: anim->roots) visit(r, mat4d::identity());
        camera_data* cams = reinterpret_cast<camera_data*>(scene->cameras.data());
        for(size_t ci = 0; ci < anim->cameras.size(); ++ci) cams[ci] = pack_camera(anim->cameras[ci], anim->aspect);
        for(size_t k = 0; k < anim->skins.size() && k < scene->skinned.size(); ++k)
        {   // model::update_joints (src/model.cc:107-118)
            const gltf_animation::skin& sk = anim->skins[k];
            std::vector<float>& out = scene->skinned[k].joint_transforms;
            out.resize(sk.joint_nodes.size() * 16);
            for(size_t jn = 0; jn < sk.joint_nodes.size(); ++jn)
            {
                const auto it = globals.find(sk.joint_nodes[jn]);
                to_glm(mul(it != globals.end() ? it->second : mat4d::identity(), sk.inverse_bind[jn]), out.data() + jn * 16);
            }
        }
    }

private:
    struct controller { const gltf_animation::clip* current = nullptr; bool loop = false, playing = false; int64_t timer = 0, loop_time = 0; };
    scene_data* scene;
    std::shared_ptr<gltf_animation> anim;
    std::map<int, controller> controllers;
};

// writes the .trsc dump load_scene_dump reads (the format of tauray_amd/scene_io.py)
inline void write_scene_dump(const scene_data& s, const std::string& path)
{
    std::ofstream f(path, std::ios::binary);
    if(!f) throw std::runtime_error("Failed to write " + path);
    const uint32_t version = 1;
    f.write("TRSC", 4); f.write(reinterpret_cast<const char*>(&version), 4);
    const std::vector<uint8_t>* sections[] = {&s.instances, &s.spans, &s.vertices, &s.indices, &s.point_lights, &s.directional_lights,
        &s.texture_infos, &s.texels, &s.envmap, &s.alias_table, &s.cameras, &s.non_opaque};
    for(const auto* sec: sections)
    {
        const uint64_t n = sec->size();
        f.write(reinterpret_cast<const char*>(&n), 8);
        f.write(reinterpret_cast<const char*>(sec->data()), (std::streamsize)n);
    }
    f.write(reinterpret_cast<const char*>(&s.envmap_width), 4); f.write(reinterpret_cast<const char*>(&s.envmap_height), 4);
    f.write(reinterpret_cast<const char*>(s.environment_factor), 16);
    f.write(reinterpret_cast<const char*>(&s.gather_emissive_triangles), 4); f.write(reinterpret_cast<const char*>(&s.projection), 4);
}

}

#endif
