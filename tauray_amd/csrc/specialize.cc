// Run-time specialisation of the shading program (shade_spec.hip) through hipRTC, with a cache of code objects on disk.
//
// The reference builds its ray-tracing pipeline when a stage is constructed, with the stage's options as #defines
// (src/path_tracer_stage.cc:30-116 load_sources + src/rt_pipeline.cc; shaderc at run time, src/shader_source.cc).  Here the
// options are data for the ahead-of-time kernels of libtrhip.so; a stage whose option set has no ahead-of-time instance gets
// one compiled the first time it renders: the device sources of the shading kernels travel inside the library as text
// (rtc_sources.inc, written by tools/embed_sources.py at build time), hipRTC compiles shade_spec.hip with the option set as
// -DTR_SPEC_* macros for the device's architecture, the code object is kept in this process and in the kernel cache
// (spec_cache_dir()), and the kernels are launched through the module API.  __graft_entry__.build() fills the cache for the
// reference's presets and for the option sets the tests and the bench use (trhip_pt_precompile), so a GPU box starts warm.
#include "specialize.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

#include "rtc_sources.inc"

namespace tr {

namespace {

const char* k_prelude =
    "typedef unsigned char uint8_t; typedef signed char int8_t; typedef unsigned short uint16_t; typedef short int16_t;\n"
    "typedef unsigned int uint32_t; typedef int int32_t; typedef unsigned long long uint64_t; typedef long long int64_t;\n"
    "typedef unsigned long uintptr_t;\n"
    "#include \"shade_spec.hip\"\n";

std::vector<std::string> spec_flags(const SpecRequest& r, const std::string& arch) {
    const trhip_pt_options& o = r.opt;
    std::vector<std::string> f = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=off"};
    auto d = [&](const char* name, int v) { f.push_back(std::string("-DTR_SPEC_") + name + "=" + std::to_string(v)); };
    d("PROGRAM", r.program);
    if (r.program == SPEC_RAYGEN) {   // what k_raygen reads: sampler (its Sobol index), film filter, depth of field, projection
        d("SAMPLER", o.sampler); d("FILM", o.film); d("PROJECTION", o.projection); d("DOF", o.depth_of_field != 0);
        for (const char* unused : {"MIS", "BOUNCE_MODE", "TRI_LIGHT_MODE", "ROULETTE", "CLAMPING", "REGULARIZATION", "HIDE_LIGHTS", "WHITE_ALBEDO", "TRANSPARENT",
                                   "PRE_TRANSFORMED", "NEE_POINT", "NEE_TRI", "NEE_DIR", "NEE_ENV", "SHADE_TRIS", "COUNT", "WIDE_TEXTURES"}) d(unused, 0);
        return f;
    }
    if (!r.ieee) { f.push_back("-fno-hip-fp32-correctly-rounded-divide-sqrt"); f.push_back("-DTR_SHADE_NATIVE_MATH=1"); }
    d("SAMPLER", o.sampler); d("FILM", o.film); d("MIS", o.mis_mode); d("BOUNCE_MODE", o.bounce_mode); d("TRI_LIGHT_MODE", o.tri_light_mode);
    d("PROJECTION", o.projection);
    d("ROULETTE", o.russian_roulette_delta > 0.0f); d("CLAMPING", o.indirect_clamping > 0.0f); d("REGULARIZATION", o.regularization_gamma != 0.0f);
    d("DOF", o.depth_of_field != 0); d("HIDE_LIGHTS", o.hide_lights != 0); d("WHITE_ALBEDO", o.use_white_albedo_on_first_bounce != 0);
    d("TRANSPARENT", o.transparent_background != 0); d("PRE_TRANSFORMED", o.pre_transformed_vertices != 0);
    d("NEE_POINT", o.nee_point > 0.0f); d("NEE_TRI", o.nee_triangles > 0.0f); d("NEE_DIR", o.nee_directional > 0.0f); d("NEE_ENV", o.nee_envmap > 0.0f);
    d("SHADE_TRIS", r.shade_tris); d("COUNT", r.count); d("WIDE_TEXTURES", r.wide_textures);
    return f;
}

uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001B3ull; }
    return h;
}

// what the code object depends on: flags (option set, arithmetic, architecture), the embedded sources, the compiler
std::string cache_name(const SpecRequest& r, const std::string& arch) {
    uint64_t h = 0xCBF29CE484222325ull;
    for (const std::string& f : spec_flags(r, arch)) h = fnv1a(h, f.c_str(), f.size() + 1);
    for (const auto& s : k_rtc_sources) { h = fnv1a(h, s.name, strlen(s.name) + 1); h = fnv1a(h, s.text, strlen(s.text) + 1); }
    h = fnv1a(h, k_prelude, strlen(k_prelude));
    int major = 0, minor = 0;
    if (hiprtcVersion(&major, &minor) == HIPRTC_SUCCESS) { h = fnv1a(h, &major, sizeof(major)); h = fnv1a(h, &minor, sizeof(minor)); }
    char buf[64];
    snprintf(buf, sizeof(buf), "spec_%016llx.hsaco", (unsigned long long)h);
    return buf;
}

bool dir_usable(const std::string& d) {
    struct stat st;
    if (stat(d.c_str(), &st) != 0 && mkdir(d.c_str(), 0755) != 0 && stat(d.c_str(), &st) != 0) return false;
    return access(d.c_str(), W_OK | X_OK) == 0;
}

// A cache file is the code object followed by a 24-byte trailer (magic, length, FNV-1a of the code): a file cut short by a full disk
// or overwritten by something else is not handed to the loader, it counts as absent and is compiled again.
constexpr char k_trailer_magic[8] = {'T', 'R', 'H', 'S', 'A', 'C', 'O', '1'};
constexpr size_t k_trailer_bytes = 24;

void append_trailer(std::vector<char>& file) {
    const uint64_t n = file.size(), h = fnv1a(0xCBF29CE484222325ull, file.data(), file.size());
    file.insert(file.end(), k_trailer_magic, k_trailer_magic + 8);
    file.insert(file.end(), reinterpret_cast<const char*>(&n), reinterpret_cast<const char*>(&n) + 8);
    file.insert(file.end(), reinterpret_cast<const char*>(&h), reinterpret_cast<const char*>(&h) + 8);
}

bool read_file(const std::string& path, std::vector<char>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    if (n <= (std::streamoff)k_trailer_bytes) return false;
    out.resize((size_t)n);
    f.seekg(0);
    if (!f.read(out.data(), n)) return false;
    const size_t code = (size_t)n - k_trailer_bytes;
    uint64_t len = 0, h = 0;
    memcpy(&len, out.data() + code + 8, 8);
    memcpy(&h, out.data() + code + 16, 8);
    if (memcmp(out.data() + code, k_trailer_magic, 8) != 0 || len != code || h != fnv1a(0xCBF29CE484222325ull, out.data(), code)) return false;
    out.resize(code);
    return true;
}

int compile(const SpecRequest& r, const std::string& arch, std::vector<char>& code, std::string* why) {
    std::vector<const char*> texts, names;
    for (const auto& s : k_rtc_sources) { texts.push_back(s.text); names.push_back(s.name); }
    hiprtcProgram prog = nullptr;
    hiprtcResult rc = hiprtcCreateProgram(&prog, k_prelude, "trhip_spec.hip", (int)texts.size(), texts.data(), names.data());
    if (rc != HIPRTC_SUCCESS) { if (why) *why = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(rc); return 1; }
    const std::vector<std::string> flags = spec_flags(r, arch);
    std::vector<const char*> opts;
    for (const std::string& f : flags) opts.push_back(f.c_str());
    rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        (void)hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
        if (why) *why = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(rc) + "\n" + log.substr(0, 4000);
        (void)hiprtcDestroyProgram(&prog);
        return 1;
    }
    size_t n = 0;
    (void)hiprtcGetCodeSize(prog, &n);
    code.resize(n);
    rc = hiprtcGetCode(prog, code.data());
    (void)hiprtcDestroyProgram(&prog);
    if (rc != HIPRTC_SUCCESS || n == 0) { if (why) *why = "hiprtcGetCode failed"; return 1; }
    return 0;
}

// the code object of `r`: cache file, else compiled (and written to the cache, atomically)
int code_object(const SpecRequest& r, const std::string& arch, std::vector<char>& code, bool* compiled, std::string* why) {
    const std::string dir = spec_cache_dir();
    const std::string path = dir.empty() ? std::string() : dir + "/" + cache_name(r, arch);
    if (compiled) *compiled = false;
    if (!path.empty() && read_file(path, code)) return 0;
    if (int rc = compile(r, arch, code, why)) return rc;
    if (compiled) *compiled = true;
    if (!path.empty()) {
        const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
        std::vector<char> file = code;
        append_trailer(file);
        bool written = false;
        { std::ofstream f(tmp, std::ios::binary); f.write(file.data(), (std::streamsize)file.size()); f.flush(); written = (bool)f; }
        if (!written || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
    }
    return 0;
}

bool drop_cached(const SpecRequest& r, const std::string& arch) {
    const std::string dir = spec_cache_dir();
    return !dir.empty() && remove((dir + "/" + cache_name(r, arch)).c_str()) == 0;
}

struct Loaded { hipModule_t module = nullptr; SpecKernels k; };
std::mutex g_mutex;
std::map<std::string, Loaded> g_loaded;      // key: device + architecture + spec flags
std::map<std::string, std::string> g_failed; // requests that could not be built: not retried every frame

}  // namespace

unsigned long long spec_fnv1a(unsigned long long h, const void* p, size_t n) { return fnv1a(h, p, n); }

unsigned long long spec_sources_hash() {
    static const unsigned long long hash = [] {
        uint64_t h = 0xCBF29CE484222325ull;
        for (const auto& s : k_rtc_sources) { h = fnv1a(h, s.name, strlen(s.name) + 1); h = fnv1a(h, s.text, strlen(s.text) + 1); }
        int major = 0, minor = 0;
        if (hiprtcVersion(&major, &minor) == HIPRTC_SUCCESS) { h = fnv1a(h, &major, sizeof(major)); h = fnv1a(h, &minor, sizeof(minor)); }
        return (unsigned long long)h;
    }();
    return hash;
}

std::string spec_key(const SpecRequest& r) {
    std::string s;
    for (const std::string& f : spec_flags(r, "")) if (f.rfind("-DTR_SPEC_", 0) == 0) s += f.substr(10) + " ";
    s += (r.ieee || r.program == SPEC_RAYGEN) ? "ieee" : "vulkan-grade";
    return s;
}

std::string spec_cache_dir() {
    static const std::string dir = [] {
        if (const char* e = getenv("TRHIP_KERNEL_CACHE")) return std::string(dir_usable(e) ? e : "");
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&spec_cache_dir), &info) && info.dli_fname) {
            std::string lib = info.dli_fname;
            const size_t slash = lib.rfind('/');
            const std::string d = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/kernel_cache";
            if (dir_usable(d)) return d;
        }
        if (const char* home = getenv("HOME")) {
            const std::string c = std::string(home) + "/.cache";
            if (dir_usable(c) && dir_usable(c + "/trhip")) return c + "/trhip";
        }
        return std::string();
    }();
    return dir;
}

int spec_precompile(const SpecRequest& r, const char* arch, std::string* why) {
    std::vector<char> code;
    std::string a = arch && *arch ? arch : "gfx950";
    return code_object(r, a.substr(0, a.find(':')), code, nullptr, why);
}

const SpecKernels* spec_kernels(const SpecRequest& r, std::string* why) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { if (why) *why = "no current HIP device"; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { if (why) *why = "hipGetDeviceProperties failed"; return nullptr; }
    // gcnArchName is a full target id ("gfx950:sramecc+:xnack-"); trhip_pt_precompile and build() compile for the bare processor, and the
    // cache file's name hashes the flags: with the feature suffix left on, every program warmed ahead of time missed and was compiled
    // again through hipRTC on first render.  A code object for the bare processor loads under any setting of the features.
    std::string arch = prop.gcnArchName;
    arch = arch.substr(0, arch.find(':'));
    if (arch.empty()) arch = "gfx950";
    std::string key = std::to_string(dev) + " " + arch;
    for (const std::string& f : spec_flags(r, arch)) key += " " + f;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_loaded.find(key);
    if (it != g_loaded.end()) return &it->second.k;
    auto bad = g_failed.find(key);
    if (bad != g_failed.end()) { if (why) *why = bad->second; return nullptr; }
    std::vector<char> code;
    std::string err;
    bool compiled = false;
    Loaded l;
    bool ok = code_object(r, arch, code, &compiled, &err) == 0;
    if (ok && hipModuleLoadData(&l.module, code.data()) != hipSuccess) {
        (void)hipGetLastError();
        l.module = nullptr;
        // a cache file that does not load (truncated by a full disk, written by another compiler): compiled again, once
        if (!compiled && drop_cached(r, arch) && code_object(r, arch, code, &compiled, &err) == 0 && compiled &&
            hipModuleLoadData(&l.module, code.data()) == hipSuccess) {
            fprintf(stderr, "[trhip] the kernel cache held a program that does not load for {%s}: compiled again\n", spec_key(r).c_str());
        } else {
            ok = false; err = "hipModuleLoadData failed for " + spec_key(r); (void)hipGetLastError();
        }
    }
    if (ok && (r.program == SPEC_RAYGEN ? hipModuleGetFunction(&l.k.raygen, l.module, "trhip_spec_raygen") != hipSuccess
                                        : (hipModuleGetFunction(&l.k.shade, l.module, "trhip_spec_shade") != hipSuccess ||
                                           hipModuleGetFunction(&l.k.shade_last, l.module, "trhip_spec_shade_last") != hipSuccess))) {
        ok = false; err = "the specialised program lacks a kernel"; (void)hipGetLastError();
    }
    if (!ok) { g_failed[key] = err; if (why) *why = err; return nullptr; }
    l.k.code_hash = fnv1a(0xCBF29CE484222325ull, code.data(), code.size());
    if (getenv("TRHIP_DEBUG")) fprintf(stderr, "[trhip] shading program for {%s}: %s\n", spec_key(r).c_str(), compiled ? "compiled through hipRTC" : "from the kernel cache");
    return &(g_loaded[key] = l).k;
}

}  // namespace tr
