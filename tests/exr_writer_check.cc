// Test helper (tests/test_cpp_host.py): writes a synthetic RGBA32F image through tr::headless (include/tauray_hip.hh) with the
// requested pixel format and compression, and dumps the source pixels next to it.
// usage: exr_writer_check <width> <height> <pixel_format 0..3> <compression 0..4> <out.exr> <out.raw>
#include "tauray_hip.hh"
#include <random>
int main(int argc, char** argv) {
    const unsigned w = (unsigned)atoi(argv[1]), h = (unsigned)atoi(argv[2]);
    const int fmt = atoi(argv[3]), comp = atoi(argv[4]);
    std::vector<float> img((size_t)w * h * 4);
    std::mt19937 rng(w * 31 + h);
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    for (unsigned y = 0; y < h; ++y) for (unsigned x = 0; x < w; ++x) {
        float* p = &img[((size_t)y * w + x) * 4];
        // smooth gradients + noise + flat regions + large values
        p[0] = 0.5f + 0.5f * std::sin(x * 0.05f) * std::cos(y * 0.03f) + (x > w / 2 ? 0.01f * u(rng) : 0.0f);
        p[1] = (y < h / 3) ? 0.25f : 10.0f * u(rng);
        p[2] = (float)x / w * 1000.0f;
        p[3] = (x + y) % 7 == 0 ? 0.0f : 1.0f;
    }
    tr::headless::options o;
    o.size = {w, h};
    o.output_format = (tr::headless::pixel_format)fmt;
    o.output_compression = (tr::headless::compression_type)comp;
    tr::headless hl(o);
    hl.write_image(argv[5], img.data());
    FILE* f = fopen(argv[6], "wb");
    fwrite(img.data(), 4, img.size(), f);
    fclose(f);
    return 0;
}
