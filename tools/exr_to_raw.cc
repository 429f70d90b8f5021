// Tool-time only: decodes an OpenEXR file to raw float32 RGB using the tinyexr
// header vendored by the reference (third-party, BSD-3).  Built in /tmp by
// tools/make_golden.py; nothing from it is committed except the decoded data.
#define TINYEXR_IMPLEMENTATION
#define TINYEXR_USE_MINIZ 1
#include "tinyexr.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    EXRVersion version;
    if (ParseEXRVersionFromFile(&version, argv[1]) != 0) return 3;
    EXRHeader header;
    InitEXRHeader(&header);
    const char* err = nullptr;
    if (ParseEXRHeaderFromFile(&header, &version, argv[1], &err) != 0) { fprintf(stderr, "%s\n", err); return 4; }
    for (int i = 0; i < header.num_channels; i++) header.requested_pixel_types[i] = TINYEXR_PIXELTYPE_FLOAT;
    EXRImage image;
    InitEXRImage(&image);
    if (LoadEXRImageFromFile(&image, &header, argv[1], &err) != 0) { fprintf(stderr, "%s\n", err); return 5; }
    int w = image.width, h = image.height;
    int ci[3] = {-1, -1, -1};
    for (int i = 0; i < header.num_channels; i++) {
        if (!strcmp(header.channels[i].name, "R")) ci[0] = i;
        if (!strcmp(header.channels[i].name, "G")) ci[1] = i;
        if (!strcmp(header.channels[i].name, "B")) ci[2] = i;
    }
    std::vector<float> out((size_t)w * h * 3);
    for (int c = 0; c < 3; c++) {
        const float* src = (const float*)image.images[ci[c]];
        for (size_t p = 0; p < (size_t)w * h; p++) out[p * 3 + c] = src[p];
    }
    FILE* f = fopen(argv[2], "wb");
    int dims[2] = {w, h};
    fwrite(dims, 4, 2, f);
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    printf("%d %d %d\n", w, h, header.num_channels);
    return 0;
}
