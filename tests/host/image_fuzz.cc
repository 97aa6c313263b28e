// tests/host/image_fuzz.cc — the image decoders (cortex.llamacpp_amd/host/image_decode.cc) take bytes straight from a request: whatever they are given, they must
// answer with an image or a reason, never read or write out of bounds, never hang.  This driver (built with -fsanitize=address,undefined by
// tests/test_image_decode.py) reads sample files, decodes each, then decodes many seeded mutations of each: byte flips, truncations, spliced garbage, length
// fields blown up.  usage: image_fuzz <mutations per file> file...
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "clip.h"

using namespace mi355;

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }

int main(int argc, char **argv) {
    const int n_mut = argc > 1 ? atoi(argv[1]) : 200;
    long decoded = 0, refused = 0;
    for (int a = 2; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
        std::vector<uint8_t> base;
        uint8_t buf[65536];
        for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) base.insert(base.end(), buf, buf + n);
        fclose(f);
        ClipImageU8 img;
        const std::string err0 = clip_image_load_from_bytes(base.data(), base.size(), img);
        if (!err0.empty() && std::string(argv[a]).find("refuse") == std::string::npos) { fprintf(stderr, "%s does not decode: %s\n", argv[a], err0.c_str()); return 3; }
        for (int m = 0; m < n_mut; m++) {
            std::vector<uint8_t> v = base;
            const int kind = (int)(rnd() % 6);
            if (kind == 0) { const int k = 1 + (int)(rnd() % 8); for (int i = 0; i < k; i++) v[rnd() % v.size()] ^= (uint8_t)(1u << (rnd() % 8)); }
            else if (kind == 1) v.resize(rnd() % v.size());                                                  // truncation
            else if (kind == 2) { const size_t at = rnd() % v.size(), len = 1 + rnd() % 64; for (size_t i = at; i < v.size() && i < at + len; i++) v[i] = (uint8_t)rnd(); }
            else if (kind == 3) { const size_t at = rnd() % v.size(); for (size_t i = at; i < v.size() && i < at + 4; i++) v[i] = 0xff; }   // a length / marker field blown up
            else if (kind == 4) { const size_t at = rnd() % v.size(); v.insert(v.begin() + (long)at, (size_t)(rnd() % 300), (uint8_t)rnd()); }
            else { const size_t at = rnd() % v.size(); for (size_t i = at; i < v.size() && i < at + 16; i++) v[i] = 0; }
            ClipImageU8 out;
            const std::string err = clip_image_load_from_bytes(v.data(), v.size(), out);
            if (err.empty()) {
                if (out.nx <= 0 || out.ny <= 0 || out.rgb.size() != (size_t)3 * out.nx * out.ny) { fprintf(stderr, "inconsistent image from a mutation of %s\n", argv[a]); return 4; }
                decoded++;
            } else refused++;
        }
    }
    printf("mutations decoded %ld refused %ld\n", decoded, refused);
    return 0;
}
