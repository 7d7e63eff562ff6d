// Host-only robustness check of the file readers of the C++ mirror: mutates good files (byte flips, truncation, length fields)
// and feeds them to rth::png::decode, rth::jpeg::decode and rth::assets::loadAsset under AddressSanitizer + UBSan.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iraytracer_amd/host tools/fuzz_decoders.cpp -lz -o /tmp/fuzz_decoders
//   /tmp/fuzz_decoders <iterations per seed file> <seed files...>
// A reader may refuse a mutated file or decode garbage; it must not read or write out of bounds, overflow or hang.
#include "rt_assets.hpp"
#include "rt_jpeg.hpp"
#include "rt_png.hpp"

#include <cstdio>
#include <random>
#include <string>
#include <vector>

static bool ends_with(const std::string& s, const char* e) { const size_t n = strlen(e); return s.size() >= n && s.compare(s.size() - n, n, e) == 0; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s iterations files...\n", argv[0]); return 2; }
    const int iters = atoi(argv[1]);
    std::mt19937 rng(12345);
    long decoded = 0, refused = 0;
    for (int a = 2; a < argc; a++) {
        const std::string path = argv[a];
        std::string good;
        if (!rth::assets::readFile(path, good)) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 2; }
        const size_t dot = path.rfind('.');
        const std::string ext = dot == std::string::npos ? "" : path.substr(dot);
        const size_t slash = path.find_last_of('/');
        const std::string tmp = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/fuzz_case" + ext;   // beside the seed: relative references (mtllib) resolve
        for (int it = 0; it < iters; it++) {
            std::string d = good;
            const int kind = (int)(rng() % 4u);
            if (kind == 0) d.resize(rng() % (d.size() + 1));                                   // truncation
            else if (kind == 1) { const int n = 1 + (int)(rng() % 8u); for (int k = 0; k < n; k++) d[rng() % d.size()] = (char)rng(); }   // byte flips
            else if (kind == 2) { const size_t p = rng() % d.size(); const size_t n = 1 + rng() % 16u; for (size_t k = p; k < p + n && k < d.size(); k++) d[k] = (char)0xFF; }   // a run of 0xFF (markers, huge lengths)
            else { const size_t p = rng() % d.size(); const size_t n = std::min(d.size() - p, (size_t)(1 + rng() % 64u)); d.erase(p, n); }   // a deleted span
            FILE* f = fopen(tmp.c_str(), "wb");
            fwrite(d.data(), 1, d.size(), f);
            fclose(f);
            bool ok = false;
            if (ends_with(path, ".png")) { std::vector<unsigned char> px; int w, h, n; ok = rth::png::decode(tmp, px, w, h, n); if (ok && px.size() != (size_t)w * h * n) { fprintf(stderr, "size mismatch\n"); return 1; } }
            else if (ends_with(path, ".jpg") || ends_with(path, ".jpeg")) { std::vector<unsigned char> px; int w, h, n; ok = rth::jpeg::decode(tmp, px, w, h, n); if (ok && px.size() != (size_t)w * h * n) { fprintf(stderr, "size mismatch\n"); return 1; } }
            else {
                rth::assets::FlatAsset as;
                ok = rth::assets::loadAsset(tmp, as);
                if (ok) for (auto& m : as.meshes) {   // what a caller would index with
                    const size_t nv = m.positions.size() / 3;
                    for (auto i : m.indices) if ((size_t)i >= nv) { fprintf(stderr, "index out of range in %s\n", path.c_str()); return 1; }
                }
            }
            (ok ? decoded : refused)++;
        }
        fprintf(stderr, "%s: done\n", path.c_str());
    }
    printf("decoded %ld refused %ld\n", decoded, refused);
    return 0;
}
