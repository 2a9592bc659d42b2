#!/bin/bash
# Host-code sanitizer pass (CPU only; GPU sanitizers are not available on this pool).
#  1. sah_builder.cpp + obj_loader.cpp + image_decode.cpp + jpeg_decode.cpp under ASan/UBSan with a small driver (random boxes, the
#     Cornell OBJ, a missing file; good, truncated, oversized-by-header and bomb PNG / TGA / PPM files; every JPEG of
#     tests/golden/images and every generated file whole, cut at every fifth byte and with 400 random byte edits each);
#  2. the oracle rebuilt with ASan/UBSan and the CPU oracle tests run against it (the regular .so is restored afterwards).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
W="$(mktemp -d /tmp/cap_asan.XXXXXX)"
trap 'rm -rf "$W"' EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"

cat > "$W/main.cpp" <<'EOF'
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
extern "C" int  cap_host_sah_build(const float*, uint32_t, float*, uint32_t*, uint32_t*);
extern "C" int  cap_obj_load(const char*, const char*, void**);
extern "C" void cap_geometry_free(void*);
extern "C" int  cap_image_decode(const uint8_t*, size_t, const char*, uint8_t**, uint32_t*, uint32_t*);
extern "C" void cap_image_free(uint8_t*);
extern "C" void cap_set_error_(const char* m) { fprintf(stderr, "  (error text: %s)\n", m); }
extern "C" int  cap_scene_upload(void*, const float*, const float*, const float*, const uint32_t*, const void*, uint32_t, uint32_t, uint32_t) { return 0; }
int main(int argc, char** argv)
{
    for (uint32_t n : {0u, 1u, 2u, 3u, 7u, 100u, 5000u, 200000u})
    {
        std::vector<float> b(8 * (size_t)n + 8);
        srand(n);
        for (uint32_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k)
            {
                // n == 7: all boxes identical (degenerate centroid extent)
                const float c = n == 7 ? 1.0f : (rand() % 2000) * 0.01f, e = (rand() % 100) * 0.001f;
                b[8 * i + k] = c - e, b[8 * i + 4 + k] = c + e;
            }
        std::vector<float>    nodes(16 * (size_t)(n > 1 ? n - 1 : 1));
        std::vector<uint32_t> order(n + 1);
        uint32_t              depth = 0;
        const int             rc    = cap_host_sah_build(b.data(), n, nodes.data(), order.data(), &depth);
        printf("sah n=%u rc=%d depth=%u\n", n, rc, depth);
        if (rc != 0) return 1;
    }
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        const std::string ext = a.size() > 4 ? a.substr(a.size() - 4) : "";
        if (ext == ".png" || ext == ".tga" || ext == ".ppm" || ext == ".jpg" || ext == ".bmp")
        {
            FILE* f = fopen(argv[i], "rb");
            if (!f) return 1;
            std::vector<uint8_t> d;
            uint8_t              buf[65536];
            size_t               n;
            while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
            fclose(f);
            int ok = 0, bad = 0;
            // the whole file, then every prefix in steps of 5 bytes
            for (size_t cut = d.size(); cut > 0; cut = cut == d.size() ? d.size() - 1 : (cut > 5 ? cut - 5 : 0))
            {
                std::vector<uint8_t> part(d.begin(), d.begin() + cut);  // exact-size heap block: over-reads are caught
                uint8_t*  px = nullptr;
                uint32_t  w = 0, h = 0;
                const int rc = cap_image_decode(part.data(), part.size(), argv[i], &px, &w, &h);
                (rc == 0 ? ok : bad)++;
                if (px) cap_image_free(px);
            }
            printf("image %s: %d prefixes decoded, %d refused\n", argv[i], ok, bad);
            {
                srand(12345);
                for (int trial = 0; trial < 400 && d.size() > 2; ++trial)  // (a fixture of two bytes or fewer has nothing to edit)
                {
                    std::vector<uint8_t> part(d);
                    for (int e = 1 + rand() % 4; e > 0; --e) part[2 + rand() % (part.size() - 2)] = (uint8_t)rand();
                    uint8_t* px = nullptr;
                    uint32_t w = 0, h = 0;
                    cap_image_decode(part.data(), part.size(), argv[i], &px, &w, &h);
                    if (px) cap_image_free(px);
                }
                printf("image %s: 400 edited copies survived\n", argv[i]);
            }
            continue;
        }
        void*     g  = nullptr;
        const int rc = cap_obj_load(argv[i], "", &g);
        printf("obj %s rc=%d\n", argv[i], rc);
        if (g) cap_geometry_free(g);
        // round 6: the hot records are parsed in place (obj_loader.cpp fast_float / fast_index) -- the file cut at every third byte and
        // with 600 random byte edits (digits, signs, slashes, blanks, line ends where numbers and indices stand): load or refuse, never fault
        std::vector<char> text;
        if (FILE* f = fopen(argv[i], "rb"))
        {
            char buf[4096];
            size_t n;
            while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.insert(text.end(), buf, buf + n);
            fclose(f);
        }
        if (text.empty() || text.size() > 200000) continue;
        const std::string tmp = std::string(argv[i]) + ".asan_tmp.obj";
        auto run = [&](const std::vector<char>& t) {
            if (FILE* f = fopen(tmp.c_str(), "wb"))
            {
                fwrite(t.data(), 1, t.size(), f);
                fclose(f);
            }
            void* h = nullptr;
            const int r2 = cap_obj_load(tmp.c_str(), "", &h);
            if (h) cap_geometry_free(h);
            return r2;
        };
        int ok = 0, refused = 0;
        for (size_t cut = 0; cut < text.size(); cut += 3) (run(std::vector<char>(text.begin(), text.begin() + cut)) == 0 ? ok : refused)++;
        srand(12345);
        const char alphabet[] = "0123456789+-./eE \t\r\n#vfnto";
        for (int k = 0; k < 600; ++k)
        {
            std::vector<char> t = text;
            for (int e = 0; e < 1 + rand() % 4; ++e) t[(size_t)rand() % t.size()] = alphabet[(size_t)rand() % (sizeof(alphabet) - 1)];
            (run(t) == 0 ? ok : refused)++;
        }
        remove(tmp.c_str());
        printf("obj %s: %d cut / edited copies loaded, %d refused\n", argv[i], ok, refused);
    }
    return 0;
}
EOF
g++ -std=c++17 $SAN -I"$ROOT/include" -I"$ROOT/capsaicin_amd/csrc" "$W/main.cpp" \
    "$ROOT/capsaicin_amd/csrc/sah_builder.cpp" "$ROOT/capsaicin_amd/csrc/obj_loader.cpp" "$ROOT/capsaicin_amd/csrc/image_decode.cpp" \
    "$ROOT/capsaicin_amd/csrc/jpeg_decode.cpp" -o "$W/host"
python3 - "$W" <<'EOF'
import io, struct, sys, zlib
import numpy as np
from PIL import Image
w = sys.argv[1]
rs = np.random.RandomState(1)
a = (np.add.outer(np.arange(23) * 3, np.arange(37) * 5)[..., None] + rs.randint(0, 9, (23, 37, 4))) % 256
Image.fromarray(a.astype(np.uint8), "RGBA").save(w + "/rgba.png")
Image.fromarray(a[..., :3].astype(np.uint8), "RGB").save(w + "/rgb0.png", compress_level=0)
p = Image.fromarray(rs.randint(0, 17, (23, 37)).astype(np.uint8), "P"); p.putpalette(rs.randint(0, 256, 51).astype(np.uint8).tolist()); p.save(w + "/pal.png")
Image.fromarray(a[..., :3].astype(np.uint8), "RGB").save(w + "/rle.tga", compression="tga_rle")
Image.fromarray(a.astype(np.uint8), "RGBA").save(w + "/raw.tga")
Image.fromarray(a[..., :3].astype(np.uint8), "RGB").save(w + "/p6.ppm")
def chunk(tag, body): return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
def png(wd, h, depth, ctype, payload): return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", wd, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", payload) + chunk(b"IEND", b"")
open(w + "/huge.png", "wb").write(png(32768, 32768, 16, 6, zlib.compress(b"\0" * 64)))
open(w + "/bomb.png", "wb").write(png(4, 4, 8, 2, zlib.compress(b"\0" * (8 << 20), 9)))
open(w + "/huge.tga", "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 65535, 65535, 32, 0) + b"\0" * 64)
open(w + "/cmap8.tga", "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 1, 1, 0, 4, 8, 0, 0, 2, 2, 8, 0) + bytes(8))
EOF
"$W/host" "$ROOT/assets/cornell_box.obj" /nonexistent/none.obj "$W"/*.png "$W"/*.tga "$W"/*.ppm "$ROOT"/tests/golden/images/*.jpg "$ROOT"/tests/golden/images/*.png \
    "$ROOT"/tests/golden/images/*.tga "$ROOT"/tests/golden/images/*.ppm "$ROOT"/tests/golden/images/*.bmp

g++ -std=c++17 $SAN -fPIC -ffp-contract=off -mfma -fno-fast-math -pthread -shared -o "$W/libcap_oracle.so" \
    "$ROOT/oracle/cap_oracle.cpp" "$ROOT/oracle/cap_oracle_post.cpp"
cp "$ROOT/oracle/libcap_oracle.so" "$W/orig.so"
restore() { cp "$W/orig.so" "$ROOT/oracle/libcap_oracle.so"; rm -rf "$W"; }
trap restore EXIT
cp "$W/libcap_oracle.so" "$ROOT/oracle/libcap_oracle.so"
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_oracle_post.py tests/test_oracle_kat.py tests/test_golden_frames.py -x -q -m "not gpu"
echo "asan_host: clean"
