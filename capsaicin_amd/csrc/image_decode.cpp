// image_decode.cpp — texture file decoding for the host layer: what TextureSystem gets from stbi_load(file, &w, &h, &n, 4) in the
// reference (src/core/src/systems/texture_system.cpp:41-45): 8-bit RGBA, rows top to bottom, grey replicated, alpha 255 when
// the file has none.  Own decoders (the reference vendors stb_image.h, a third-party header that is not carried over):
//   PNG   colour types 0 / 2 / 3 / 4 / 6, bit depths 1..16, all five filters, zlib inflate (stored / fixed / dynamic blocks);
//         Adam7-interlaced files are refused
//   TGA   types 1 / 2 / 3 and their run-length forms 9 / 10 / 11, 8 / 15 / 16 / 24 / 32 bits, either origin
//   PPM   binary P6, maxval 255 (the container tools/make_sponza_class.py writes)
// JPEG is not decoded: such a texture is reported missing, which the reference treats as a warning and a black texel
// (texture_system.cpp:50-56).
#include "../../include/capsaicin_scene.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <exception>
#include <vector>

namespace
{
// largest image the decoders allocate for: 64 Mi pixels (256 MB of RGBA8) -- sizes come from the file's header, before any data
constexpr uint64_t kMaxPixels = 1ull << 26;

typedef std::vector<uint8_t> Bytes;

// ---------------------------------------------------------------- inflate (RFC 1950 / 1951)
struct BitReader
{
    const uint8_t* p;
    size_t         n, pos = 0;
    uint32_t       acc = 0;
    int            bits = 0;
    bool           fail = false;
    uint32_t       get(int count)
    {
        while (bits < count)
        {
            if (pos >= n)
            {
                fail = true;
                return 0;
            }
            acc |= (uint32_t)p[pos++] << bits;
            bits += 8;
        }
        const uint32_t v = count ? (acc & ((count == 32 ? 0u : (1u << count)) - 1u)) : 0u;
        acc >>= count;
        bits -= count;
        return v;
    }
    void align() { acc = 0, bits = 0; }
};

struct Huffman
{
    uint16_t count[16] = {0};
    uint16_t symbol[288];
    bool     build(const uint8_t* lengths, int n)
    {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) ++count[lengths[i]];
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len)
        {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len)
        {
            code |= (int)br.get(1);
            if (br.fail) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// `limit`: the caller knows how many bytes the stream may produce (PNG: (stride + 1) * height); one byte more is a bomb or damage
bool inflate_blocks(BitReader& br, Bytes* out, size_t limit)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29]  = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30]  = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    bool last = false;
    while (!last)
    {
        last = br.get(1) != 0;
        const uint32_t type = br.get(2);
        if (br.fail) return false;
        if (type == 0)
        {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) return false;
            if (out->size() + len > limit) return false;
            out->insert(out->end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        uint8_t lengths[320];
        if (type == 1)
        {
            int i = 0;
            for (; i < 144; ++i) lengths[i] = 8;
            for (; i < 256; ++i) lengths[i] = 9;
            for (; i < 280; ++i) lengths[i] = 7;
            for (; i < 288; ++i) lengths[i] = 8;
            lit.build(lengths, 288);
            for (i = 0; i < 30; ++i) lengths[i] = 5;
            dist.build(lengths, 30);
        }
        else
        {
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
            if (br.fail || nlen > 286 || ndist > 30) return false;
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.get(3);
            Huffman lencode;
            if (!lencode.build(cl, 19)) return false;
            int i = 0;
            while (i < nlen + ndist)
            {
                const int sym = lencode.decode(br);
                if (sym < 0) return false;
                if (sym < 16)
                    lengths[i++] = (uint8_t)sym;
                else
                {
                    int     rep  = 0;
                    uint8_t prev = 0;
                    if (sym == 16)
                    {
                        if (i == 0) return false;
                        prev = lengths[i - 1], rep = 3 + (int)br.get(2);
                    }
                    else if (sym == 17)
                        rep = 3 + (int)br.get(3);
                    else
                        rep = 11 + (int)br.get(7);
                    if (br.fail || i + rep > nlen + ndist) return false;
                    while (rep--) lengths[i++] = prev;
                }
            }
            if (lengths[256] == 0) return false;
            if (!lit.build(lengths, nlen)) return false;
            if (!dist.build(lengths + nlen, ndist)) return false;
        }
        while (true)
        {
            const int sym = lit.decode(br);
            if (sym < 0) return false;
            if (sym < 256)
            {
                if (out->size() >= limit) return false;
                out->push_back((uint8_t)sym);
            }
            else if (sym == 256)
                break;
            else
            {
                const int ls = sym - 257;
                if (ls >= 29) return false;
                const int len = lbase[ls] + (int)br.get(lext[ls]);
                const int ds  = dist.decode(br);
                if (ds < 0 || ds >= 30) return false;
                const size_t d = dbase[ds] + br.get(dext[ds]);
                if (br.fail || d > out->size() || out->size() + (size_t)len > limit) return false;
                const size_t from = out->size() - d;
                for (int k = 0; k < len; ++k) out->push_back((*out)[from + k]);
            }
        }
    }
    return true;
}

bool zlib_inflate(const uint8_t* p, size_t n, Bytes* out, size_t limit)
{
    if (n < 6 || (p[0] & 0x0f) != 8 || ((p[0] << 8) | p[1]) % 31 != 0 || (p[1] & 0x20)) return false;
    BitReader br{p + 2, n - 2};
    return inflate_blocks(br, out, limit);  // the Adler-32 trailer is not checked: a damaged file fails the size / filter checks below
}

// ---------------------------------------------------------------- PNG
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool decode_png(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) return false;
    size_t   pos = 8;
    uint32_t W = 0, H = 0;
    int      depth = 0, ctype = 0, interlace = 0;
    Bytes    idat, plte, trns;
    bool     have_ihdr = false, end = false;
    while (!end && pos + 12 <= d.size())
    {
        const uint32_t len = be32(&d[pos]);
        const char*    tag = (const char*)&d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return false;
        const uint8_t* body = &d[pos + 8];
        if (!memcmp(tag, "IHDR", 4))
        {
            if (len != 13) return false;
            W = be32(body), H = be32(body + 4), depth = body[8], ctype = body[9], interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return false;
            have_ihdr = true;
        }
        else if (!memcmp(tag, "PLTE", 4))
            plte.assign(body, body + len);
        else if (!memcmp(tag, "tRNS", 4))
            trns.assign(body, body + len);
        else if (!memcmp(tag, "IDAT", 4))
            idat.insert(idat.end(), body, body + len);
        else if (!memcmp(tag, "IEND", 4))
            end = true;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !W || !H || W > 32768 || H > 32768 || (uint64_t)W * H > kMaxPixels || interlace != 0) return false;
    int channels;
    switch (ctype)
    {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return false;
    if (ctype == 3 && (depth == 16 || plte.size() < 3)) return false;
    const size_t bpp_bits = (size_t)channels * depth, stride = (W * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    Bytes        raw;
    raw.reserve((stride + 1) * H);
    if (!zlib_inflate(idat.data(), idat.size(), &raw, (stride + 1) * H) || raw.size() < (stride + 1) * H) return false;
    // un-filter in place (PNG spec 9.2): a = left, b = up, c = upper left
    Bytes prev(stride, 0);
    for (uint32_t y = 0; y < H; ++y)
    {
        uint8_t*      row = &raw[(stride + 1) * y + 1];
        const uint8_t f   = row[-1];
        for (size_t i = 0; i < stride; ++i)
        {
            const int a = i >= bpp ? row[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int       pred;
            switch (f)
            {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4:
            {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred        = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return false;
            }
            row[i] = (uint8_t)(row[i] + pred);
        }
        memcpy(prev.data(), row, stride);
    }
    rgba->assign((size_t)W * H * 4, 255);
    for (uint32_t y = 0; y < H; ++y)
    {
        const uint8_t* row = &raw[(stride + 1) * y + 1];
        for (uint32_t x = 0; x < W; ++x)
        {
            uint8_t* o = &(*rgba)[4 * ((size_t)y * W + x)];
            // sample k of the pixel as an 8-bit value (16-bit: the high byte, as stb's 16 -> 8 conversion; < 8 bits: scaled)
            auto sample = [&](int k) -> int {
                if (depth == 8) return row[(size_t)x * channels + k];
                if (depth == 16) return row[2 * ((size_t)x * channels + k)];
                const size_t bit = (size_t)x * depth;
                const int    v   = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
            };
            if (ctype == 3)
            {
                const size_t idx = (size_t)sample(0);
                if (3 * idx + 2 < plte.size()) o[0] = plte[3 * idx], o[1] = plte[3 * idx + 1], o[2] = plte[3 * idx + 2];
                else o[0] = o[1] = o[2] = 0;
                o[3] = idx < trns.size() ? trns[idx] : 255;
            }
            else if (ctype == 0 || ctype == 4)
            {
                o[0] = o[1] = o[2] = (uint8_t)sample(0);
                if (ctype == 4) o[3] = (uint8_t)sample(1);
            }
            else
            {
                o[0] = (uint8_t)sample(0), o[1] = (uint8_t)sample(1), o[2] = (uint8_t)sample(2);
                if (ctype == 6) o[3] = (uint8_t)sample(3);
            }
        }
    }
    // tRNS colour keys of grey / RGB images (rare in textures) are honoured for 8-bit files
    if (depth == 8 && ((ctype == 0 && trns.size() >= 2) || (ctype == 2 && trns.size() >= 6)))
        for (size_t i = 0; i < (size_t)W * H; ++i)
        {
            uint8_t* o = &(*rgba)[4 * i];
            if (ctype == 0 ? o[0] == trns[1] : (o[0] == trns[1] && o[1] == trns[3] && o[2] == trns[5])) o[3] = 0;
        }
    *w = W, *h = H;
    return true;
}

// ---------------------------------------------------------------- TGA
bool decode_tga(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    if (d.size() < 18) return false;
    const int idlen = d[0], cmap_type = d[1], type = d[2];
    const int cmap_first = d[3] | (d[4] << 8), cmap_len = d[5] | (d[6] << 8), cmap_bits = d[7];
    const uint32_t W = d[12] | (d[13] << 8), H = d[14] | (d[15] << 8);
    const int      bits = d[16], desc = d[17];
    const bool     rle = type == 9 || type == 10 || type == 11;
    const int      base = rle ? type - 8 : type;
    if (!(base == 1 || base == 2 || base == 3) || !W || !H || (uint64_t)W * H > kMaxPixels) return false;
    // colour-map entries are expanded like pixels: only the depths expand() knows (anything else was read as 16 bits: wrong
    // colours and a 1-byte over-read for an 8-bit map)
    if (cmap_type == 1 && base == 1 && !(cmap_bits == 15 || cmap_bits == 16 || cmap_bits == 24 || cmap_bits == 32)) return false;
    if (base == 1 && (cmap_type != 1 || bits != 8)) return false;
    if (base == 2 && !(bits == 15 || bits == 16 || bits == 24 || bits == 32)) return false;
    if (base == 3 && !(bits == 8 || bits == 16)) return false;
    if (cmap_type > 1) return false;
    size_t pos = 18 + (size_t)idlen;
    const size_t cmap_bytes = cmap_type ? (size_t)cmap_len * ((cmap_bits + 7) / 8) : 0;
    if (pos + cmap_bytes > d.size()) return false;
    const uint8_t* cmap = &d[pos];
    pos += cmap_bytes;
    const size_t px = (size_t)(bits + 7) / 8;
    Bytes        raw((size_t)W * H * px);
    if (!rle)
    {
        if (pos + raw.size() > d.size()) return false;
        memcpy(raw.data(), &d[pos], raw.size());
    }
    else
    {
        size_t o = 0;
        while (o < raw.size())
        {
            if (pos >= d.size()) return false;
            const int hdr = d[pos++], n = (hdr & 127) + 1;
            if (o + (size_t)n * px > raw.size()) return false;
            if (hdr & 128)
            {
                if (pos + px > d.size()) return false;
                for (int k = 0; k < n; ++k, o += px) memcpy(&raw[o], &d[pos], px);
                pos += px;
            }
            else
            {
                if (pos + (size_t)n * px > d.size()) return false;
                memcpy(&raw[o], &d[pos], (size_t)n * px);
                pos += (size_t)n * px, o += (size_t)n * px;
            }
        }
    }
    auto expand = [](const uint8_t* p, int b, uint8_t* o) {  // one colour value of b bits -> RGBA
        if (b == 24 || b == 32)
            o[0] = p[2], o[1] = p[1], o[2] = p[0], o[3] = b == 32 ? p[3] : 255;
        else  // 15 / 16: A RRRRR GGGGG BBBBB, little endian
        {
            const int v = p[0] | (p[1] << 8);
            o[0] = (uint8_t)(((v >> 10) & 31) * 255 / 31), o[1] = (uint8_t)(((v >> 5) & 31) * 255 / 31), o[2] = (uint8_t)((v & 31) * 255 / 31);
            o[3] = 255;
        }
    };
    rgba->assign((size_t)W * H * 4, 255);
    const bool top = (desc & 0x20) != 0, right = (desc & 0x10) != 0;
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x)
        {
            const uint8_t* p = &raw[((size_t)y * W + x) * px];
            uint8_t*       o = &(*rgba)[4 * ((size_t)(top ? y : H - 1 - y) * W + (right ? W - 1 - x : x))];
            if (base == 3)
            {
                o[0] = o[1] = o[2] = p[0];
                if (bits == 16) o[3] = p[1];
            }
            else if (base == 1)
            {
                const int idx = (int)p[0] - cmap_first;
                if (idx < 0 || idx >= cmap_len) o[0] = o[1] = o[2] = 0;
                else expand(cmap + (size_t)idx * ((cmap_bits + 7) / 8), cmap_bits, o);
            }
            else
                expand(p, bits, o);
        }
    *w = W, *h = H;
    return true;
}

// ---------------------------------------------------------------- PPM (P6, maxval 255)
bool decode_ppm(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    if (d.size() < 2 || d[0] != 'P' || d[1] != '6') return false;
    size_t pos = 0;
    auto   token = [&]() {
        std::string t;
        while (pos < d.size())
        {
            if (d[pos] == '#')
                while (pos < d.size() && d[pos] != '\n') ++pos;
            else if (std::isspace(d[pos]))
                ++pos;
            else
                break;
        }
        while (pos < d.size() && !std::isspace(d[pos])) t += (char)d[pos++];
        return t;
    };
    if (token() != "P6") return false;
    const long W = std::atol(token().c_str()), H = std::atol(token().c_str()), M = std::atol(token().c_str());
    if (W <= 0 || H <= 0 || M != 255 || W > 32768 || H > 32768 || (uint64_t)W * (uint64_t)H > kMaxPixels) return false;
    ++pos;  // single whitespace after maxval
    if (d.size() < pos + (size_t)W * H * 3) return false;
    rgba->resize((size_t)W * H * 4);
    for (size_t i = 0; i < (size_t)W * H; ++i)
    {
        (*rgba)[4 * i + 0] = d[pos + 3 * i + 0];
        (*rgba)[4 * i + 1] = d[pos + 3 * i + 1];
        (*rgba)[4 * i + 2] = d[pos + 3 * i + 2];
        (*rgba)[4 * i + 3] = 255;
    }
    *w = (uint32_t)W, *h = (uint32_t)H;
    return true;
}
}  // namespace

extern "C" void cap_set_error_(const char* msg);

extern "C" int cap_image_decode(const uint8_t* bytes, size_t size, const char* name_hint, uint8_t** out_rgba8, uint32_t* out_width,
                                uint32_t* out_height)
{
    if (!bytes || !out_rgba8 || !out_width || !out_height)
    {
        cap_set_error_("cap_image_decode: NULL argument");
        return CAP_ERR_INVALID_ARG;
    }
    Bytes    rgba;
    uint32_t w = 0, h = 0;
    bool     ok = false;
    try  // no exception crosses the C ABI: a header that asks for more memory than there is ends as a status, not std::terminate
    {
    const Bytes d(bytes, bytes + size);
    ok = decode_png(d, &rgba, &w, &h) || decode_ppm(d, &rgba, &w, &h);
    if (!ok)
    {
        // TGA has no signature: only tried for a .tga name, or as a last resort without a name
        std::string n = name_hint ? name_hint : "";
        for (auto& c : n) c = (char)std::tolower((unsigned char)c);
        const bool tga_name = n.size() >= 4 && n.compare(n.size() - 4, 4, ".tga") == 0;
        if (tga_name || n.empty()) ok = decode_tga(d, &rgba, &w, &h);
    }
    }
    catch (const std::exception&)
    {
        cap_set_error_("cap_image_decode: out of memory");
        return CAP_ERR_IO;
    }
    if (!ok)
    {
        cap_set_error_("cap_image_decode: not a PNG (non-interlaced), TGA or binary PPM this build decodes");
        return CAP_ERR_UNSUPPORTED;
    }
    uint8_t* p = (uint8_t*)std::malloc(rgba.size());
    if (!p)
    {
        cap_set_error_("cap_image_decode: out of memory");
        return CAP_ERR_IO;
    }
    memcpy(p, rgba.data(), rgba.size());
    *out_rgba8 = p, *out_width = w, *out_height = h;
    return CAP_OK;
}

extern "C" void cap_image_free(uint8_t* rgba8) { std::free(rgba8); }
