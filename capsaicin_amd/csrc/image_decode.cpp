// image_decode.cpp — texture file decoding for the host layer: what TextureSystem gets from stbi_load(file, &w, &h, &n, 4) in the
// reference (src/core/src/systems/texture_system.cpp:41-45): 8-bit RGBA, rows top to bottom, grey replicated, alpha 255 when
// the file has none.  Own decoders (the reference vendors stb_image.h, a third-party header that is not carried over):
//   PNG   colour types 0 / 2 / 3 / 4 / 6, bit depths 1..16, all five filters, Adam7 interlace, zlib inflate (stored / fixed /
//         dynamic blocks)
//   TGA   types 1 / 2 / 3 and their run-length forms 9 / 10 / 11, 8 / 15 / 16 / 24 / 32 bits, either origin (stb's reading of the
//         format where it departs from Truevision's)
//   PNM   binary P6 (the container tools/make_sponza_class.py writes) and P5, maxval up to 255
//   JPEG  baseline and progressive Huffman-coded frames (jpeg_decode.cpp)
//   BMP   1 / 4 / 8 bits with a palette, 16 / 24 / 32 bits (channel masks), every header size stb reads, not run-length coded
// Anything else (GIF, PSD, HDR, PIC) is reported missing, which the reference treats as a warning and a
// black texel (texture_system.cpp:50-56).  tests/test_image_ref.py holds every decoder to stb_image's output bit for bit.
#include "../../include/capsaicin_scene.h"
#include "image_decode.h"

#include <algorithm>
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
    bool     have_ihdr = false, have_trns = false, end = false;
    while (!end && pos + 12 <= d.size())
    {
        const uint32_t len = be32(&d[pos]);
        const char*    tag = (const char*)&d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return false;
        const uint8_t* body = &d[pos + 8];
        if (!memcmp(tag, "IHDR", 4))
        {
            if (len != 13 || have_ihdr) return false;
            W = be32(body), H = be32(body + 4), depth = body[8], ctype = body[9], interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return false;
            have_ihdr = true;
        }
        else if (!memcmp(tag, "PLTE", 4))
        {
            if (!have_ihdr || len > 768 || len % 3 != 0) return false;
            plte.assign(body, body + len);
        }
        else if (!memcmp(tag, "tRNS", 4))
        {
            // stb_image.h:4936-4957: a key only for grey / RGB, exactly one 16-bit value per channel; palette alpha after PLTE
            if (!have_ihdr || !idat.empty()) return false;
            if (ctype == 3 ? (plte.empty() || len > plte.size() / 3) : ((ctype & 4) || len != (ctype == 2 ? 6u : 2u))) return false;
            trns.assign(body, body + len);
            have_trns = true;
        }
        else if (!memcmp(tag, "IDAT", 4))
        {
            if (!have_ihdr) return false;
            idat.insert(idat.end(), body, body + len);
        }
        else if (!memcmp(tag, "IEND", 4))
            end = true;
        else if (!have_ihdr || !(tag[0] & 0x20))
            return false;  // something before IHDR, or a critical chunk this decoder does not know (stb refuses both)
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !W || !H || W > 32768 || H > 32768 || (uint64_t)W * H > kMaxPixels || interlace > 1) return false;
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
    const size_t bpp_bits = (size_t)channels * depth, bpp = (bpp_bits + 7) / 8;
    // the image as one pass, or as the seven Adam7 passes (PNG spec 8.2): pixel (x0 + i * dx, y0 + j * dy) of the image is pixel
    // (i, j) of the pass; every pass is a filtered image of its own
    struct Pass
    {
        uint32_t x0, y0, dx, dy;
    };
    static const Pass whole[1] = {{0, 0, 1, 1}};
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    const Pass*       passes   = interlace ? adam7 : whole;
    const int         n_passes = interlace ? 7 : 1;
    size_t            total    = 0;
    for (int p = 0; p < n_passes; ++p)
    {
        const size_t pw = (W - passes[p].x0 + passes[p].dx - 1) / passes[p].dx, ph = (H - passes[p].y0 + passes[p].dy - 1) / passes[p].dy;
        if (W > passes[p].x0 && H > passes[p].y0) total += ((pw * bpp_bits + 7) / 8 + 1) * ph;
    }
    // deflate cannot expand by more than 1032 : 1 (258 bytes from two bits of a fixed-code match, RFC 1951): a stream too short for the
    // header's size is refused before anything of that size is reserved
    if ((uint64_t)idat.size() * 1032ull + 1032ull < (uint64_t)total) return false;
    Bytes raw;
    raw.reserve(total);
    if (!zlib_inflate(idat.data(), idat.size(), &raw, total) || raw.size() < total) return false;
    rgba->assign((size_t)W * H * 4, 255);
    size_t offset = 0;
    for (int p = 0; p < n_passes; ++p)
    {
        const Pass& ps = passes[p];
        if (W <= ps.x0 || H <= ps.y0) continue;
        const uint32_t pw = (W - ps.x0 + ps.dx - 1) / ps.dx, ph = (H - ps.y0 + ps.dy - 1) / ps.dy;
        const size_t   stride = ((size_t)pw * bpp_bits + 7) / 8;
        // un-filter in place (PNG spec 9.2): a = left, b = up, c = upper left
        Bytes prev(stride, 0);
        for (uint32_t y = 0; y < ph; ++y)
        {
            uint8_t*      row = &raw[offset + (stride + 1) * y + 1];
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
                    const int q = a + b - c, pa = abs(q - a), pb = abs(q - b), pc = abs(q - c);
                    pred        = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
                }
                row[i] = (uint8_t)(row[i] + pred);
            }
            memcpy(prev.data(), row, stride);
            for (uint32_t x = 0; x < pw; ++x)
            {
                uint8_t* o = &(*rgba)[4 * ((size_t)(ps.y0 + y * ps.dy) * W + ps.x0 + x * ps.dx)];
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
                // colour key of grey / RGB files (stb_image.h:4952-4956): 16-bit files compare whole samples, the others the low
                // byte of the key scaled like the samples
                if (have_trns && ctype != 3)
                {
                    bool hit = true;
                    for (int k = 0; k < channels; ++k)
                    {
                        const int key = (trns[2 * k] << 8) | trns[2 * k + 1];
                        if (depth == 16)
                        {
                            const size_t at = 2 * ((size_t)x * channels + k);
                            hit             = hit && ((row[at] << 8) | row[at + 1]) == key;
                        }
                        else
                            hit = hit && sample(k) == (uint8_t)((key & 255) * (255 / ((1 << depth) - 1)));
                    }
                    if (hit) o[3] = 0;
                }
            }
        }
        offset += (stride + 1) * ph;
    }
    *w = W, *h = H;
    return true;
}

// ---------------------------------------------------------------- TGA
// The rules are stb's (stb_image.h:5565-5806), not the Truevision text, wherever the two part: the pixel layout follows the bit
// count (8 grey, 15 / 16 five-bit RGB -- grey + alpha only for image type 3 --, 24 BGR, 32 BGRA) whatever the image type says;
// "first colour-map entry" is a count of BYTES skipped in front of the map and indices are not offset by it; an index beyond the
// map reads entry 0; the right-to-left bit is ignored; only the top-to-bottom bit flips rows.
bool decode_tga(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    if (d.size() < 18) return false;
    const int idlen = d[0], cmap_type = d[1], type = d[2];
    const int cmap_skip = d[3] | (d[4] << 8), cmap_len = d[5] | (d[6] << 8), cmap_bits = d[7];
    const uint32_t W = d[12] | (d[13] << 8), H = d[14] | (d[15] << 8);
    const int      bits = d[16], desc = d[17];
    const bool     rle = type >= 8, indexed = cmap_type == 1;
    const int      base = rle ? type - 8 : type;
    auto depth_ok = [](int b) { return b == 8 || b == 15 || b == 16 || b == 24 || b == 32; };
    if (cmap_type > 1 || !W || !H || (uint64_t)W * H > kMaxPixels || !depth_ok(bits)) return false;
    if (indexed ? (base != 1 || !depth_ok(cmap_bits) || !(bits == 8 || bits == 16) || cmap_len == 0) : !(base == 2 || base == 3)) return false;
    const int  value_bits = indexed ? cmap_bits : bits;             // what a colour value looks like
    const bool grey_alpha = !indexed && base == 3 && bits == 16;
    size_t     pos        = 18 + (size_t)idlen;
    const uint8_t* cmap   = nullptr;
    if (indexed)
    {
        pos += (size_t)cmap_skip;
        const size_t cmap_bytes = (size_t)cmap_len * ((cmap_bits + 7) / 8);
        if (pos + cmap_bytes > d.size()) return false;
        cmap = &d[pos];
        pos += cmap_bytes;
    }
    if (pos > d.size()) return false;
    const size_t px = (size_t)(bits + 7) / 8;
    // the file must be able to back its header before the image is allocated: all pixels when stored plainly, at least one packet
    // (a count byte and one pixel) per 128 pixels when run-length coded
    const uint64_t pixels = (uint64_t)W * H, left = d.size() - pos;
    if (rle ? ((pixels + 127) / 128) * (1 + px) > left : pixels * px > left) return false;
    Bytes raw((size_t)pixels * px);
    if (!rle)
        memcpy(raw.data(), &d[pos], raw.size());
    else
    {
        size_t o = 0;
        while (o < raw.size())
        {
            if (pos >= d.size()) return false;
            const int    hdr = d[pos++];
            const size_t n   = std::min<size_t>((hdr & 127) + 1, (raw.size() - o) / px);  // a last packet may run past the image
            if (hdr & 128)
            {
                if (pos + px > d.size()) return false;
                for (size_t k = 0; k < n; ++k, o += px) memcpy(&raw[o], &d[pos], px);
                pos += px;
            }
            else
            {
                if (pos + n * px > d.size()) return false;
                memcpy(&raw[o], &d[pos], n * px);
                pos += n * px, o += n * px;
            }
        }
    }
    auto expand = [&](const uint8_t* p, uint8_t* o) {  // one colour value -> RGBA
        if (value_bits == 8)
            o[0] = o[1] = o[2] = p[0], o[3] = 255;
        else if (grey_alpha)
            o[0] = o[1] = o[2] = p[0], o[3] = p[1];
        else if (value_bits == 24 || value_bits == 32)
            o[0] = p[2], o[1] = p[1], o[2] = p[0], o[3] = value_bits == 32 ? p[3] : 255;
        else  // 15 / 16: x RRRRR GGGGG BBBBB, little endian; the top bit is not alpha
        {
            const int v = p[0] | (p[1] << 8);
            o[0] = (uint8_t)(((v >> 10) & 31) * 255 / 31), o[1] = (uint8_t)(((v >> 5) & 31) * 255 / 31), o[2] = (uint8_t)((v & 31) * 255 / 31);
            o[3] = 255;
        }
    };
    rgba->assign((size_t)W * H * 4, 255);
    const bool top = (desc & 0x20) != 0;
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x)
        {
            const uint8_t* p = &raw[((size_t)y * W + x) * px];
            uint8_t*       o = &(*rgba)[4 * ((size_t)(top ? y : H - 1 - y) * W + x)];
            if (indexed)
            {
                int idx = bits == 8 ? p[0] : (p[0] | (p[1] << 8));
                if (idx >= cmap_len) idx = 0;
                expand(cmap + (size_t)idx * ((cmap_bits + 7) / 8), o);
            }
            else
                expand(p, o);
        }
    *w = W, *h = H;
    return true;
}

// ---------------------------------------------------------------- BMP
// stb's reading again (stb_image.h:5131-5474): headers of 12 / 40 / 56 / 108 / 124 bytes; 1 / 4 / 8 bits through a BGRx palette whose
// length follows from the data offset, 24 bits BGR, 16 / 32 bits through channel masks (defaults 5-5-5 and 8-8-8-8; the V4 / V5
// headers' own masks; BI_BITFIELDS behind a 40-byte header), each channel widened to eight bits by bit replication; run-length
// compression is refused; a 32-bit file whose alpha bytes are all zero is opaque; rows bottom-up unless the height is negative.
bool decode_bmp(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    if (d.size() < 26 || d[0] != 'B' || d[1] != 'M') return false;
    auto u16 = [&](size_t at) -> uint32_t { return at + 2 <= d.size() ? (uint32_t)(d[at] | (d[at + 1] << 8)) : 0u; };
    auto u32 = [&](size_t at) -> uint32_t { return at + 4 <= d.size() ? (uint32_t)d[at] | ((uint32_t)d[at + 1] << 8) | ((uint32_t)d[at + 2] << 16) | ((uint32_t)d[at + 3] << 24) : 0u; };
    const uint32_t offset = u32(10), hsz = u32(14);
    if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return false;
    if (d.size() < 14 + (size_t)hsz) return false;
    int32_t  W, H;
    uint32_t planes, bpp, compress = 0;
    if (hsz == 12)
        W = (int32_t)u16(18), H = (int32_t)u16(20), planes = u16(22), bpp = u16(24);
    else
        W = (int32_t)u32(18), H = (int32_t)u32(22), planes = u16(26), bpp = u16(28), compress = u32(30);
    if (planes != 1 || compress == 1 || compress == 2) return false;
    uint32_t mr = 0, mg = 0, mb = 0, ma = 0, extra = 14;
    bool     alpha_may_be_unused = false;  // default 32-bit masks: all-zero alpha means "no alpha"
    if (hsz == 40 || hsz == 56)
    {
        if (bpp == 16 || bpp == 32)
        {
            if (compress == 0)
            {
                if (bpp == 32) mr = 0xffu << 16, mg = 0xffu << 8, mb = 0xffu, ma = 0xffu << 24, alpha_may_be_unused = true;
                else mr = 31u << 10, mg = 31u << 5, mb = 31u;
            }
            else if (compress == 3)
            {
                const size_t at = 14 + (size_t)hsz;
                mr = u32(at), mg = u32(at + 4), mb = u32(at + 8), extra += 12;
                if (mr == mg && mg == mb) return false;
            }
            else
                return false;
        }
    }
    else if (hsz == 108 || hsz == 124)
        mr = u32(54), mg = u32(58), mb = u32(62), ma = u32(66);
    const bool flip = H > 0;
    if (H == INT32_MIN) return false;  // (its negation is not an int32)
    if (H < 0) H = -H;
    if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > kMaxPixels) return false;
    // Cheap refusal BEFORE the output is allocated: the pixel rows alone need this many bytes behind the headers, so a
    // short file that declares a huge extent costs nothing (the exact position checks follow where the layout is known)
    {
        const uint64_t row_bits = (uint64_t)W * (uint64_t)(bpp > 0 ? bpp : 1);
        if (((row_bits + 7) >> 3) * (uint64_t)H > (uint64_t)d.size()) return false;
    }
    int64_t psize = 0;
    if (hsz == 12)
    {
        if (bpp < 24) psize = ((int64_t)offset - extra - 24) / 3;
    }
    else if (bpp < 16)
        psize = ((int64_t)offset - extra - hsz) >> 2;
    uint32_t all_a = alpha_may_be_unused ? 0u : 255u;
    rgba->assign((size_t)W * H * 4, 255);
    size_t pos = (size_t)extra + hsz;
    if (bpp < 16)
    {
        if (psize <= 0 || psize > 256 || !(bpp == 1 || bpp == 4 || bpp == 8)) return false;
        const size_t entry = hsz == 12 ? 3 : 4;
        if (pos + (size_t)psize * entry > d.size()) return false;
        uint8_t pal[256][3];
        memset(pal, 0, sizeof(pal));
        for (int64_t i = 0; i < psize; ++i) pal[i][2] = d[pos + i * entry], pal[i][1] = d[pos + i * entry + 1], pal[i][0] = d[pos + i * entry + 2];
        const int64_t skip = (int64_t)offset - extra - hsz - psize * (int64_t)entry;
        if (skip < 0) return false;
        pos += (size_t)psize * entry + (size_t)skip;
        const size_t row = bpp == 1 ? ((size_t)W + 7) >> 3 : (bpp == 4 ? ((size_t)W + 1) >> 1 : (size_t)W), stride = (row + 3) & ~(size_t)3;
        if (pos > d.size() || (uint64_t)stride * (H - 1) + row > d.size() - pos) return false;
        for (int32_t y = 0; y < H; ++y)
        {
            const uint8_t* src = &d[pos + stride * (size_t)y];
            uint8_t*       o   = &(*rgba)[4 * (size_t)(flip ? H - 1 - y : y) * W];
            for (int32_t x = 0; x < W; ++x, o += 4)
            {
                const int v = bpp == 8 ? src[x] : (bpp == 4 ? (src[x >> 1] >> ((x & 1) ? 0 : 4)) & 15 : (src[x >> 3] >> (7 - (x & 7))) & 1);
                o[0] = pal[v][0], o[1] = pal[v][1], o[2] = pal[v][2];
            }
        }
    }
    else
    {
        if (!(bpp == 16 || bpp == 24 || bpp == 32)) return false;
        const int64_t skip = (int64_t)offset - extra - hsz;
        if (skip < 0) return false;
        pos += (size_t)skip;
        const int  easy = bpp == 24 ? 1 : ((bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) ? 2 : 0);
        auto high_bit = [](uint32_t z) { int n = -1; while (z) ++n, z >>= 1; return n; };
        auto bit_count = [](uint32_t z) { int n = 0; while (z) n += (int)(z & 1u), z >>= 1; return n; };
        int rs = 0, gs = 0, bs = 0, as = 0, rc = 0, gc = 0, bc = 0, ac = 0;
        if (!easy)
        {
            if (!mr || !mg || !mb) return false;
            rs = high_bit(mr) - 7, gs = high_bit(mg) - 7, bs = high_bit(mb) - 7, as = high_bit(ma) - 7;
            rc = bit_count(mr), gc = bit_count(mg), bc = bit_count(mb), ac = bit_count(ma);
            if (rc > 8 || gc > 8 || bc > 8 || ac > 8) return false;  // (stb's widening is undefined beyond eight bits)
        }
        // a channel of `bits` bits with its top bit moved to bit 7, widened to eight by replication (stb_image.h:5176-5195)
        auto widen = [](uint32_t v, int shift, int bits) -> uint32_t {
            static const uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01}, sh[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
            v = shift < 0 ? v << -shift : v >> shift;
            v >>= (8 - bits);
            return (v * mul[bits]) >> sh[bits];
        };
        const size_t px = bpp / 8, row = (size_t)W * px, stride = bpp == 32 ? row : (row + 3) & ~(size_t)3;
        if (pos > d.size() || (uint64_t)stride * (H - 1) + row > d.size() - pos) return false;
        for (int32_t y = 0; y < H; ++y)
        {
            const uint8_t* src = &d[pos + stride * (size_t)y];
            uint8_t*       o   = &(*rgba)[4 * (size_t)(flip ? H - 1 - y : y) * W];
            for (int32_t x = 0; x < W; ++x, o += 4, src += px)
            {
                uint32_t a = 255;
                if (easy)
                {
                    o[0] = src[2], o[1] = src[1], o[2] = src[0];
                    if (easy == 2) a = src[3];
                }
                else
                {
                    const uint32_t v = bpp == 16 ? (uint32_t)(src[0] | (src[1] << 8)) : ((uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24));
                    o[0] = (uint8_t)widen(v & mr, rs, rc), o[1] = (uint8_t)widen(v & mg, gs, gc), o[2] = (uint8_t)widen(v & mb, bs, bc);
                    if (ma) a = widen(v & ma, as, ac);
                }
                all_a |= a;
                o[3] = (uint8_t)a;
            }
        }
    }
    if (all_a == 0)
        for (size_t i = 3; i < rgba->size(); i += 4) (*rgba)[i] = 255;
    *w = (uint32_t)W, *h = (uint32_t)H;
    return true;
}

// ---------------------------------------------------------------- PNM (P5 / P6, one byte per sample)
bool decode_ppm(const Bytes& d, Bytes* rgba, uint32_t* w, uint32_t* h)
{
    if (d.size() < 2 || d[0] != 'P' || (d[1] != '6' && d[1] != '5')) return false;
    const size_t ch = d[1] == '6' ? 3 : 1;
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
    token();
    const long W = std::atol(token().c_str()), H = std::atol(token().c_str()), M = std::atol(token().c_str());
    // (a maximum below 255 is accepted and, as in stb, not rescaled)
    if (W <= 0 || H <= 0 || M < 1 || M > 255 || W > 32768 || H > 32768 || (uint64_t)W * (uint64_t)H > kMaxPixels) return false;
    ++pos;  // single whitespace after maxval
    if (d.size() < pos + (size_t)W * H * ch) return false;
    rgba->resize((size_t)W * H * 4);
    for (size_t i = 0; i < (size_t)W * H; ++i)
    {
        (*rgba)[4 * i + 0] = d[pos + ch * i];
        (*rgba)[4 * i + 1] = d[pos + ch * i + (ch == 3 ? 1 : 0)];
        (*rgba)[4 * i + 2] = d[pos + ch * i + (ch == 3 ? 2 : 0)];
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
    ok = cap::decode_jpeg(d.data(), d.size(), &rgba, &w, &h, kMaxPixels) || decode_png(d, &rgba, &w, &h) || decode_bmp(d, &rgba, &w, &h) || decode_ppm(d, &rgba, &w, &h);
    // TGA has no signature: tried last, on the strength of its header fields alone, as stb does (stb_image.h:1095-1099)
    if (!ok) ok = decode_tga(d, &rgba, &w, &h);
    (void)name_hint;
    }
    catch (const std::exception&)
    {
        cap_set_error_("cap_image_decode: out of memory");
        return CAP_ERR_IO;
    }
    if (!ok)
    {
        cap_set_error_("cap_image_decode: not a JPEG (Huffman-coded, 8-bit), PNG, BMP (not run-length coded), TGA or binary PNM this build decodes");
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
