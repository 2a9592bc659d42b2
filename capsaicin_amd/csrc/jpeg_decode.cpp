// jpeg_decode.cpp — JPEG (ITU-T T.81) texture decoding for cap_image_decode: Huffman-coded baseline / extended sequential (SOF0,
// SOF1) and progressive (SOF2) frames, 8 bits per sample, 1 / 3 / 4 components, any sampling factors up to 4, restart intervals.
//
// Entropy decoding follows the standard (Annexes F and G); what makes the output bit-identical to what the reference's
// TextureSystem receives is the arithmetic AFTER the coefficients, which the standard leaves open and which is restated here from
// the decoder the reference vendors and calls (stb_image.h v2.25, src/core/src/utils/stb_image.h; stbi_load(..., 4) at
// src/core/src/systems/texture_system.cpp:45):
//   * inverse DCT: 12-bit fixed-point constants, column pass rounded to 2 extra bits, row pass rounded at bit 17, +128, clamp
//     (stb_image.h:2311-2402);
//   * chroma upsampling: 2x horizontally / vertically / both by the (3, 1)/4 and (9, 3, 3, 1)/16 triangle filters with stb's
//     rounding and edge rules, nearest-neighbour for every other factor (stb_image.h:3312-3383, 3494-3503);
//   * YCbCr -> RGB: the 12-bit constants int(c * 4096 + 0.5), the Cb term of green floored on its own, result >> 4 after a bias
//     of 8 on a luma scaled by 16 (stb_image.h:3507-3537; its SSE2 form :3540-3600 computes the same integers);
//   * RGB pass-through when the component ids are 'R','G','B' or an Adobe APP14 marker says transform 0 without a JFIF header;
//     CMYK / YCCK through the (x * k + 128) * 257 >> 16 product (stb_image.h:3687-3760).
// oracle/_ref/libstb_ref.so (that header compiled as it lies in the reference tree) pins all of this in tests/test_image_ref.py.
// Not decoded, as in stb: arithmetic coding, lossless and hierarchical frames, 12-bit samples.  Unlike stb, a stream that is
// corrupt (bad Huffman code, missing tables, truncated scan without EOI) is an error here, not a partly grey image.
#include "image_decode.h"

#include <cstring>

namespace cap
{
namespace
{
constexpr uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable
{
    bool     defined = false;
    uint8_t  values[256];
    int32_t  maxcode[18];   // largest code of each length, -1 when the length is unused (T.81 F.2.2.3)
    int32_t  valptr[17];    // index of the first value of each length minus its smallest code
    uint16_t lookup[512];   // 9-bit prefix -> (length << 8) | value, 0 when the code is longer

    bool build(const uint8_t counts[16], const uint8_t* vals, int n)
    {
        int32_t code = 0, k = 0;
        memset(lookup, 0, sizeof(lookup));
        for (int len = 1; len <= 16; ++len)
        {
            valptr[len] = k - code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code)
            {
                if (code >= (1 << len)) return false;
                if (len <= 9)
                    for (int f = 0; f < (1 << (9 - len)); ++f) lookup[(code << (9 - len)) | f] = (uint16_t)((len << 8) | vals[k]);
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memcpy(values, vals, (size_t)n);
        defined = true;
        return true;
    }
};

// Bits of one entropy-coded segment, most significant first; FF 00 is a data byte FF, any other FF xx ends the segment (the
// reader then supplies zero bits and remembers where the marker is).
struct ScanBits
{
    const uint8_t* d;
    size_t         n, pos;
    uint64_t       acc  = 0;
    int            bits = 0;
    bool           at_marker = false;

    void fill()
    {
        while (bits <= 48)
        {
            uint32_t b = 0;
            if (!at_marker)
            {
                if (pos >= n)
                    at_marker = true;
                else if (d[pos] != 0xff)
                    b = d[pos++];
                else if (pos + 1 < n && d[pos + 1] == 0x00)
                    b = 0xff, pos += 2;
                else
                    at_marker = true;
            }
            acc |= (uint64_t)b << (56 - bits);
            bits += 8;
        }
    }
    uint32_t peek(int count)
    {
        if (bits < count) fill();
        return (uint32_t)(acc >> (64 - count));
    }
    void     skip(int count) { acc <<= count, bits -= count; }
    uint32_t get(int count)
    {
        if (count == 0) return 0;
        const uint32_t v = peek(count);
        skip(count);
        return v;
    }
    // byte alignment at the end of an interval: whatever is buffered is padding
    void drop() { acc = 0, bits = 0; }
};

struct Component
{
    int id = 0, h = 1, v = 1, tq = 0;
    int dc_table = 0, ac_table = 0;
    int width = 0, height = 0;        // samples the image really has in this component
    int blocks_w = 0, blocks_h = 0;   // blocks allocated (whole MCUs)
    int dc_pred = 0;
    std::vector<int16_t> coeff;       // 64 per block, natural (row-major) order
    std::vector<uint8_t> samples;     // blocks_w * 8 by blocks_h * 8 after the inverse DCT
};

struct Decoder
{
    const uint8_t* d;
    size_t         n, pos = 0;
    HuffTable      dc[4], ac[4];
    uint16_t       quant[4][64];
    bool           quant_defined[4] = {false, false, false, false};
    Component      comp[4];
    int            ncomp = 0, width = 0, height = 0, h_max = 1, v_max = 1, mcus_x = 0, mcus_y = 0;
    bool           progressive = false, have_frame = false, jfif = false;
    int            adobe_transform = -1, rgb_ids = 0;
    int            restart_interval = 0;
    // current scan
    int      scan_n = 0, order[4], ss = 0, se = 63, ah = 0, al = 0;
    uint32_t eob_run = 0;

    int  u8() { return pos < n ? d[pos++] : -1; }
    int  u16()
    {
        if (pos + 2 > n) return pos = n, -1;
        const int v = (d[pos] << 8) | d[pos + 1];
        pos += 2;
        return v;
    }

    // ---- tables and headers
    bool read_dqt()
    {
        int len = u16() - 2;
        while (len > 0)
        {
            const int q = u8();
            if (q < 0) return false;
            const int wide = q >> 4, t = q & 15;
            if (wide > 1 || t > 3) return false;
            for (int i = 0; i < 64; ++i)
            {
                const int v = wide ? u16() : u8();
                if (v < 0) return false;
                quant[t][kZigzag[i]] = (uint16_t)v;
            }
            quant_defined[t] = true;
            len -= wide ? 129 : 65;
        }
        return len == 0;
    }
    bool read_dht()
    {
        int len = u16() - 2;
        while (len > 0)
        {
            const int q = u8();
            if (q < 0) return false;
            const int cls = q >> 4, t = q & 15;
            if (cls > 1 || t > 3) return false;
            uint8_t counts[16], vals[256];
            int     total = 0;
            for (int i = 0; i < 16; ++i)
            {
                const int c = u8();
                if (c < 0) return false;
                counts[i] = (uint8_t)c, total += c;
            }
            if (total > 256 || pos + (size_t)total > n) return false;
            memcpy(vals, d + pos, (size_t)total);
            pos += (size_t)total;
            if (!(cls ? ac[t] : dc[t]).build(counts, vals, total)) return false;
            len -= 17 + total;
        }
        return len == 0;
    }
    bool read_app(int marker)
    {
        int len = u16();
        if (len < 2) return false;
        len -= 2;
        if (pos + (size_t)len > n) return false;
        const uint8_t* p = d + pos;
        if (marker == 0xe0 && len >= 5 && !memcmp(p, "JFIF\0", 5)) jfif = true;
        if (marker == 0xee && len >= 12 && !memcmp(p, "Adobe\0", 6)) adobe_transform = p[11];
        pos += (size_t)len;
        return true;
    }
    bool read_frame(uint64_t max_pixels)
    {
        const int len = u16(), precision = u8();
        height = u16(), width = u16(), ncomp = u8();
        if (len < 11 || precision != 8 || height <= 0 || width <= 0) return false;
        if (ncomp != 1 && ncomp != 3 && ncomp != 4) return false;
        if (len != 8 + 3 * ncomp || (uint64_t)width * (uint64_t)height > max_pixels) return false;
        rgb_ids = 0;
        for (int i = 0; i < ncomp; ++i)
        {
            Component& c = comp[i];
            c.id         = u8();
            const int hv = u8();
            c.tq         = u8();
            if (c.tq < 0 || c.tq > 3) return false;
            c.h = hv >> 4, c.v = hv & 15;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) return false;
            if (ncomp == 3 && c.id == "RGB"[i]) ++rgb_ids;
            h_max = c.h > h_max ? c.h : h_max, v_max = c.v > v_max ? c.v : v_max;
        }
        mcus_x = (width + 8 * h_max - 1) / (8 * h_max), mcus_y = (height + 8 * v_max - 1) / (8 * v_max);
        // Nothing is allocated for a header the file cannot back: every 8 x 8 block of every component is coded at least once, with
        // a Huffman code of at least one bit, so a file of n bytes holds at most 8 n blocks (a 100-byte file may not ask for 64 Mi pixels)
        uint64_t blocks = 0;
        for (int i = 0; i < ncomp; ++i)
            blocks += (uint64_t)(((width * comp[i].h + h_max - 1) / h_max + 7) / 8) * (uint64_t)(((height * comp[i].v + v_max - 1) / v_max + 7) / 8);
        if (blocks > 8ull * n) return false;
        for (int i = 0; i < ncomp; ++i)
        {
            Component& c = comp[i];
            // factors that do not divide the largest one (3 against 4) have no defined upsampling here or in stb: refused
            if (h_max % c.h != 0 || v_max % c.v != 0) return false;
            c.width      = (width * c.h + h_max - 1) / h_max;
            c.height     = (height * c.v + v_max - 1) / v_max;
            c.blocks_w = mcus_x * c.h, c.blocks_h = mcus_y * c.v;
            c.coeff.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        have_frame = true;
        return true;
    }
    bool read_scan_header()
    {
        const int len = u16();
        scan_n        = u8();
        if (scan_n < 1 || scan_n > 4 || scan_n > ncomp || len != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; ++i)
        {
            const int id = u8(), tabs = u8();
            int       which = 0;
            while (which < ncomp && comp[which].id != id) ++which;
            if (which == ncomp || tabs < 0) return false;
            comp[which].dc_table = tabs >> 4, comp[which].ac_table = tabs & 15;
            if (comp[which].dc_table > 3 || comp[which].ac_table > 3) return false;
            order[i] = which;
        }
        ss = u8(), se = u8();
        const int a = u8();
        if (a < 0) return false;
        ah = a >> 4, al = a & 15;
        if (progressive)
        {
            if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) return false;
            if (ss == 0 && se != 0) return false;   // DC and AC coefficients never share a progressive scan
            if (ss != 0 && scan_n != 1) return false;
        }
        else
        {
            if (ss != 0 || ah != 0 || al != 0) return false;
            se = 63;
        }
        return true;
    }

    // ---- entropy decoding
    static int huff(ScanBits& br, const HuffTable& t)
    {
        const uint32_t look = br.peek(16);
        const uint16_t fast = t.lookup[look >> 7];
        if (fast)
        {
            br.skip(fast >> 8);
            return fast & 0xff;
        }
        for (int len = 10; len <= 16; ++len)
        {
            const int32_t code = (int32_t)(look >> (16 - len));
            if (code <= t.maxcode[len])
            {
                br.skip(len);
                return t.values[(t.valptr[len] + code) & 0xff];
            }
        }
        return -1;
    }
    // predictions of a damaged stream may run away: they wrap instead of overflowing
    static int wrap_add(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
    // T.81 F.2.2.1 EXTEND(RECEIVE(s), s)
    static int receive_extend(ScanBits& br, int s)
    {
        if (s == 0) return 0;
        const int v = (int)br.get(s);
        return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }

    bool block_sequential(ScanBits& br, Component& c, int16_t* blk)
    {
        const HuffTable &td = dc[c.dc_table], &ta = ac[c.ac_table];
        const uint16_t*  q  = quant[c.tq];
        if (!td.defined || !ta.defined) return false;
        const int s = huff(br, td);
        if (s < 0 || s > 16) return false;
        c.dc_pred = wrap_add(c.dc_pred, receive_extend(br, s));
        blk[0] = (int16_t)((uint32_t)c.dc_pred * q[0]);   // sequential frames are dequantised as they are read (tables of that moment)
        for (int k = 1; k < 64;)
        {
            const int rs = huff(br, ta);
            if (rs < 0) return false;
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0)
            {
                if (rs != 0xf0) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return false;
            const int z = kZigzag[k++];
            blk[z]      = (int16_t)((uint32_t)receive_extend(br, sz) * q[z]);
        }
        return true;
    }
    bool block_dc_progressive(ScanBits& br, Component& c, int16_t* blk)
    {
        if (ah == 0)
        {
            const HuffTable& td = dc[c.dc_table];
            if (!td.defined) return false;
            const int s = huff(br, td);
            if (s < 0 || s > 16) return false;
            c.dc_pred = wrap_add(c.dc_pred, receive_extend(br, s));
            blk[0] = (int16_t)((uint32_t)c.dc_pred << al);
        }
        else if (br.get(1))
            blk[0] = (int16_t)(blk[0] + (1 << al));
        return true;
    }
    // one already non-zero coefficient meets a correction bit (T.81 G.1.2.3)
    void refine(ScanBits& br, int16_t& v, int bit)
    {
        if (br.get(1) && (v & bit) == 0) v = (int16_t)(v > 0 ? v + bit : v - bit);
    }
    bool block_ac_progressive(ScanBits& br, Component& c, int16_t* blk)
    {
        const HuffTable& ta = ac[c.ac_table];
        if (!ta.defined) return false;
        if (ah == 0)
        {
            if (eob_run)
            {
                --eob_run;
                return true;
            }
            for (int k = ss; k <= se;)
            {
                const int rs = huff(br, ta);
                if (rs < 0) return false;
                const int r = rs >> 4, sz = rs & 15;
                if (sz == 0)
                {
                    if (r < 15)
                    {
                        eob_run = (1u << r) + br.get(r) - 1u;
                        break;
                    }
                    k += 16;
                    continue;
                }
                k += r;
                if (k > 63) return false;
                blk[kZigzag[k++]] = (int16_t)(receive_extend(br, sz) * (1 << al));
            }
            return true;
        }
        const int bit = 1 << al;
        int       k   = ss;
        if (!eob_run)
        {
            while (k <= se)
            {
                const int rs = huff(br, ta);
                if (rs < 0) return false;
                int       r = rs >> 4;
                const int sz = rs & 15;
                int       fresh = 0;
                if (sz == 0)
                {
                    if (r < 15)
                    {
                        eob_run = (1u << r) + br.get(r);   // this block is the first of the run
                        break;
                    }
                }
                else
                {
                    if (sz != 1) return false;
                    fresh = br.get(1) ? bit : -bit;
                }
                // pass r zero-history coefficients, refining the non-zero ones on the way, then place the new one
                for (; k <= se; ++k)
                {
                    int16_t& v = blk[kZigzag[k]];
                    if (v != 0)
                        refine(br, v, bit);
                    else if (r-- == 0)
                    {
                        v = (int16_t)fresh;
                        ++k;
                        break;
                    }
                }
            }
        }
        if (eob_run)
        {
            for (; k <= se; ++k)
            {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0) refine(br, v, bit);
            }
            --eob_run;
        }
        return true;
    }

    bool decode_block(ScanBits& br, Component& c, int bx, int by)
    {
        int16_t* blk = &c.coeff[64 * ((size_t)by * c.blocks_w + bx)];
        if (!progressive) return block_sequential(br, c, blk);
        return ss == 0 ? block_dc_progressive(br, c, blk) : block_ac_progressive(br, c, blk);
    }

    // An interval has ended: the coded bits stop at a byte boundary and RSTn follows.  false = no restart marker there (the scan
    // is over, as for stb).
    bool restart(ScanBits& br)
    {
        br.drop();
        size_t p = br.pos;
        while (p + 1 < n && !(d[p] == 0xff && d[p + 1] != 0x00 && d[p + 1] != 0xff)) ++p;
        if (p + 1 >= n || d[p + 1] < 0xd0 || d[p + 1] > 0xd7)
        {
            br.pos = p, br.at_marker = true;
            return false;
        }
        br.pos = p + 2, br.at_marker = false;
        for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
        eob_run = 0;
        return true;
    }

    bool read_scan()
    {
        if (!read_scan_header()) return false;
        ScanBits br{d, n, pos};
        for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
        eob_run       = 0;
        int  todo     = restart_interval ? restart_interval : 0x7fffffff;
        bool more     = true;
        if (scan_n == 1)
        {
            // a scan of one component walks that component's own blocks, row by row, without MCU padding (T.81 A.2.2)
            Component& c  = comp[order[0]];
            const int  bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < bh && more; ++by)
                for (int bx = 0; bx < bw && more; ++bx)
                {
                    if (!decode_block(br, c, bx, by)) return false;
                    if (--todo <= 0) more = restart(br), todo = restart_interval;
                }
        }
        else
        {
            for (int my = 0; my < mcus_y && more; ++my)
                for (int mx = 0; mx < mcus_x && more; ++mx)
                {
                    for (int k = 0; k < scan_n; ++k)
                    {
                        Component& c = comp[order[k]];
                        for (int y = 0; y < c.v; ++y)
                            for (int x = 0; x < c.h; ++x)
                                if (!decode_block(br, c, mx * c.h + x, my * c.v + y)) return false;
                    }
                    if (--todo <= 0) more = restart(br), todo = restart_interval;
                }
        }
        // continue at the marker that ended the coded data
        size_t p = br.pos;
        while (p + 1 < n && !(d[p] == 0xff && d[p + 1] != 0x00 && d[p + 1] != 0xff)) ++p;
        pos = p;
        return true;
    }

    // ---- reconstruction
    // 1-D pass shared by columns and rows: even part in x[4], odd part in t[4] (scaled by 4096)
    // All sums and products of the inverse DCT wrap modulo 2^32 (unsigned arithmetic): identical to the plain int form on any
    // real image, and defined behaviour on the coefficient garbage a damaged file can produce.
    typedef uint32_t U;
    static int sar(U v, int s) { return (int32_t)v >> s; }
    static void idct_1d(U s0, U s1, U s2, U s3, U s4, U s5, U s6, U s7, U x[4], U t[4])
    {
        auto fix = [](double c) { return (U)(int)(c * 4096 + 0.5); };
        const U z  = (s2 + s6) * fix(0.5411961f);
        const U e2 = z + s6 * fix(-1.847759065f), e3 = z + s2 * fix(0.765366865f);
        const U e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
        x[0] = e0 + e3, x[3] = e0 - e3, x[1] = e1 + e2, x[2] = e1 - e2;
        const U a = s7 + s3, b = s5 + s1, c = s7 + s1, dd = s5 + s3;
        const U w = (a + b) * fix(1.175875602f);
        const U pc = w + c * fix(-0.899976223f), pd = w + dd * fix(-2.562915447f);
        const U pa = a * fix(-1.961570560f), pb = b * fix(-0.390180644f);
        t[3] = s1 * fix(1.501321110f) + pc + pb;
        t[2] = s3 * fix(3.072711026f) + pd + pa;
        t[1] = s5 * fix(2.053119869f) + pd + pb;
        t[0] = s7 * fix(0.298631336f) + pc + pa;
    }
    static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void idct_block(const int16_t* blk, uint8_t* out, size_t stride)
    {
        U mid[64], x[4], t[4];
        auto c = [&](int k) { return (U)(int)blk[k]; };
        for (int i = 0; i < 8; ++i)
        {
            idct_1d(c(i), c(8 + i), c(16 + i), c(24 + i), c(32 + i), c(40 + i), c(48 + i), c(56 + i), x, t);
            for (int k = 0; k < 4; ++k)
            {
                mid[8 * k + i]       = (U)sar(x[k] + 512u + t[3 - k], 10);
                mid[8 * (7 - k) + i] = (U)sar(x[k] + 512u - t[3 - k], 10);
            }
        }
        const U bias = 65536u + (128u << 17);
        for (int i = 0; i < 8; ++i)
        {
            const U* m = mid + 8 * i;
            idct_1d(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], x, t);
            uint8_t* o = out + stride * i;
            for (int k = 0; k < 4; ++k)
            {
                o[k]     = clamp8(sar(x[k] + bias + t[3 - k], 17));
                o[7 - k] = clamp8(sar(x[k] + bias - t[3 - k], 17));
            }
        }
    }
    void reconstruct()
    {
        for (int i = 0; i < ncomp; ++i)
        {
            Component&   c      = comp[i];
            const size_t stride = (size_t)c.blocks_w * 8;
            c.samples.assign(stride * c.blocks_h * 8, 0);
            const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx)
                {
                    int16_t* blk = &c.coeff[64 * ((size_t)by * c.blocks_w + bx)];
                    // progressive frames are dequantised at the end with the tables of the end; only the blocks the image
                    // covers (stb_image.h:2973-2987), the MCU padding stays as decoded
                    if (progressive && bx < bw && by < bh)
                        for (int k = 0; k < 64; ++k) blk[k] = (int16_t)(blk[k] * quant[c.tq][k]);
                    if (progressive && !(bx < bw && by < bh)) continue;
                    idct_block(blk, &c.samples[stride * 8 * by + 8 * bx], stride);
                }
        }
    }

    // one output row of a component at full resolution (stb_image.h:3312-3383, 3494-3503)
    static void upsample_row(const uint8_t* near, const uint8_t* far, int w, int hs, int vs, uint8_t* out)
    {
        if (hs == 1 && vs == 1)
            memcpy(out, near, (size_t)w);
        else if (hs == 1 && vs == 2)
            for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        else if (hs == 2 && vs == 1)
        {
            if (w == 1)
            {
                out[0] = out[1] = near[0];
                return;
            }
            out[0] = near[0];
            for (int i = 0; i + 1 < w; ++i)
            {
                out[2 * i + 1] = (uint8_t)((3 * near[i] + near[i + 1] + 2) >> 2);
                out[2 * i + 2] = (uint8_t)((3 * near[i + 1] + near[i] + 2) >> 2);
            }
            // the last source sample's left output takes the weights of its neighbour's right one (stb_image.h:3353: 3 * in[w - 2]
            // + in[w - 1]), and so does every texture the reference has ever loaded
            out[2 * w - 2] = (uint8_t)((3 * near[w - 2] + near[w - 1] + 2) >> 2);
            out[2 * w - 1] = near[w - 1];
        }
        else if (hs == 2 && vs == 2)
        {
            int cur = 3 * near[0] + far[0];
            out[0]  = (uint8_t)((cur + 2) >> 2);
            for (int i = 1; i < w; ++i)
            {
                const int prev = cur;
                cur            = 3 * near[i] + far[i];
                out[2 * i - 1] = (uint8_t)((3 * prev + cur + 8) >> 4);
                out[2 * i]     = (uint8_t)((3 * cur + prev + 8) >> 4);
            }
            out[2 * w - 1] = (uint8_t)((cur + 2) >> 2);
        }
        else
            for (int i = 0; i < w; ++i)
                for (int j = 0; j < hs; ++j) out[i * hs + j] = near[i];
    }

    static int floor_shift(int v, int s) { return v >= 0 ? v >> s : -((-v + (1 << s) - 1) >> s); }
    static uint8_t blend(uint8_t x, uint8_t k)
    {
        const unsigned t = (unsigned)x * k + 128u;
        return (uint8_t)((t + (t >> 8)) >> 8);
    }

    void emit(std::vector<uint8_t>* rgba)
    {
        rgba->assign((size_t)width * height * 4, 255);
        const bool is_rgb = ncomp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
        // row state per component: line0 / line1 are sample rows, step counts the output rows inside one source row
        struct RowState
        {
            int hs, vs, step, ypos, line0, line1, w_lores;
        } rs[4];
        std::vector<uint8_t> line[4];
        for (int k = 0; k < ncomp; ++k)
        {
            rs[k].hs = h_max / comp[k].h, rs[k].vs = v_max / comp[k].v;
            rs[k].step = rs[k].vs >> 1, rs[k].ypos = 0, rs[k].line0 = rs[k].line1 = 0;
            rs[k].w_lores = (width + rs[k].hs - 1) / rs[k].hs;
            line[k].assign((size_t)rs[k].w_lores * rs[k].hs + 4, 0);
        }
        auto fix = [](float c) { return (int)(c * 4096.0f + 0.5f); };
        const int cr_r = fix(1.40200f), cr_g = -fix(0.71414f), cb_g = -fix(0.34414f), cb_b = fix(1.77200f);
        for (int y = 0; y < height; ++y)
        {
            for (int k = 0; k < ncomp; ++k)
            {
                RowState&      r      = rs[k];
                const size_t   stride = (size_t)comp[k].blocks_w * 8;
                const bool     lower  = r.step >= (r.vs >> 1);
                const uint8_t* l0     = &comp[k].samples[stride * r.line0];
                const uint8_t* l1     = &comp[k].samples[stride * r.line1];
                upsample_row(lower ? l1 : l0, lower ? l0 : l1, r.w_lores, r.hs, r.vs, line[k].data());
                if (++r.step >= r.vs)
                {
                    r.step = 0, r.line0 = r.line1;
                    if (++r.ypos < comp[k].height) ++r.line1;
                }
            }
            uint8_t* o = &(*rgba)[(size_t)y * width * 4];
            for (int x = 0; x < width; ++x, o += 4)
            {
                if (ncomp == 1)
                    o[0] = o[1] = o[2] = line[0][x];
                else if (is_rgb || (ncomp == 4 && adobe_transform == 0))
                    o[0] = line[0][x], o[1] = line[1][x], o[2] = line[2][x];
                else
                {
                    const int luma = (line[0][x] << 4) + 8, cb = line[1][x] - 128, cr = line[2][x] - 128;
                    o[0] = clamp8(floor_shift(luma + floor_shift(cr_r * cr, 8), 4));
                    o[1] = clamp8(floor_shift(luma + floor_shift(cr_g * cr, 8) + floor_shift(cb_g * cb, 8), 4));
                    o[2] = clamp8(floor_shift(luma + floor_shift(cb_b * cb, 8), 4));
                }
                if (ncomp == 4 && adobe_transform == 0)
                    for (int k = 0; k < 3; ++k) o[k] = blend(o[k], line[3][x]);
                else if (ncomp == 4 && adobe_transform == 2)
                    for (int k = 0; k < 3; ++k) o[k] = blend((uint8_t)(255 - o[k]), line[3][x]);
            }
        }
    }

    bool run(std::vector<uint8_t>* rgba, uint32_t* w, uint32_t* h, uint64_t max_pixels)
    {
        if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return false;
        pos = 2;
        for (;;)
        {
            // next marker: FF, optional FF fill bytes, code
            int m = u8();
            if (m != 0xff) return false;
            while (m == 0xff) m = u8();
            if (m < 0) return false;
            if (m == 0xd9) break;
            bool ok;
            if (m == 0xdb)
                ok = read_dqt();
            else if (m == 0xc4)
                ok = read_dht();
            else if (m == 0xdd)
                ok = u16() == 4 && (restart_interval = u16()) >= 0;
            else if ((m >= 0xe0 && m <= 0xef) || m == 0xfe)
                ok = read_app(m);
            else if (m == 0xc0 || m == 0xc1 || m == 0xc2)
                ok = !have_frame && (progressive = m == 0xc2, read_frame(max_pixels));
            else if (m == 0xda)
                ok = have_frame && read_scan();
            else if (m == 0xdc)
                ok = u16() == 4 && u16() == height;
            else
                ok = false;   // arithmetic / lossless / hierarchical frames and anything unknown
            if (!ok) return false;
        }
        if (!have_frame) return false;
        for (int i = 0; i < ncomp; ++i)
            if (!quant_defined[comp[i].tq]) return false;
        reconstruct();
        emit(rgba);
        *w = (uint32_t)width, *h = (uint32_t)height;
        return true;
    }
};
}  // namespace

bool decode_jpeg(const uint8_t* data, size_t size, std::vector<uint8_t>* rgba, uint32_t* w, uint32_t* h, uint64_t max_pixels)
{
    Decoder dec;
    dec.d = data, dec.n = size;
    memset(dec.quant, 0, sizeof(dec.quant));
    return dec.run(rgba, w, h, max_pixels);
}
}  // namespace cap
