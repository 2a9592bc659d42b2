"""Test helper: writes sequential (SOF0) JPEG streams straight from quantised coefficient blocks, so that tests can reach layouts
PIL / libjpeg never emit: vertical-only and 4x subsampling, one-component scans in baseline files, 16-bit quantisation tables,
tables redefined between scans, restart intervals of any length, fill bytes, DNL, comment segments.  No DCT is involved: a decoder
is judged on coefficients -> pixels, and random sparse coefficients exercise that directly.
"""
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
          42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def canonical(counts, symbols):
    """{symbol: (code, length)} of a table given as T.81 BITS / HUFFVAL."""
    assert sum(counts) == len(symbols) and sum(c / 2.0 ** (i + 1) for i, c in enumerate(counts)) < 1.0
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[symbols[k]] = (code, length)
            code, k = code + 1, k + 1
        code <<= 1
    return table


# DC: categories 0..11, a skewed code; AC: the 162 run/size symbols of 8-bit data, code lengths 2..16 (so that both the short
# look-up and the long canonical path of a decoder are taken)
DC_COUNTS = [0, 1, 2, 2, 2, 2, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]
DC_SYMBOLS = [2, 0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]
AC_COUNTS = [0, 1, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 25]
AC_SYMBOLS = [0x01, 0x00, 0x02, 0x11, 0x03, 0x12, 0x21, 0xf0] + [s for s in ((r << 4) | z for z in range(1, 11) for r in range(16))
                                                               if s not in (0x01, 0x02, 0x11, 0x03, 0x12, 0x21)]


class BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xff
            self.out.append(b)
            if b == 0xff:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)  # pad with ones
        data, self.out = bytes(self.out), bytearray()
        return data


def _magnitude(v):
    size = int(abs(v)).bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def _segment(marker, body):
    return bytes([0xff, marker]) + struct.pack(">H", len(body) + 2) + body


def _dht(cls, ident, counts, symbols):
    return _segment(0xc4, bytes([(cls << 4) | ident]) + bytes(counts) + bytes(symbols))


def _dqt(ident, table, wide):
    z = [int(table[ZIGZAG[i]]) for i in range(64)]
    return _segment(0xdb, bytes([(16 if wide else 0) | ident]) + (struct.pack(">64H", *z) if wide else bytes(z)))


def random_blocks(rs, nblocks, density=0.15, amplitude=40, dc_range=60):
    """Sparse random quantised coefficients, natural order, (nblocks, 64) int."""
    c = np.where(rs.rand(nblocks, 64) < density, rs.randint(-amplitude, amplitude + 1, (nblocks, 64)), 0)
    c[:, 0] = rs.randint(-dc_range, dc_range + 1, nblocks)
    c[rs.rand(nblocks) < 0.1, 1:] = 0   # some DC-only blocks
    long_run = rs.rand(nblocks) < 0.1  # some blocks whose only AC coefficient sits behind runs of 16 zeros
    c[long_run, 1:] = 0
    c[long_run, ZIGZAG[50]] = 3
    return c


def encode(width, height, sampling, coeffs, quant, *, interleaved=True, restart=0, wide_quant=False, component_ids=None,
           fill_bytes=False, comment=False, dnl=False, adobe_transform=None, jfif=True, requant_between_scans=None):
    """sampling: [(h, v)] per component; coeffs[i]: (blocks_y, blocks_x, 64) for component i over whole MCUs; quant[i]: 64 ints."""
    ncomp = len(sampling)
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    ids = component_ids or list(range(1, ncomp + 1))
    dc_t, ac_t = canonical(DC_COUNTS, DC_SYMBOLS), canonical(AC_COUNTS, AC_SYMBOLS)
    out = bytearray(b"\xff\xd8")
    if jfif:
        out += _segment(0xe0, b"JFIF\0\1\1\0\0\1\0\1\0\0")
    if adobe_transform is not None:
        out += _segment(0xee, b"Adobe\0" + struct.pack(">HHHB", 100, 0, 0, adobe_transform))
    if comment:
        out += _segment(0xfe, b"made by tests/jpeg_craft.py")
    for i in range(ncomp):
        out += _dqt(i, quant[i], wide_quant)
    frame = struct.pack(">BHHB", 8, height, width, ncomp)
    for i, (h, v) in enumerate(sampling):
        frame += bytes([ids[i], (h << 4) | v, i])
    out += _segment(0xc0, frame)
    out += _dht(0, 0, DC_COUNTS, DC_SYMBOLS) + _dht(1, 0, AC_COUNTS, AC_SYMBOLS)
    if restart:
        out += _segment(0xdd, struct.pack(">H", restart))

    def put_block(bw, blk, pred):
        size, bits = _magnitude(int(blk[0]) - pred)
        bw.put(*dc_t[size])
        if size:
            bw.put(bits, size)
        run = 0
        for k in range(1, 64):
            v = int(blk[ZIGZAG[k]])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bw.put(*ac_t[0xf0])
                run -= 16
            size, bits = _magnitude(v)
            bw.put(*ac_t[(run << 4) | size])
            bw.put(bits, size)
            run = 0
        if run:
            bw.put(*ac_t[0x00])
        return int(blk[0])

    def scan(components):
        nonlocal out
        if fill_bytes:
            out += b"\xff\xff\xff"
        hdr = bytes([len(components)]) + b"".join(bytes([ids[i], 0x00]) for i in components) + bytes([0, 63, 0])
        out += _segment(0xda, hdr)
        bw, pred, count, rst = BitWriter(), [0] * ncomp, 0, 0

        def mcu_done():
            nonlocal count, rst, pred
            count += 1
            if restart and count % restart == 0:
                out.extend(bw.flush())
                out.extend(bytes([0xff, 0xd0 + rst]))
                rst, pred = (rst + 1) & 7, [0] * ncomp
        if len(components) == 1:
            i = components[0]
            h, v = sampling[i]
            bw_, bh_ = -(-(-(-width * h // hmax)) // 8), -(-(-(-height * v // vmax)) // 8)
            for by in range(bh_):
                for bx in range(bw_):
                    pred[i] = put_block(bw, coeffs[i][by, bx], pred[i])
                    mcu_done()
        else:
            for yy in range(my):
                for xx in range(mx):
                    for i in components:
                        h, v = sampling[i]
                        for y in range(v):
                            for x in range(h):
                                pred[i] = put_block(bw, coeffs[i][yy * v + y, xx * h + x], pred[i])
                    mcu_done()
        out.extend(bw.flush())
        # a restart marker that closes the very last interval is not written by real encoders: drop it
        if restart and count % restart == 0 and out[-2] == 0xff and 0xd0 <= out[-1] <= 0xd7:
            del out[-2:]

    if interleaved or ncomp == 1:
        scan(list(range(ncomp)))
    else:
        for i in range(ncomp):
            scan([i])
            # a table replaced AFTER the scan that used it: a sequential frame is dequantised block by block, so nothing changes
            if requant_between_scans is not None:
                out += _dqt(i, requant_between_scans[i], wide_quant)
    if dnl:
        out += _segment(0xdc, struct.pack(">H", height))
    out += b"\xff\xd9"
    return bytes(out)


def random_file(rs, width, height, sampling, **kw):
    hmax, vmax = max(h for h, _ in sampling), max(v for _, v in sampling)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    # sample values stay inside the range where 16-bit intermediates (stb_image.h:2409-2585, the SSE2 inverse DCT) cannot saturate
    wide = kw.get("wide_quant", False)
    amplitude, dc_range, q_lo, q_hi = (2, 4, 256, 400) if wide else (40, 60, 1, 40)
    coeffs = [random_blocks(rs, my * v * mx * h, amplitude=amplitude, dc_range=dc_range).reshape(my * v, mx * h, 64) for h, v in sampling]
    quant = [rs.randint(q_lo, q_hi, 64) for _ in sampling]
    if kw.pop("requant", False):
        kw["requant_between_scans"] = [rs.randint(q_lo, q_hi, 64) for _ in sampling]
    return encode(width, height, sampling, coeffs, quant, **kw)
