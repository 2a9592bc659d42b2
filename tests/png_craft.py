"""Test helper: writes PNG files from sample arrays with full control over what PIL does not expose -- Adam7 interlacing, the
filter type of every scanline, bit depths 1 / 2 / 4 / 16, PLTE / tRNS chunks, IDAT split over several chunks."""
import struct
import zlib

import numpy as np

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def _pack(rows, depth):
    """(h, w * channels) sample values -> list of packed scanlines."""
    if depth == 16:
        return [r.astype(">u2").tobytes() for r in rows]
    if depth == 8:
        return [r.astype(np.uint8).tobytes() for r in rows]
    out = []
    for r in rows:
        bits = np.unpackbits(r.astype(np.uint8)[:, None], axis=1)[:, 8 - depth:].reshape(-1)
        out.append(np.packbits(bits).tobytes())
    return out


def _filter(line, prev, bpp, ftype):
    cur, up = np.frombuffer(line, np.uint8).astype(int), np.frombuffer(prev, np.uint8).astype(int)
    left = np.concatenate([np.zeros(bpp, int), cur[:-bpp]]) if len(cur) > bpp else np.zeros(len(cur), int)
    upleft = np.concatenate([np.zeros(bpp, int), up[:-bpp]]) if len(cur) > bpp else np.zeros(len(cur), int)
    if ftype == 0:
        pred = 0
    elif ftype == 1:
        pred = left
    elif ftype == 2:
        pred = up
    elif ftype == 3:
        pred = (left + up) >> 1
    else:
        p = left + up - upleft
        pa, pb, pc = abs(p - left), abs(p - up), abs(p - upleft)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
    return bytes([ftype]) + ((cur - pred) & 0xff).astype(np.uint8).tobytes()


def write(samples, depth, ctype, interlace=False, plte=None, trns=None, rs=None, idat_pieces=1, level=6):
    """samples: (h, w, channels) integers below 2**depth; rs: RandomState choosing a filter type per scanline (None: type 0)."""
    h, w, ch = samples.shape
    assert ch == CHANNELS[ctype]
    bpp = max(1, ch * depth // 8)
    stream = b""
    for x0, y0, dx, dy in (ADAM7 if interlace else [(0, 0, 1, 1)]):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        lines = _pack(sub.reshape(sub.shape[0], -1), depth)
        prev = bytes(len(lines[0]))
        for line in lines:
            stream += _filter(line, prev, bpp, 0 if rs is None else int(rs.randint(0, 5)))
            prev = line
    z = zlib.compress(stream, level)
    cuts = [len(z) * i // idat_pieces for i in range(idat_pieces + 1)]
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", bytes(plte))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    for a, b in zip(cuts[:-1], cuts[1:]):
        out += chunk(b"IDAT", z[a:b])
    return out + chunk(b"IEND", b"")


def random_file(rs, w, h, depth, ctype, interlace, **kw):
    samples = rs.randint(0, 1 << depth, (h, w, CHANNELS[ctype]))
    if ctype == 3:
        kw.setdefault("plte", rs.randint(0, 256, 3 * (1 << depth)).astype(np.uint8).tobytes())
    return write(samples, depth, ctype, interlace, rs=rs, **kw)
