"""Test helper: writes BMP files field by field -- every header size stb reads (12 / 40 / 56 / 108 / 124 bytes), palettes of 1 / 4 / 8
bits, 16 / 24 / 32 bits with default or explicit channel masks, top-down rows."""
import struct

import numpy as np


def write(rs, w, h, bpp, header=40, masks=None, top_down=False, palette_entries=None, compression=None):
    """Random pixel content.  masks: (r, g, b[, a]) for BI_BITFIELDS (40 / 56-byte headers: three masks after the header; 108 / 124:
    four in the header)."""
    rowbytes = (w * bpp + 7) // 8
    stride = (rowbytes + 3) & ~3
    rows = []
    for _ in range(h):
        if bpp < 8:
            n = 1 << bpp if palette_entries is None else palette_entries
            vals = rs.randint(0, n, w)
            bits = np.unpackbits(vals.astype(np.uint8)[:, None], axis=1)[:, 8 - bpp:].reshape(-1)
            row = np.packbits(bits).tobytes()
        elif bpp == 8:
            row = rs.randint(0, 256 if palette_entries is None else palette_entries, w).astype(np.uint8).tobytes()
        else:
            row = rs.randint(0, 256, rowbytes).astype(np.uint8).tobytes()
        rows.append(row + bytes(stride - len(row)))
    palette = b""
    if bpp <= 8:
        n = (1 << bpp) if palette_entries is None else palette_entries
        entry = 3 if header == 12 else 4
        palette = rs.randint(0, 256, n * entry).astype(np.uint8).tobytes()
    comp = compression if compression is not None else (3 if masks is not None else 0)
    if header == 12:
        hdr = struct.pack("<IHHHH", 12, w, h, 1, bpp)
    else:
        hdr = struct.pack("<IiiHHIIiiII", header, w, -h if top_down else h, 1, bpp, comp, stride * h, 2835, 2835, 0, 0)
        if header == 56:
            m = list(masks or (0, 0, 0)) + [0] * 4
            hdr += struct.pack("<IIII", *m[:4])
        elif header in (108, 124):
            m = list(masks or (0, 0, 0, 0)) + [0] * 4
            hdr += struct.pack("<IIII", *m[:4]) + b"BGRs" + bytes(48)
            if header == 124:
                hdr += bytes(16)
        assert len(hdr) == header, (len(hdr), header)
    after = b""
    if header == 40 and masks is not None:
        after = struct.pack("<III", *masks[:3])
    offset = 14 + len(hdr) + len(after) + len(palette)
    body = b"".join(rows)
    return b"BM" + struct.pack("<IHHI", offset + len(body), 0, 0, offset) + hdr + after + palette + body
