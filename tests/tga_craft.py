"""Test helper: writes TGA files field by field (PIL writes only the common ones): any image type / bits per pixel / colour-map
depth / descriptor byte, 16-bit indices, an image id, run-length packets that cross scanlines."""
import struct

import numpy as np


def _rle(pixels, nbytes, rs):
    """Random mix of run and raw packets over a bytearray of pixels; the pixels a run packet covers are made equal."""
    out, i, n = bytearray(), 0, len(pixels) // nbytes
    while i < n:
        run = int(min(n - i, rs.randint(1, 129)))
        if rs.rand() < 0.5:
            out.append(0x80 | (run - 1))
            out += pixels[i * nbytes:(i + 1) * nbytes]
            pixels[i * nbytes:(i + run) * nbytes] = pixels[i * nbytes:(i + 1) * nbytes] * run
        else:
            out.append(run - 1)
            out += pixels[i * nbytes:(i + run) * nbytes]
        i += run
    return bytes(out)


def random_file(rs, w, h, image_type, bpp, descriptor=0, cmap_bits=0, cmap_len=0, cmap_first=0, id_len=0, rle=False):
    hdr = struct.pack("<BBBHHBHHHHBB", id_len, 1 if cmap_bits else 0, image_type | (8 if rle else 0), cmap_first, cmap_len, cmap_bits,
                      0, 0, w, h, bpp, descriptor)
    body = bytes(rs.randint(0, 256, id_len).astype(np.uint8))
    # stb skips `cmap_first` BYTES in front of the colour map: give it bytes to skip, so that nothing runs off the end of the file
    body += bytes(rs.randint(0, 256, cmap_first).astype(np.uint8))
    nbytes = (bpp + 7) // 8
    if cmap_bits:
        body += bytes(rs.randint(0, 256, cmap_len * ((cmap_bits + 7) // 8)).astype(np.uint8))
        # a few indices beyond the map: they read entry 0
        idx = rs.randint(0, cmap_len + 2, w * h)
        px = bytearray(idx.astype(np.uint8).tobytes() if nbytes == 1 else idx.astype("<u2").tobytes())
    else:
        px = bytearray(rs.randint(0, 256, w * h * nbytes).astype(np.uint8).tobytes())
    return hdr + body + (_rle(px, nbytes, rs) if rle else bytes(px))
