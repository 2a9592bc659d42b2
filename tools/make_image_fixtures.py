#!/usr/bin/env python3
"""Writes tests/golden/images/: small texture files of every container and layout cap_image_decode accepts, and expected.npz,
what the REFERENCE's decoder makes of each (stb_image.h v2.25 as vendored under /root/reference, compiled by `make -C oracle ref`
and called through oracle/stb_ref.py: stbi_load_from_memory(..., 4), the call of texture_system.cpp:45).  The files are inputs, the
arrays are the reference's outputs; tests/test_image_ref.py holds the product decoder to them on any machine, and to a live stb
over a much larger corpus where oracle/_ref exists.

    python tools/make_image_fixtures.py          (needs PIL and oracle/_ref/libstb_ref.so)
"""
import io
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeg_craft  # noqa: E402
import bmp_craft  # noqa: E402
import png_craft  # noqa: E402
import tga_craft  # noqa: E402
from oracle import stb_ref  # noqa: E402


def picture(mode, w, h, seed):
    """Gradients, a saturated patch, a black patch and noise: smooth areas, hard edges and clamping in one small image."""
    rs = np.random.RandomState(seed)
    ch = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4, "CMYK": 4}[mode]
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([np.sin(xx * 0.21 + c) * 60 + np.cos(yy * 0.13 * (c + 1)) * 50 + 128 + (xx * yy * (c + 1)) % 37 for c in range(ch)], -1)
    a = a + rs.randint(-20, 20, (h, w, ch))
    a[h // 3:h // 2, w // 4:w // 2] = 255 * (np.arange(ch) % 2)
    a[:h // 5, :w // 6] = 0
    a[h // 2:, w // 2:] += rs.randint(-120, 120, (h - h // 2, w - w // 2, ch))
    a = np.clip(a, 0, 255).astype(np.uint8)
    from PIL import Image
    return Image.fromarray(a[..., 0] if ch == 1 else a, mode)


def saved(im, fmt, **kw):
    b = io.BytesIO()
    im.save(b, fmt, **kw)
    return b.getvalue()


def png_chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def raw_png(w, h, depth, ctype, rows, extra=b""):
    """rows: h byte strings of packed samples (filter type 0 is prepended)."""
    return (b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + extra +
            png_chunk(b"IDAT", zlib.compress(b"".join(b"\0" + r for r in rows))) + png_chunk(b"IEND", b""))


def corpus():
    from PIL import Image
    rs = np.random.RandomState(11)
    f = {}
    # JPEG as libjpeg writes it
    f["rgb444_baseline.jpg"] = saved(picture("RGB", 37, 23, 1), "JPEG", quality=75, subsampling=0)
    f["rgb422_progressive.jpg"] = saved(picture("RGB", 37, 23, 2), "JPEG", quality=85, subsampling=1, progressive=True)
    f["rgb420_optimised.jpg"] = saved(picture("RGB", 64, 48, 3), "JPEG", quality=90, subsampling=2, optimize=True)
    f["rgb420_progressive_restart.jpg"] = saved(picture("RGB", 45, 31, 4), "JPEG", quality=80, subsampling=2, progressive=True,
                                                restart_marker_blocks=3)
    f["rgb411.jpg"] = saved(picture("RGB", 53, 41, 5), "JPEG", quality=85, subsampling="4:1:1")
    f["grey_progressive.jpg"] = saved(picture("L", 17, 9, 6), "JPEG", quality=70, progressive=True)
    f["cmyk.jpg"] = saved(picture("CMYK", 33, 20, 7), "JPEG", quality=80)
    f["rgb_ids.jpg"] = saved(picture("RGB", 20, 20, 8), "JPEG", quality=90, keep_rgb=True)
    f["one_pixel.jpg"] = saved(picture("RGB", 1, 1, 9), "JPEG", quality=95)
    # JPEG layouts libjpeg does not write (tests/jpeg_craft.py)
    f["craft_v2.jpg"] = jpeg_craft.random_file(rs, 37, 23, [(1, 2), (1, 1), (1, 1)])
    f["craft_4x4.jpg"] = jpeg_craft.random_file(rs, 33, 70, [(4, 4), (1, 1), (1, 1)], restart=2)
    f["craft_mixed.jpg"] = jpeg_craft.random_file(rs, 45, 31, [(4, 2), (2, 2), (1, 1)], comment=True, fill_bytes=True)
    f["craft_scans_requant.jpg"] = jpeg_craft.random_file(rs, 45, 31, [(2, 2), (1, 1), (1, 1)], interleaved=False, requant=True, restart=5)
    f["craft_wide_quant.jpg"] = jpeg_craft.random_file(rs, 24, 24, [(2, 1), (1, 1), (1, 1)], wide_quant=True, dnl=True)
    f["craft_ycck.jpg"] = jpeg_craft.random_file(rs, 19, 12, [(1, 1)] * 4, adobe_transform=2, jfif=False)
    f["craft_adobe_rgb.jpg"] = jpeg_craft.random_file(rs, 19, 12, [(1, 1)] * 3, adobe_transform=0, jfif=False)
    # PNG
    f["rgb8.png"] = saved(picture("RGB", 37, 23, 20), "PNG")
    f["rgba8_stored.png"] = saved(picture("RGBA", 19, 11, 21), "PNG", compress_level=0)
    f["grey_alpha8.png"] = saved(picture("LA", 19, 11, 22), "PNG")
    pal = Image.fromarray(rs.randint(0, 17, (13, 21)).astype(np.uint8), "P")
    pal.putpalette(rs.randint(0, 256, 17 * 3).astype(np.uint8).tolist())
    f["palette_trns.png"] = saved(pal, "PNG", transparency=3)
    im = picture("RGB", 16, 9, 23)
    f["rgb8_colour_key.png"] = saved(im, "PNG", transparency=tuple(int(v) for v in np.asarray(im)[0, 0]))
    a = rs.randint(0, 65536, (5, 7, 3)).astype(">u2")
    f["rgb16.png"] = raw_png(7, 5, 16, 2, [a[y].tobytes() for y in range(5)])
    a = rs.randint(0, 65536, (5, 7, 2)).astype(">u2")
    f["grey_alpha16.png"] = raw_png(7, 5, 16, 4, [a[y].tobytes() for y in range(5)])
    for bits in (1, 2, 4):
        g = rs.randint(0, 1 << bits, (6, 13)).astype(np.uint8)
        rows = [np.packbits(np.unpackbits(g[y][:, None], axis=1)[:, 8 - bits:].reshape(-1)).tobytes() for y in range(6)]
        f["grey%d.png" % bits] = raw_png(13, 6, bits, 0, rows)
        f["palette%d.png" % bits] = raw_png(13, 6, bits, 3, rows, png_chunk(b"PLTE", rs.randint(0, 256, (1 << bits) * 3).astype(np.uint8).tobytes()))
    # PNG layouts PIL does not write (tests/png_craft.py): Adam7, a filter type drawn per scanline, colour keys at every depth
    f["adam7_rgb8.png"] = png_craft.random_file(rs, 37, 23, 8, 2, True, idat_pieces=3)
    f["adam7_rgba16.png"] = png_craft.random_file(rs, 9, 7, 16, 6, True)
    f["adam7_palette4_trns.png"] = png_craft.random_file(rs, 13, 6, 4, 3, True, trns=rs.randint(0, 256, 9).astype(np.uint8).tobytes())
    f["adam7_grey1.png"] = png_craft.random_file(rs, 5, 5, 1, 0, True)
    for depth, ctype, name in ((2, 0, "grey2_key"), (16, 0, "grey16_key"), (16, 2, "rgb16_key"), (8, 2, "adam7_rgb8_key")):
        smp = rs.randint(0, 1 << min(depth, 3), (6, 11, png_craft.CHANNELS[ctype])) * ((1 << depth) // 8 + 1) % (1 << depth)
        key = b"".join(int(v).to_bytes(2, "big") for v in smp[2, 3])
        f[name + ".png"] = png_craft.write(smp, depth, ctype, name.startswith("adam7"), trns=key, rs=rs)
    # TGA
    f["rgb.tga"] = saved(picture("RGB", 41, 19, 30), "TGA")
    run = np.asarray(picture("RGBA", 41, 19, 31)).copy()
    run[4:9, 5:30] = run[4, 5]
    f["rgba_rle.tga"] = saved(Image.fromarray(run, "RGBA"), "TGA", compression="tga_rle")
    f["grey_top_left.tga"] = saved(picture("L", 9, 7, 32), "TGA", orientation=1)
    f["palette.tga"] = saved(pal, "TGA")
    px = rs.randint(0, 65536, (5, 7)).astype("<u2")
    f["rgb555.tga"] = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 7, 5, 16, 0) + px.tobytes()
    f["grey_alpha.tga"] = struct.pack("<BBBHHBHHHHBB", 0, 0, 3, 0, 0, 0, 0, 0, 7, 5, 16, 8) + px.tobytes()
    # TGA corners where stb's reading is the contract (tests/tga_craft.py): an 8-bit colour map (grey entries), 16-bit indices, bytes
    # skipped in front of the map, 8 bits in a "true colour" file, 24 bits in a "grey" one, the right-to-left bit (ignored)
    f["craft_cmap8.tga"] = tga_craft.random_file(rs, 9, 5, 1, 8, cmap_bits=8, cmap_len=11)
    f["craft_index16_rle.tga"] = tga_craft.random_file(rs, 9, 5, 1, 16, cmap_bits=24, cmap_len=300, rle=True, descriptor=0x20)
    f["craft_cmap_skip.tga"] = tga_craft.random_file(rs, 9, 5, 1, 8, cmap_bits=32, cmap_len=17, cmap_first=3, id_len=5)
    f["craft_type2_bpp8.tga"] = tga_craft.random_file(rs, 9, 5, 2, 8)
    f["craft_type3_bpp24_rle.tga"] = tga_craft.random_file(rs, 9, 5, 3, 24, rle=True)
    f["craft_right_to_left.tga"] = tga_craft.random_file(rs, 9, 5, 2, 15, descriptor=0x10)
    # BMP: what PIL writes, and what it does not (tests/bmp_craft.py)
    f["rgb24.bmp"] = saved(picture("RGB", 13, 7, 50), "BMP")
    f["palette8.bmp"] = saved(pal, "BMP")
    f["craft_rgb565_top_down.bmp"] = bmp_craft.write(rs, 9, 5, 16, 40, masks=(0xf800, 0x7e0, 0x1f), top_down=True)
    f["craft_rgb555.bmp"] = bmp_craft.write(rs, 9, 5, 16, 40)
    f["craft_bgra32_v5.bmp"] = bmp_craft.write(rs, 9, 5, 32, 124, masks=(0xff0000, 0xff00, 0xff, 0xff000000))
    f["craft_palette4.bmp"] = bmp_craft.write(rs, 9, 5, 4, 40, palette_entries=11)
    f["craft_palette1_v4.bmp"] = bmp_craft.write(rs, 9, 5, 1, 108)
    f["craft_os2_rgb24.bmp"] = bmp_craft.write(rs, 9, 5, 24, 12)
    # PNM
    f["rgb.ppm"] = saved(picture("RGB", 13, 5, 40), "PPM")
    f["grey.pgm"] = saved(picture("L", 13, 5, 41), "PPM")
    return f


def main():
    out = os.path.join(ROOT, "tests", "golden", "images")
    os.makedirs(out, exist_ok=True)
    expected = {}
    for name, data in sorted(corpus().items()):
        ref = stb_ref.decode(data)
        assert ref is not None, (name, stb_ref.failure_reason())
        with open(os.path.join(out, name), "wb") as fh:
            fh.write(data)
        expected[name] = ref
        print("%-34s %6d bytes -> %s" % (name, len(data), ref.shape))
    np.savez_compressed(os.path.join(out, "expected.npz"), **expected)


if __name__ == "__main__":
    main()
