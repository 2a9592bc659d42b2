"""Texture decoding of the host layer (capsaicin_amd/csrc/image_decode.cpp, C ABI cap_image_decode) against PIL on files PIL
writes: what TextureSystem's stbi_load(file, &w, &h, &n, 4) hands over in the reference (texture_system.cpp:41-45) -- 8-bit RGBA,
rows top to bottom, grey replicated, alpha 255 when the file has none.  No GPU."""
import io

import numpy as np
import pytest

from capsaicin_amd import capi

PIL = pytest.importorskip("PIL.Image")


def image(mode, w=37, h=23, seed=0):
    rs = np.random.RandomState(seed)
    if mode == "P":
        im = PIL.fromarray(rs.randint(0, 17, (h, w)).astype(np.uint8), "P")
        im.putpalette(rs.randint(0, 256, 17 * 3).astype(np.uint8).tolist())
        return im
    if mode == "1":
        return PIL.fromarray((rs.randint(0, 2, (h, w)) * 255).astype(np.uint8), "L").convert("1")
    if mode == "I;16":
        return PIL.fromarray(rs.randint(0, 65536, (h, w)).astype(np.uint16), "I;16")
    ch = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
    # smooth gradients + noise: exercises every PNG filter heuristics pick
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([(xx * 5 + yy * 3 + 40 * c) % 256 for c in range(ch)], -1) + rs.randint(0, 9, (h, w, ch))
    return PIL.fromarray(np.squeeze(a % 256).astype(np.uint8), mode)


def expected(im):
    if im.mode == "I;16":  # stb keeps the high byte of 16-bit samples
        g = (np.asarray(im).astype(np.uint16) >> 8).astype(np.uint8)
        return np.stack([g, g, g, np.full_like(g, 255)], -1)
    return np.asarray(im.convert("RGBA"))


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1", "I;16"])
@pytest.mark.parametrize("level", [0, 1, 9])
def test_png(native_lib, mode, level):
    im = image(mode, seed=level)
    buf = io.BytesIO()
    im.save(buf, "PNG", compress_level=level)  # level 0: stored blocks; others: dynamic / fixed Huffman
    got = capi.image_decode(buf.getvalue(), "t.png")
    assert got.shape == (im.height, im.width, 4) and np.array_equal(got, expected(im))


def test_png_large_and_optimised(native_lib):
    im = image("RGB", 300, 211, seed=5)
    buf = io.BytesIO()
    im.save(buf, "PNG", optimize=True)
    assert np.array_equal(capi.image_decode(buf.getvalue()), expected(im))


def test_png_short_interlaced_and_garbage_are_refused(native_lib):
    # a hand-made interlaced file whose data ends before the seventh pass (55 bytes are due)
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\0" * 52)) + chunk(b"IEND", b"")
    with pytest.raises(capi.CapError):
        capi.image_decode(png, "i.png")
    with pytest.raises(capi.CapError):
        capi.image_decode(b"\xff\xd8\xff\xe0 not a format this build decodes", "t.jpg")
    with pytest.raises(capi.CapError):
        capi.image_decode(b"\x89PNG\r\n\x1a\n" + b"\0" * 40, "t.png")


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L"])
@pytest.mark.parametrize("rle", [False, True])
def test_tga(native_lib, mode, rle):
    im = image(mode, 41, 19, seed=3)
    if rle:  # runs for the run-length packets
        a = np.asarray(im).copy()
        a[4:9, 5:30] = a[4, 5]
        im = PIL.fromarray(a, mode)
    buf = io.BytesIO()
    im.save(buf, "TGA", compression="tga_rle" if rle else None)
    got = capi.image_decode(buf.getvalue(), "t.tga")
    assert np.array_equal(got, expected(im))


def test_tga_top_left_origin(native_lib):
    im = image("RGB", 9, 7, seed=8)
    buf = io.BytesIO()
    im.save(buf, "TGA", orientation=1)  # top-left origin
    assert np.array_equal(capi.image_decode(buf.getvalue(), "t.TGA"), expected(im))


def test_ppm(native_lib):
    im = image("RGB", 13, 5, seed=2)
    buf = io.BytesIO()
    im.save(buf, "PPM")
    assert np.array_equal(capi.image_decode(buf.getvalue(), "t.ppm"), expected(im))


def test_oversized_truncated_and_bomb_inputs_end_as_errors(native_lib):
    """Sizes come out of a file's header before any data is seen: a 100-byte file must not be able to ask for
    gigabytes, a deflate stream must stop at the size its header promised, a colour map must have a depth the expander knows, and
    nothing may leave cap_image_decode as a C++ exception."""
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    def png(w, h, depth, ctype, payload):
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", payload) + chunk(b"IEND", b"")
    # 32768 x 32768 16-bit RGBA: 8 GiB by its header
    with pytest.raises(capi.CapError):
        capi.image_decode(png(32768, 32768, 16, 6, zlib.compress(b"\0" * 64)), "big.png")
    # a zlib bomb behind a 4 x 4 header: 64 MiB of zeros where 52 bytes are due
    with pytest.raises(capi.CapError):
        capi.image_decode(png(4, 4, 8, 2, zlib.compress(b"\0" * (64 << 20), 9)), "bomb.png")
    # truncated streams of a good file, every few bytes
    im = image("RGBA", 21, 17, seed=4)
    buf = io.BytesIO()
    im.save(buf, "PNG")
    good = buf.getvalue()
    assert np.array_equal(capi.image_decode(good, "g.png"), expected(im))
    for cut in range(8, len(good) - 12, 7):
        try:
            capi.image_decode(good[:cut], "cut.png")
        except capi.CapError:
            pass
    # TGA: 65535 x 65535 x 32 bits uncompressed by its header; a colour map of a depth nobody defines; a truncated RLE body
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 65535, 65535, 32, 0)
    with pytest.raises(capi.CapError):
        capi.image_decode(hdr + b"\0" * 64, "big.tga")
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 1, 1, 0, 4, 12, 0, 0, 2, 2, 8, 0)
    with pytest.raises(capi.CapError):
        capi.image_decode(hdr + bytes(8) + bytes(4), "cmap12.tga")
    # (an 8-bit map is a map of grey values in the reference's decoder: tests/test_image_ref.py)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 1, 1, 0, 4, 8, 0, 0, 2, 2, 8, 0)
    assert np.array_equal(capi.image_decode(hdr + bytes([9, 8, 7, 6]) + bytes([0, 1, 2, 3]), "cmap8.tga")[..., 0], [[7, 6], [9, 8]])
    buf = io.BytesIO()
    image("RGB", 41, 19, seed=3).save(buf, "TGA", compression="tga_rle")
    for cut in range(18, len(buf.getvalue()), 11):
        try:
            capi.image_decode(buf.getvalue()[:cut], "cut.tga")
        except capi.CapError:
            pass
    with pytest.raises(capi.CapError):
        capi.image_decode(b"P6\n40000 40000\n255\n" + b"\0" * 100, "big.ppm")


def test_bmp_header_cannot_buy_memory(native_lib):
    """a 26-byte BMP that declares 8192 x 8192 pixels used to make the decoder allocate and touch 256 MB before it noticed
    that the file cannot back the header (16 decode threads of the host layer: 4 GB).  The refusal now comes before the allocation --
    seen here as peak resident memory of the process -- and a height of INT32_MIN (whose negation overflows) is refused as well."""
    import resource
    import struct

    def bmp(w, h, bpp=24, body=b""):
        return b"BM" + struct.pack("<IHHI", 54 + len(body), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, bpp, 0, 0, 0, 0, 0, 0) + body

    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for k in range(40):
        with pytest.raises(capi.CapError):
            capi.image_decode(bmp(8192, 8192), "huge.bmp")
    with pytest.raises(capi.CapError):
        capi.image_decode(bmp(4, -2147483648), "minint.bmp")
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert after - before < 64 * 1024, "the refused files grew the process by %d KiB" % (after - before)
    # a file that does back its header still decodes
    body = bytes(range(3 * 4 * 2)) + b""
    assert capi.image_decode(bmp(4, 2, 24, body), "ok.bmp").shape == (2, 4, 4)
