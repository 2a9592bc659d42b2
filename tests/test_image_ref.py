"""Texture decoding pinned by the reference's own decoder.  No GPU.

The reference loads every texture through the stb_image.h it vendors (stbi_load(..., 4), texture_system.cpp:41-45).  That header
is the one part of the reference that compiles in this container (`make -C oracle ref` -> oracle/_ref/libstb_ref.so), so for this
boundary the product (capsaicin_amd/csrc/image_decode.cpp, jpeg_decode.cpp behind cap_image_decode) is compared with the
reference itself, tolerance 0:
  * test_golden_*: tests/golden/images/ (inputs) against expected.npz (stb's outputs, written by tools/make_image_fixtures.py) --
    runs wherever the product library loads;
  * test_live_*: only where oracle/_ref exists -- a few hundred more files made on the spot (PIL/libjpeg writes the common JPEG
    layouts, tests/*_craft.py the rest), each decoded by both, and the committed expectations re-derived.
"""
import io
import itertools
import os

import numpy as np
import pytest

import bmp_craft
import jpeg_craft
import png_craft
import tga_craft
from capsaicin_amd import capi
from oracle import stb_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images")
NAMES = sorted(n for n in os.listdir(GOLDEN) if n != "expected.npz")
needs_ref = pytest.mark.skipif(not stb_ref.available(), reason="oracle/_ref/libstb_ref.so not built (make -C oracle ref)")


def product(data, name):
    try:
        return capi.image_decode(data, name)
    except capi.CapError:
        return None


def same(got, ref):
    return got is not None and ref is not None and got.shape == ref.shape and np.array_equal(got, ref)


def test_golden_set_covers_every_container():
    ext = {os.path.splitext(n)[1] for n in NAMES}
    assert ext == {".jpg", ".png", ".tga", ".bmp", ".ppm", ".pgm"} and len(NAMES) >= 57


@pytest.mark.parametrize("name", NAMES)
def test_golden_image(native_lib, name):
    expected = np.load(os.path.join(GOLDEN, "expected.npz"))[name]
    got = product(open(os.path.join(GOLDEN, name), "rb").read(), name)
    assert got is not None, "refused"
    assert got.shape == expected.shape and got.dtype == np.uint8
    assert np.array_equal(got, expected), "%d pixels differ from the reference's decoder" % (got != expected).any(-1).sum()


@needs_ref
def test_live_expectations_are_the_reference_decoder_s():
    """expected.npz is not hand-made: the reference's decoder, run here, gives the same arrays."""
    expected = np.load(os.path.join(GOLDEN, "expected.npz"))
    assert sorted(expected.files) == NAMES
    for name in NAMES:
        ref = stb_ref.decode(open(os.path.join(GOLDEN, name), "rb").read())
        assert same(ref, expected[name]), name


REFERENCE_NOISE = "/root/reference/assets/textures/bluenoise256.png"


@pytest.mark.skipif(not os.path.exists(REFERENCE_NOISE), reason="the reference tree is only in the build container")
def test_reference_blue_noise_texture(native_lib, bluenoise):
    """The one texture the reference ships is its random-number source (raytracing_system.cpp:642-646): the product's decoder, the
    reference's decoder and the raw copy under assets/ (what cap_bluenoise_upload gets) agree on all 256 x 256 x 4 bytes."""
    data = open(REFERENCE_NOISE, "rb").read()
    got = product(data, "bluenoise256.png")
    assert same(got, bluenoise)
    if stb_ref.available():
        assert same(stb_ref.decode(data), bluenoise)


def _picture(mode, w, h, seed):
    from PIL import Image
    rs = np.random.RandomState(seed)
    ch = {"L": 1, "RGB": 3, "CMYK": 4}[mode]
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([np.sin(xx * 0.21 + c) * 60 + np.cos(yy * 0.13 * (c + 1)) * 50 + 128 + (xx * yy * (c + 1)) % 37 for c in range(ch)], -1)
    a = a + rs.randint(-20, 20, (h, w, ch))
    a[h // 3:h // 2, w // 4:w // 2] = 255 * (np.arange(ch) % 2)
    a[:h // 5, :w // 6] = 0
    a[h // 2:, w // 2:] += rs.randint(-120, 120, (h - h // 2, w - w // 2, ch))
    a = np.clip(a, 0, 255).astype(np.uint8)
    return Image.fromarray(a[..., 0] if ch == 1 else a, mode)


@needs_ref
def test_live_libjpeg_layouts(native_lib):
    """Baseline / progressive (spectral selection + successive approximation) x 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1 x optimised tables x
    restart intervals x grey / CMYK / RGB-id files, at sizes around the MCU edges."""
    pytest.importorskip("PIL.Image")
    sizes = [(64, 48), (37, 23), (1, 1), (8, 8), (17, 9), (15, 33), (7, 70), (100, 3), (129, 65), (2, 2)]
    files = []
    for (w, h), sub, prog, q in itertools.product(sizes, (0, 1, 2, "4:1:1"), (False, True), (30, 90, 100)):
        files.append(("RGB", w, h, dict(subsampling=sub, progressive=prog, quality=q, optimize=q == 90)))
    for (w, h), prog, mode in itertools.product(sizes, (False, True), ("L", "CMYK")):
        files.append((mode, w, h, dict(progressive=prog, quality=80)))
    for (w, h), prog, sub, blocks in itertools.product(sizes[:3], (False, True), (0, 2), (1, 3, 7)):
        files.append(("RGB", w, h, dict(progressive=prog, subsampling=sub, quality=80, restart_marker_blocks=blocks)))
    files.append(("RGB", 37, 23, dict(quality=90, keep_rgb=True)))
    for n, (mode, w, h, kw) in enumerate(files):
        b = io.BytesIO()
        _picture(mode, w, h, n).save(b, "JPEG", **kw)
        ref = stb_ref.decode(b.getvalue())
        assert ref is not None and same(product(b.getvalue(), "t.jpg"), ref), (mode, w, h, kw)
    assert len(files) > 300


@needs_ref
def test_live_crafted_layouts(native_lib):
    """Everything a sequential JPEG may legally be that libjpeg does not write: vertical / 4x / mixed sampling factors (each with
    its own upsampler in stb), four components under each Adobe transform, one scan per component, restart after every MCU,
    16-bit tables, tables replaced between scans, fill bytes, comments, DNL."""
    rs = np.random.RandomState(7)
    layouts = [[(1, 1)], [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)],
               [(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(4, 2), (1, 1), (1, 1)], [(2, 4), (1, 1), (1, 1)],
               [(4, 4), (1, 1), (1, 1)], [(2, 2), (2, 1), (1, 2)], [(4, 2), (2, 2), (1, 1)], [(2, 2), (1, 1), (2, 2)], [(1, 1)] * 4,
               [(2, 2), (1, 1), (1, 1), (2, 2)], [(1, 1), (2, 2), (2, 2)]]
    count = 0
    for s, (w, h), inter, restart in itertools.product(layouts, [(37, 23), (1, 1), (33, 70)], (True, False), (0, 1, 5)):
        data = jpeg_craft.random_file(rs, w, h, s, interleaved=inter, restart=restart)
        ref = stb_ref.decode(data)
        assert ref is not None and same(product(data, "t.jpg"), ref), (s, w, h, inter, restart)
        count += 1
    extras = [dict(wide_quant=True), dict(fill_bytes=True), dict(comment=True), dict(dnl=True), dict(interleaved=False, requant=True),
              dict(jfif=False, adobe_transform=0), dict(jfif=True, adobe_transform=0), dict(jfif=False, adobe_transform=1),
              dict(component_ids=[82, 71, 66]), dict(component_ids=[0, 1, 2])]
    for kw in extras:
        data = jpeg_craft.random_file(rs, 45, 31, [(2, 2), (1, 1), (1, 1)], **kw)
        assert same(product(data, "t.jpg"), stb_ref.decode(data)), kw
    for t, s in itertools.product((0, 1, 2), ([(1, 1)] * 4, [(2, 2), (1, 1), (1, 1), (2, 2)])):
        data = jpeg_craft.random_file(rs, 45, 31, s, adobe_transform=t, jfif=False)
        assert same(product(data, "t.jpg"), stb_ref.decode(data)), (t, s)
    assert count == 288


@needs_ref
def test_live_texture_sized_jpeg(native_lib):
    """One file of the size real material textures have (1024 x 1024, 4:2:0): every pixel."""
    pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    _picture("RGB", 1024, 1024, 99).save(b, "JPEG", quality=88, subsampling=2, progressive=True)
    assert same(product(b.getvalue(), "big.jpg"), stb_ref.decode(b.getvalue()))


@needs_ref
def test_live_png_layouts(native_lib):
    """Every colour type x bit depth x {plain, Adam7} at sizes that leave some interlace passes empty, a random filter type per
    scanline, IDAT in pieces, palette alpha and colour keys (half of them equal to a pixel of the image, so the key really fires --
    16-bit files compare whole samples, shallower ones a scaled key)."""
    rs = np.random.RandomState(5)
    sizes = [(1, 1), (2, 3), (3, 2), (5, 5), (8, 8), (9, 7), (13, 6), (37, 23), (4, 1), (1, 9)]
    count = keyed = 0
    for ctype, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth, inter, (w, h) in itertools.product(depths, (False, True), sizes):
            smp = rs.randint(0, 1 << depth, (h, w, png_craft.CHANNELS[ctype]))
            kw = {}
            if ctype == 3:
                kw["plte"] = rs.randint(0, 256, 3 * (1 << depth)).astype(np.uint8).tobytes()
                if rs.rand() < 0.5:
                    kw["trns"] = rs.randint(0, 256, min(1 << depth, 5)).astype(np.uint8).tobytes()
            elif ctype in (0, 2) and rs.rand() < 0.6:
                pick = smp[rs.randint(0, h), rs.randint(0, w)] if rs.rand() < 0.5 else rs.randint(0, 1 << depth, smp.shape[2])
                kw["trns"] = b"".join(int(v).to_bytes(2, "big") for v in pick)
            data = png_craft.write(smp, depth, ctype, inter, rs=rs, idat_pieces=int(rs.randint(1, 4)), **kw)
            ref = stb_ref.decode(data)
            assert ref is not None and same(product(data, "t.png"), ref), (ctype, depth, inter, w, h, list(kw))
            count += 1
            keyed += int("trns" in kw and ctype != 3 and (ref[..., 3] == 0).any())
    assert count == 300 and keyed > 20


@needs_ref
def test_live_png_refusals_match(native_lib):
    """Files stb refuses are refused: a colour key on a type with alpha, a key of the wrong length, palette alpha before the palette
    or longer than it, a key after the image data, an unknown critical chunk, a palette of broken length."""
    rs = np.random.RandomState(6)
    smp = lambda ctype: rs.randint(0, 256, (4, 5, png_craft.CHANNELS[ctype]))
    pal = rs.randint(0, 256, 3 * 7).astype(np.uint8).tobytes()
    cases = {
        "key_with_alpha": png_craft.write(smp(6), 8, 6, trns=b"\0\1\0\2\0\3"),
        "key_length": png_craft.write(smp(2), 8, 2, trns=b"\0\1"),
        "alpha_longer_than_palette": png_craft.write(smp(3) % 7, 8, 3, plte=pal, trns=bytes(8)),
        "palette_length": png_craft.write(smp(3) % 7, 8, 3, plte=pal[:-1]),
    }
    good = png_craft.write(smp(2), 8, 2)
    idat = good.index(b"IDAT") - 4
    iend = good.index(b"IEND") - 4
    cases["critical_chunk"] = good[:idat] + png_craft.chunk(b"ABCD", b"xyz") + good[idat:]
    cases["key_after_idat"] = good[:iend] + png_craft.chunk(b"tRNS", bytes(6)) + good[iend:]
    palimg = png_craft.write(smp(3) % 7, 8, 3, plte=pal)
    plte = palimg.index(b"PLTE") - 4
    cases["alpha_before_palette"] = palimg[:plte] + png_craft.chunk(b"tRNS", bytes(3)) + palimg[plte:]
    for name, data in cases.items():
        assert stb_ref.decode(data) is None, name + ": the reference decodes this"
        assert product(data, "t.png") is None, name
    # and an ancillary chunk nobody knows is skipped by both
    ok = good[:idat] + png_craft.chunk(b"abCd", b"xyz") + good[idat:]
    assert same(product(ok, "t.png"), stb_ref.decode(ok))


@needs_ref
def test_live_tga_layouts(native_lib):
    """TGA has no signature and stb reads it its own way (the layout follows the bit count, not the image type; "first colour-map
    entry" skips bytes; the right-to-left bit is ignored; indices beyond the map read entry 0): image types 1 / 2 / 3 x every depth
    x every descriptor byte x run-length packets crossing rows, with whatever name the file has."""
    rs = np.random.RandomState(9)
    count = 0
    for (w, h), rle, desc in itertools.product([(7, 5), (1, 1), (41, 19)], (False, True), (0, 0x20, 0x10, 0x30, 8, 0x28)):
        files = [dict(image_type=2, bpp=b) for b in (8, 15, 16, 24, 32)] + [dict(image_type=3, bpp=b) for b in (8, 15, 16, 24, 32)]
        files += [dict(image_type=1, bpp=8, cmap_bits=b, cmap_len=17) for b in (8, 15, 16, 24, 32)]
        files += [dict(image_type=1, bpp=16, cmap_bits=24, cmap_len=300), dict(image_type=1, bpp=8, cmap_bits=32, cmap_len=5, id_len=7)]
        if not rle:  # (with packets the skipped bytes shift the packet headers too: the stream then ends early, see below)
            files += [dict(image_type=1, bpp=8, cmap_bits=b, cmap_len=17, cmap_first=3, id_len=5) for b in (16, 24)]
        for kw in files:
            data = tga_craft.random_file(rs, w, h, descriptor=desc, rle=rle, **kw)
            ref = stb_ref.decode(data)
            assert ref is not None and same(product(data, "no_extension"), ref), (w, h, rle, hex(desc), kw)
            count += 1
    assert count > 450
    # what neither takes for a TGA
    for kw in (dict(image_type=2, bpp=12), dict(image_type=1, bpp=8, cmap_bits=12, cmap_len=4), dict(image_type=4, bpp=8),
               dict(image_type=1, bpp=24, cmap_bits=24, cmap_len=4)):
        data = tga_craft.random_file(rs, 4, 4, **kw)
        assert stb_ref.decode(data) is None and product(data, "t.tga") is None, kw
    zero = bytearray(tga_craft.random_file(rs, 4, 4, 2, 24))
    zero[12:14] = b"\0\0"
    assert stb_ref.decode(bytes(zero)) is None and product(bytes(zero), "t.tga") is None
    # the one deliberate difference: a file shorter than its header demands is an error here; stb pads it with zero bytes
    data = tga_craft.random_file(rs, 7, 5, 2, 24)[:-10]
    assert stb_ref.decode(data) is not None and product(data, "t.tga") is None


@needs_ref
def test_live_bmp_layouts(native_lib):
    """BMP as stb reads it: header sizes 12 / 40 / 56 / 108 / 124, palettes of 1 / 4 / 8 bits (any length), 24 bits, 16 / 32 bits with
    the default masks, BI_BITFIELDS behind a 40-byte header and the V4 / V5 headers' own masks (5-6-5, 4-4-4-4, 6-6-6, swapped
    8-8-8-8), bottom-up and top-down, the all-zero alpha channel that means "opaque".  Not compared, because stb's own output is
    undefined there (uninitialised palette entries / masks read from pixel data): OS/2 headers with a palette, BI_BITFIELDS behind a
    56-byte header."""
    pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    count = 0

    def both(data, what):
        nonlocal count
        ref = stb_ref.decode(data)
        assert ref is not None and same(product(data, "x.bmp"), ref), what
        count += 1
    for (w, h), hdr, td in itertools.product([(13, 7), (1, 1), (8, 3), (5, 2)], (12, 40, 56, 108, 124), (False, True)):
        if hdr == 12:
            if not td:
                both(bmp_craft.write(rs, w, h, 24, 12), (w, h, hdr))
            continue
        for bpp in (1, 4, 8, 24):
            both(bmp_craft.write(rs, w, h, bpp, hdr, top_down=td), (w, h, hdr, bpp, td))
        for bpp in (16, 32):
            default = None if hdr in (40, 56) else ((0x7c00, 0x3e0, 0x1f, 0) if bpp == 16 else (0xff0000, 0xff00, 0xff, 0xff000000))
            both(bmp_craft.write(rs, w, h, bpp, hdr, top_down=td, masks=default), (w, h, hdr, bpp, td))
            if hdr == 56:
                continue
            sets = ((0xf800, 0x7e0, 0x1f, 0), (0x0f00, 0x00f0, 0x000f, 0xf000)) if bpp == 16 else \
                ((0xff, 0xff00, 0xff0000, 0xff000000), (0x3f000000, 0x00fc0000, 0x0003f000, 0), (0xff0000, 0xff00, 0xff, 0))
            for masks in sets:
                both(bmp_craft.write(rs, w, h, bpp, hdr, masks=masks, top_down=td), (w, h, hdr, bpp, td, masks))
    both(bmp_craft.write(rs, 9, 4, 8, 40, palette_entries=17), "17 entries")
    both(bmp_craft.write(rs, 9, 4, 4, 40, palette_entries=5), "5 entries")
    zero_alpha = bytearray(bmp_craft.write(rs, 6, 3, 32, 40))
    for i in range(18):
        zero_alpha[54 + 4 * i + 3] = 0
    both(bytes(zero_alpha), "alpha all zero")
    assert (product(bytes(zero_alpha), "x.bmp")[..., 3] == 255).all()
    assert count > 250
    # refused by both: run-length coding, a header size nobody defines, two planes
    rle = bmp_craft.write(rs, 9, 4, 8, 40, compression=1)
    odd = bytearray(bmp_craft.write(rs, 9, 4, 24, 40))
    odd[14:18] = (52).to_bytes(4, "little")
    planes = bytearray(bmp_craft.write(rs, 9, 4, 24, 40))
    planes[26:28] = (2).to_bytes(2, "little")
    for data in (rle, bytes(odd), bytes(planes)):
        assert stb_ref.decode(data) is None and product(data, "x.bmp") is None
    # the deliberate difference again: a file that ends before its last row is an error here (stb pads with zero bytes)
    short = bmp_craft.write(rs, 9, 4, 24, 40)[:-5]
    assert stb_ref.decode(short) is not None and product(short, "x.bmp") is None


def test_corrupt_jpegs_end_as_errors(native_lib):
    """Truncated, bit-flipped and table-less streams: an error status (a missing texture), never a crash.  (stb hands back a partly
    decoded image for some of these; the product does not guess.)"""
    good = open(os.path.join(GOLDEN, "rgb420_progressive_restart.jpg"), "rb").read()
    base = open(os.path.join(GOLDEN, "rgb444_baseline.jpg"), "rb").read()
    for data in (good, base):
        for cut in (2, 3, 4, 20, 100, len(data) // 2, len(data) - 2):
            assert product(data[:cut], "t.jpg") is None
    rs = np.random.RandomState(3)
    for data in (good, base):
        for _ in range(200):
            d = bytearray(data)
            for _ in range(rs.randint(1, 4)):
                d[rs.randint(2, len(d))] = rs.randint(0, 256)
            product(bytes(d), "t.jpg")   # any outcome but a crash
    # a frame whose scan names a table that was never defined
    sos = base.index(b"\xff\xc4")
    end = base.index(b"\xff\xda")
    assert product(base[:sos] + base[end:], "t.jpg") is None
    # a header the file cannot back (8000 x 8000 in a 1.4 KB file: a block needs at least one coded bit) is refused before allocation
    big = bytearray(base)
    sof = base.index(b"\xff\xc0")
    big[sof + 5:sof + 9] = (8000).to_bytes(2, "big") * 2
    assert product(bytes(big), "t.jpg") is None
    # sampling factors that do not divide the largest one: no defined upsampling
    bad = bytearray(base)
    sof = base.index(b"\xff\xc0")
    bad[sof + 11] = 0x31   # first component 3 x 1
    bad[sof + 14] = 0x41   # second 4 x 1
    assert product(bytes(bad), "t.jpg") is None
