"""include/tauray_image.hh (texture files -> RGBA8, what stb_image does for the reference's glTF loader, src/gltf.cc:520-576) through
its C entry point trhip_image_decode - the decoder both hosts use.  Checked against an independent decoder (Pillow: libpng /
libjpeg-turbo) on files made here: PNG must agree exactly (16-bit samples: the rounded 8-bit value), JPEG - whose inverse DCT and
chroma upsampling the standard leaves to the decoder - to a few levels.  No GPU involved."""
import ctypes as C
import io
import struct
import zlib

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def decode(data: bytes):
    from tauray_amd import _lib
    L = _lib.lib()
    w, h, ch, p = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    if L.trhip_image_decode(data, len(data), C.byref(w), C.byref(h), C.byref(ch), C.byref(p)) != 0:
        raise _lib.TrhipError(L.trhip_last_error().decode())
    a = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
    L.trhip_image_free(p)
    return a, ch.value


def _picture(w, h, seed=1):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 5.0 + y / 9.0), (x * 3 + y * 2) % 256,
                  255 * ((x // 3 + y // 2) % 2)], -1) + rng.normal(0, 6, (h, w, 4))
    return np.clip(a, 0, 255).astype(np.uint8)


def _save(arr, mode, fmt, **kw):
    b = io.BytesIO()
    PIL.fromarray(arr, mode).save(b, fmt, **kw)
    return b.getvalue()


def _png(w, h, depth, ctype, rows, interlace=0, extra=b""):
    """A PNG file from raw scanlines (filter type 0 on every row; `rows` = list of passes, each a list of packed row bytes)."""
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    raw = b"".join(b"\x00" + r for p in rows for r in p)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("size", [(64, 48), (33, 17), (1, 1), (5, 3), (200, 131)])
def test_png_colour_types_match_pillow(size):
    w, h = size
    a = _picture(w, h)
    cases = {"rgba": (_save(a, "RGBA", "PNG"), 4), "rgb": (_save(a[..., :3].copy(), "RGB", "PNG"), 3), "grey": (_save(a[..., 0].copy(), "L", "PNG"), 1),
             "grey_alpha": (_save(a[..., [0, 3]].copy(), "LA", "PNG"), 2),
             "palette": (_save(np.asarray(PIL.fromarray(a[..., :3].copy(), "RGB").quantize(37)), "P", "PNG"), None)}
    pal = PIL.fromarray(a[..., :3].copy(), "RGB").quantize(37)
    b = io.BytesIO(); pal.save(b, "PNG"); cases["palette"] = (b.getvalue(), 3)
    b = io.BytesIO(); pal.save(b, "PNG", transparency=5); cases["palette_trns"] = (b.getvalue(), 4)
    for name, (data, ch) in cases.items():
        got, n = decode(data)
        ref = np.array(PIL.open(io.BytesIO(data)).convert("RGBA"))
        assert got.shape == (h, w, 4) and np.array_equal(got, ref), name
        assert n == ch, name


def test_png_sixteen_bit_sub_byte_and_interlaced():
    w, h = 37, 23
    rng = np.random.default_rng(3)
    # 16 bits per sample, RGBA and grey: the 8-bit value is the rounded one
    v = rng.integers(0, 65536, size=(h, w, 4), dtype=np.uint16)
    rows = [[v[y].astype(">u2").tobytes() for y in range(h)]]
    got, ch = decode(_png(w, h, 16, 6, rows))
    assert ch == 4 and np.array_equal(got, ((v.astype(np.uint32) * 255 + 32767) // 65535).astype(np.uint8))
    g16 = v[..., 0]
    got, ch = decode(_png(w, h, 16, 0, [[g16[y].astype(">u2").tobytes() for y in range(h)]]))
    want = ((g16.astype(np.uint32) * 255 + 32767) // 65535).astype(np.uint8)
    assert ch == 1 and np.array_equal(got[..., 0], want) and np.array_equal(got[..., 2], want) and (got[..., 3] == 255).all()
    # ... and the texels a scene stores keep all sixteen bits (trhip_image_decode_texels; the reference's R16G16B16A16Unorm, src/gltf.cc:548-556):
    # RGBA as it is, grey expanded to (g, g, g, 65535) as stb_image does for tinygltf; an 8-bit file stays 8 bits
    from tauray_amd.gltf import decode_image
    t = decode_image(_png(w, h, 16, 6, rows))
    assert t.dtype == np.uint16 and np.array_equal(t, v)
    t = decode_image(_png(w, h, 16, 0, [[g16[y].astype(">u2").tobytes() for y in range(h)]]))
    assert t.dtype == np.uint16 and np.array_equal(t[..., 0], g16) and np.array_equal(t[..., 1], g16) and np.array_equal(t[..., 2], g16) and (t[..., 3] == 65535).all()
    t = decode_image(_png(w, h, 16, 2, [[v[y, :, :3].astype(">u2").tobytes() for y in range(h)]]))
    assert t.dtype == np.uint16 and np.array_equal(t[..., :3], v[..., :3]) and (t[..., 3] == 65535).all()
    assert decode_image(_png(w, h, 8, 6, [[(v[y] >> 8).astype(np.uint8).tobytes() for y in range(h)]])).dtype == np.uint8
    # 1, 2 and 4 bits per sample (grey): Pillow reads those too
    for depth in (1, 2, 4):
        g = rng.integers(0, 1 << depth, size=(h, w), dtype=np.uint8)
        packed = []
        for y in range(h):
            bits = np.zeros(((w * depth + 7) // 8) * 8, dtype=np.uint8)
            bits[:w * depth] = np.unpackbits(g[y][:, None], axis=1)[:, 8 - depth:].reshape(-1)
            packed.append(np.packbits(bits).tobytes())
        data = _png(w, h, depth, 0, [packed])
        got, _ = decode(data)
        assert np.array_equal(got[..., 0], (g.astype(np.uint32) * 255 // ((1 << depth) - 1)).astype(np.uint8)), depth
        assert np.array_equal(got, np.array(PIL.open(io.BytesIO(data)).convert("RGBA"))) or depth != 1      # Pillow maps 1-bit grey the same way
    # Adam7: the seven passes of an RGB image, by hand
    a = _picture(w, h, 9)[..., :3]
    passes = []
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[y0::dy, x0::dx]
        passes.append([sub[y].tobytes() for y in range(sub.shape[0])] if sub.size else [])
    data = _png(w, h, 8, 2, passes, interlace=1)
    got, ch = decode(data)
    assert ch == 3 and np.array_equal(got[..., :3], a) and (got[..., 3] == 255).all()
    assert np.array_equal(got, np.array(PIL.open(io.BytesIO(data)).convert("RGBA")))
    # tRNS on an RGB image: one colour becomes transparent
    key = a[3, 4]
    trns = struct.pack(">HHH", *[int(c) for c in key])
    body = b"tRNS" + trns
    data = _png(w, h, 8, 2, [[a[y].tobytes() for y in range(h)]], extra=struct.pack(">I", 6) + body + struct.pack(">I", zlib.crc32(body)))
    got, ch = decode(data)
    assert ch == 4 and got[3, 4, 3] == 0 and np.array_equal(got[..., 3] == 0, (a == key).all(-1))


@pytest.mark.parametrize("size", [(64, 48), (33, 17), (200, 131), (8, 8), (1, 1), (17, 40)])
def test_jpeg_matches_an_independent_decoder_to_a_few_levels(size):
    w, h = size
    a = _picture(w, h)[..., :3].copy()
    for mode, kw in (("RGB", dict(subsampling=0)), ("RGB", dict(subsampling=1)), ("RGB", dict(subsampling=2)), ("L", {}),
                     ("RGB", dict(subsampling=2, restart_marker_blocks=3)), ("RGB", dict(subsampling=0, optimize=True)), ("RGB", dict(subsampling=2, quality=35))):
        data = _save(a if mode == "RGB" else a[..., 0].copy(), mode, "JPEG", **{"quality": 90, **kw})
        got, ch = decode(data)
        ref = np.array(PIL.open(io.BytesIO(data)).convert("RGBA")).astype(int)
        d = np.abs(got.astype(int) - ref)
        assert got.shape == (h, w, 4) and ch == (3 if mode == "RGB" else 1) and (got[..., 3] == 255).all()
        assert d.max() <= 6 and d[..., :3].mean() <= 0.4, (mode, kw, int(d.max()), float(d.mean()))


@pytest.mark.parametrize("size", [(64, 48), (33, 17), (200, 131), (8, 8), (1, 1), (17, 40), (131, 77)])
def test_progressive_jpeg_gives_the_pixels_of_the_sequential_file(size):
    """Progressive files (SOF2: spectral selection + successive approximation, DC scans interleaved, AC scans per component with
    end-of-band runs and refinement passes) carry the same quantised coefficients as the sequential file Pillow writes of the same
    picture at the same quality, so this decoder must return the same pixels for both, bit for bit; and both stay within a few
    levels of libjpeg's own decoding."""
    w, h = size
    a = _picture(w, h, seed=w + h)[..., :3].copy()
    for mode, kw in (("RGB", dict(subsampling=0)), ("RGB", dict(subsampling=1)), ("RGB", dict(subsampling=2)), ("L", {}),
                     ("RGB", dict(subsampling=2, quality=35)), ("RGB", dict(subsampling=0, quality=98)), ("RGB", dict(subsampling=2, restart_marker_blocks=2))):
        src = a if mode == "RGB" else a[..., 0].copy()
        kw = {"quality": 90, **kw}
        seq = _save(src, mode, "JPEG", **kw)
        prog = _save(src, mode, "JPEG", progressive=True, **kw)
        assert b"\xff\xc2" in prog and b"\xff\xc2" not in seq
        got_seq, _ = decode(seq)
        got, ch = decode(prog)
        assert ch == (3 if mode == "RGB" else 1)
        assert np.array_equal(got, got_seq), (mode, kw)
        ref = np.array(PIL.open(io.BytesIO(prog)).convert("RGBA")).astype(int)
        d = np.abs(got.astype(int) - ref)
        assert d.max() <= 6 and d[..., :3].mean() <= 0.4, (mode, kw, int(d.max()), float(d.mean()))


def test_a_progressive_file_cut_after_its_first_scan_is_a_picture_of_block_means():
    """Scans are applied as they come: a progressive file that ends after its first scan (the DC terms, here with a point transform
    of one bit) decodes to 8 x 8 blocks that are flat at the block means of the picture.  Grey, so that no chroma resampling is involved."""
    a = _picture(70, 50)[..., 1].copy()
    prog = _save(a, "L", "JPEG", progressive=True, quality=95)
    first = prog.index(b"\xff\xda")
    second = prog.index(b"\xff\xda", first + 2)
    cut = prog.rfind(b"\xff\xc4", first + 12, second)      # the Huffman table in front of the second scan, if there is one
    got, ch = decode(prog[:cut if cut > 0 else second] + b"\xff\xd9")
    assert ch == 1
    blocks = got[:48, :64, 0].astype(int).reshape(6, 8, 8, 8)
    assert (blocks.max((1, 3)) - blocks.min((1, 3))).max() == 0
    means = a[:48, :64].astype(float).reshape(6, 8, 8, 8).mean((1, 3))
    assert np.abs(blocks[:, 0, :, 0] - means).max() <= 2.0


def test_random_images_match_pillow():
    """Seeded draws: sizes from 1 x 1 to a few hundred pixels that are multiples of nothing, PNG in every mode Pillow writes (1-bit, grey,
    grey + alpha, palette with and without transparency, RGB, RGBA, 16-bit grey) at several compression levels, JPEG at qualities 1-100
    with 4:4:4 / 4:2:2 / 4:2:0 chroma, grey, progressive or sequential, optimised tables, restart intervals in blocks or rows.  PNG must
    agree exactly, JPEG to a few levels.  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL run longer campaigns."""
    import os
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "7")))
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "40"))):
        w, h = int(rng.integers(1, 260)), int(rng.integers(1, 200))
        a = _picture(w, h, seed=int(rng.integers(0, 1 << 30)))
        if rng.uniform() < 0.5:
            mode = str(rng.choice(["1", "L", "LA", "P", "PA", "RGB", "RGBA", "I;16"]))
            what = f"draw {k}: PNG {mode} {w}x{h}"
            if mode == "1":
                img = PIL.fromarray(a[..., 0] > 128)
            elif mode == "L":
                img = PIL.fromarray(a[..., 0], "L")
            elif mode == "LA":
                img = PIL.fromarray(a[..., [0, 3]].copy(), "LA")
            elif mode in ("P", "PA"):
                img = PIL.fromarray(a[..., :3].copy(), "RGB").quantize(int(rng.choice([2, 7, 16, 200])))
                if mode == "PA":
                    img.info["transparency"] = int(rng.integers(0, 2))
            elif mode == "RGB":
                img = PIL.fromarray(a[..., :3].copy(), "RGB")
            elif mode == "RGBA":
                img = PIL.fromarray(a, "RGBA")
            else:
                img = PIL.fromarray((a[..., 0].astype(np.uint16) * 257 + rng.integers(0, 200, (h, w)).astype(np.uint16)), "I;16")
            b = io.BytesIO()
            kw = {"compress_level": int(rng.integers(0, 10))}
            if mode == "PA":
                kw["transparency"] = img.info["transparency"]
            img.save(b, "PNG", **kw)
            data = b.getvalue()
            got, ch = decode(data)
            if mode == "I;16":
                v = np.array(PIL.open(io.BytesIO(data))).astype(np.uint32)
                want = ((v * 255 + 32767) // 65535).astype(np.uint8)      # rounded v * 255 / 65535
                assert np.array_equal(got[..., 0], want) and np.array_equal(got[..., 1], want) and (got[..., 3] == 255).all(), what
            else:
                ref = np.array(PIL.open(io.BytesIO(data)).convert("RGBA"))
                assert got.shape == ref.shape and np.array_equal(got, ref), f"{what}: {int((got != ref).any(-1).sum())} pixels differ"
        else:
            grey = rng.uniform() < 0.25
            kw = {"quality": int(rng.integers(1, 101))}
            if not grey:
                kw["subsampling"] = int(rng.integers(0, 3))
            if rng.uniform() < 0.5:
                kw["progressive"] = True
            if rng.uniform() < 0.3:
                kw["optimize"] = True
            r = rng.uniform()
            if r < 0.25:
                kw["restart_marker_blocks"] = int(rng.integers(1, 9))
            elif r < 0.4:
                kw["restart_marker_rows"] = int(rng.integers(1, 4))
            what = f"draw {k}: JPEG {'L' if grey else 'RGB'} {w}x{h} {kw}"
            data = _save(a[..., 0].copy() if grey else a[..., :3].copy(), "L" if grey else "RGB", "JPEG", **kw)
            got, ch = decode(data)
            ref = np.array(PIL.open(io.BytesIO(data)).convert("RGBA")).astype(int)
            d = np.abs(got.astype(int) - ref)
            assert got.shape == (h, w, 4) and ch == (1 if grey else 3) and (got[..., 3] == 255).all(), what
            if not grey and kw["subsampling"] and w <= 4:
                continue      # libjpeg replicates chroma instead of interpolating it when a chroma row has at most two samples (jdsample.c)
            # coarse quantisation (quality below ~20) leaves the two inverse DCTs and chroma filters further apart on single pixels
            lim = 6 if kw["quality"] >= 25 else 14
            assert d.max() <= lim and d[..., :3].mean() <= 0.6, f"{what}: max {int(d.max())}, mean {float(d.mean()):.3f}"


def test_damaged_images_end_in_an_error_or_an_image():
    """Bit flips, truncation and overwritten stretches in PNG and JPEG files (sequential and progressive): the decoder answers with an
    error or with an image of the announced size - it is fed files from the outside world (a campaign of 4 500 such files by hand:
    profiles/r3/fuzz_campaign.txt).  TRHIP_FUZZ_SEED / TRHIP_FUZZ_DRAWS_SMALL for more."""
    import os
    from tauray_amd import _lib
    rng = np.random.default_rng(int(os.environ.get("TRHIP_FUZZ_SEED", "2")))
    decoded = refused = 0
    for k in range(int(os.environ.get("TRHIP_FUZZ_DRAWS_SMALL", "200"))):
        w, h = int(rng.integers(1, 80)), int(rng.integers(1, 60))
        a = _picture(w, h, seed=k)
        r = rng.uniform()
        if r < 0.35:
            data = _save(a[..., :3].copy(), "RGB", "JPEG", quality=int(rng.integers(5, 100)), subsampling=int(rng.integers(0, 3)), progressive=bool(rng.integers(0, 2)))
        elif r < 0.5:
            data = _save(a[..., 0].copy(), "L", "JPEG", quality=50, progressive=bool(rng.integers(0, 2)))
        else:
            data = _save(a, "RGBA", "PNG") if r < 0.75 else _save(a[..., :3].copy(), "RGB", "PNG")
        b = bytearray(data)
        m = rng.uniform()
        if m < 0.6:
            for q in rng.integers(2, len(b), int(rng.integers(1, 8))):
                b[q] ^= 1 << int(rng.integers(0, 8))
        elif m < 0.8:
            b = b[:int(rng.integers(2, len(b)))]
        else:
            q = int(rng.integers(2, len(b)))
            b[q:q + int(rng.integers(1, 20))] = bytes(rng.integers(0, 256, int(rng.integers(1, 20)), dtype=np.uint8))
        try:
            img, _ = decode(bytes(b))
            assert img.ndim == 3 and img.shape[2] == 4 and img.shape[0] > 0 and img.shape[1] > 0
            decoded += 1
        except _lib.TrhipError:
            refused += 1
    assert decoded > 0 and refused > 0


def test_unreadable_files_fail_loudly():
    from tauray_amd import _lib
    a = _picture(16, 16)[..., :3].copy()
    with pytest.raises(_lib.TrhipError):
        decode(b"GIF89a" + b"\0" * 64)
    good = _save(a, "RGB", "PNG")
    with pytest.raises(_lib.TrhipError):
        decode(good[:len(good) // 2])
    with pytest.raises(_lib.TrhipError):
        decode(_save(a, "RGB", "JPEG")[:200])
