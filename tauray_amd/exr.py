"""OpenEXR files for the Python mirror: `.exr` environment maps / textures in (texture::load_from_file -> read_exr,
src/texture.cc:70-163, 405-429) and the frames headless writes out (src/headless.cc:355-412).  Both directions go through the C ABI
(trhip_exr_decode / trhip_exr_encode -> include/tauray_exr.hh), the same code the C++ host includes, so the two hosts read a file
into the same floats."""
from __future__ import annotations

import ctypes as C

import numpy as np

NONE, RLE, ZIPS, ZIP, PIZ = 0, 1, 2, 3, 4      # OpenEXR compression codes = headless::compression_type (src/headless.hh:25-32)


def decode_exr(data: bytes) -> np.ndarray:
    """-> (H, W, channels) float32, channels <= 4 ordered R, G, B, A (or file order for other channel names), row 0 = top."""
    from . import _lib
    L = _lib.lib()
    w, h, n, p = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    _lib.check(L.trhip_exr_decode(bytes(data), len(data), C.byref(w), C.byref(h), C.byref(n), C.byref(p)))
    out = np.ctypeslib.as_array(p, (h.value, w.value, n.value)).copy()
    L.trhip_exr_free(p)
    return out


def load_exr(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_exr(f.read())


def load_exr_rgba(path: str) -> np.ndarray:
    """The texture the reference makes of the file: three channels get alpha 1 (src/texture.cc:423-428); floats stay floats."""
    img = load_exr(path)
    if img.shape[2] == 3:
        img = np.concatenate([img, np.ones(img.shape[:2] + (1,), np.float32)], axis=2)
    if img.shape[2] != 4:
        raise ValueError(f"{path}: {img.shape[2]} channels; an environment map needs three or four")
    return np.ascontiguousarray(img)


def encode_exr(rgba: np.ndarray, alpha: bool = False, half: bool = True, compression: int = PIZ) -> bytes:
    """(H, W, 4) float32 -> the bytes of the scanline file headless::save_image writes: channels [A,] B, G, R."""
    from . import _lib
    L = _lib.lib()
    img = np.ascontiguousarray(rgba, dtype=np.float32)
    assert img.ndim == 3 and img.shape[2] == 4
    p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _lib.check(L.trhip_exr_encode(img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0], int(alpha), int(half), int(compression), C.byref(p), C.byref(n)))
    out = bytes(np.ctypeslib.as_array(p, (n.value,)))
    L.trhip_exr_free(p)
    return out
