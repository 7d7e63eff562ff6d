"""The display / screenshot side of the reference's window without GL:

 * `present_rgba32f(ctx, fb, w, h)`  the float4 texels RenderWindow uploads to its RGBA32F texture
   (src/RenderWindow.cpp:172-176, 282-283); hand the array to glTexSubImage2D(..., GL_RGBA, GL_FLOAT, ...);
 * `screenshot(ctx, fb, w, h)`       what takeScreenshot (src/RenderWindow.cpp:389-421) reads back and flips: RGBA8 through
   the sRGB transfer function, row 0 = top, converted on the GPU by rt_present;
 * `save_png(path, rgba8)`           the lodepng::encode step, with a small zlib writer of our own;
 * `to_srgb8(img, thresholds)`       host restatement of the pinned 8-bit step, byte for byte what the device does.
"""
import ctypes as C
import struct
import zlib

import numpy as np

PRESENT_RGBA32F, PRESENT_SRGB8 = 0, 1


def _bind(L):
    if not getattr(L, "_present_bound", False):
        L.rt_present.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_void_p]
        L.rt_srgb8_thresholds.restype = C.POINTER(C.c_float)
        L._present_bound = True


def thresholds(L):
    """The library's table: T[k-1] = linear value whose sRGB encoding is (k - 0.5) / 255, k = 1..255."""
    _bind(L)
    return np.ctypeslib.as_array(L.rt_srgb8_thresholds(), shape=(255,)).copy()


def thresholds_python():
    """The same table computed here (IEC 61966-2-1 inverse transfer function in binary64, rounded to binary32)."""
    e = (np.arange(1, 256, dtype=np.float64) - 0.5) / 255.0
    return np.where(e <= 0.04045, e / 12.92, ((e + 0.055) / 1.055) ** 2.4).astype(np.float32)


def to_srgb8(img, table, flip=True):
    """Linear float RGBA (row 0 = bottom) -> RGBA8: byte = number of thresholds <= value (NaN -> 0); alpha linear."""
    img = np.asarray(img, np.float32)
    out = np.empty(img.shape, np.uint8)
    rgb = img[..., :3]
    out[..., :3] = np.where(np.isnan(rgb), 0, np.searchsorted(table, np.nan_to_num(rgb, nan=-1.0), side="right")).astype(np.uint8)
    a = np.nan_to_num(img[..., 3], nan=0.0).clip(0.0, 1.0)
    out[..., 3] = (a * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    return out[::-1].copy() if flip else out


def present_rgba32f(ctx, fb, w, h, flip=False):
    _bind(ctx.L)
    a = np.empty((h, w, 4), np.float32)
    ctx.check(ctx.L.rt_present(ctx.h, fb, PRESENT_RGBA32F, int(flip), a.ctypes.data))
    return a


def screenshot(ctx, fb, w, h, flip=True):
    _bind(ctx.L)
    a = np.empty((h, w, 4), np.uint8)
    ctx.check(ctx.L.rt_present(ctx.h, fb, PRESENT_SRGB8, int(flip), a.ctypes.data))
    return a


def save_png(path, rgba8):
    """RGBA8, row 0 = top."""
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    h, w = rgba8.shape[:2]
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(h))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
