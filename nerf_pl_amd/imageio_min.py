"""Image / depth writers of the reference's eval loop (eval.py:119-149) without its imageio / cv2 dependencies:
PNG (8-bit RGB or grey, zlib-deflated, filter 0), PFM (datasets/depth_utils.py:43-69 `save_pfm` / `read_pfm`, same bytes)
and the raw little-endian float32 dump of `--depth_format bytes` (eval.py:135-137).  Host-side, numpy only."""
import re
import struct
import sys
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def png_bytes(img, level=6):
    """(H, W, 3) or (H, W) uint8 -> PNG file contents."""
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8:
        raise ValueError("write_png expects uint8 (eval.py:139 converts with (img*255).astype(np.uint8))")
    if img.ndim == 2:
        color, ch = 0, 1
    elif img.ndim == 3 and img.shape[2] == 3:
        color, ch = 2, 3
    elif img.ndim == 3 and img.shape[2] == 4:
        color, ch = 6, 4
    else:
        raise ValueError("write_png expects (H, W), (H, W, 3) or (H, W, 4)")
    h, w = img.shape[:2]
    rows = np.empty((h, 1 + w * ch), dtype=np.uint8)
    rows[:, 0] = 0                                           # filter type 0 (None) on every scanline
    rows[:, 1:] = img.reshape(h, w * ch)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(rows.tobytes(), level)) + _chunk(b"IEND", b"")


def write_png(path, img, level=6):
    """imageio.imwrite(path, img_uint8) of eval.py:141."""
    with open(path, "wb") as f:
        f.write(png_bytes(img, level))


def read_png(path):
    """Decoder for the files write_png produces (8-bit, non-interlaced, filter 0-4) — used by the tests."""
    data = open(path, "rb").read() if isinstance(path, str) else path
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, color = hdr[:4]
    ch = {0: 1, 2: 3, 6: 4}[color]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
    if np.any(raw[:, 0] != 0):
        raise NotImplementedError("only filter 0 is decoded")
    out = raw[:, 1:].reshape(h, w, ch)
    return out[:, :, 0].copy() if ch == 1 else out.copy()


def save_pfm(filename, image, scale=1):
    """datasets/depth_utils.py:43-69: bottom-to-top float32 rows, header 'Pf' (grey) / 'PF' (colour), negative scale =
    little-endian."""
    image = np.flipud(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if len(image.shape) == 3 and image.shape[2] == 3:
        color = True
    elif len(image.shape) == 2 or len(image.shape) == 3 and image.shape[2] == 1:
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or endian == "=" and sys.byteorder == "little":
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        f.write(np.ascontiguousarray(image).tobytes())


def read_pfm(filename):
    """datasets/depth_utils.py:5-40 -> (data, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        color = header == "PF"
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = map(int, m.groups())
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    data = np.reshape(data, (height, width, 3) if color else (height, width))
    return np.flipud(data), abs(scale)


def depth_bytes(depth):
    """eval.py:135-137 (`--depth_format bytes`): the float32 depth map's raw bytes."""
    return np.ascontiguousarray(depth, dtype=np.float32).tobytes()
