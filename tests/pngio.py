"""Golden PNG access for the tests: PIL when it is importable, else a small stdlib reader."""
import os
import struct
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_png_stdlib(path):
    """8-bit grey / RGB, non-interlaced PNG -> uint8 [h, w, 3] with the standard library only
    (the goldens are exactly that; used when PIL is not importable on the box)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
    w, h, depth, ctype, _, _, interlace = ihdr
    assert depth == 8 and ctype in (0, 2) and interlace == 0
    bpp = 1 if ctype == 0 else 3
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:  # Sub: a prefix sum per channel
            cur = np.zeros_like(line)
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        else:  # Average / Paeth depend on the pixel just decoded: plain loop
            cur = np.zeros_like(line)
            for i in range(w * bpp):
                a = int(cur[i - bpp]) if i >= bpp else 0
                b = int(prev[i])
                c = int(prev[i - bpp]) if i >= bpp else 0
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (int(line[i]) + pred) & 255
        out[y] = cur
        prev = cur
    img = out.reshape(h, w, bpp)
    return np.repeat(img, 3, axis=2) if bpp == 1 else img


def read_golden(name):
    path = os.path.join(GOLDEN, f"{name}.png")
    try:
        from PIL import Image
    except ImportError:
        return read_png_stdlib(path)
    return np.array(Image.open(path).convert("RGB"))


