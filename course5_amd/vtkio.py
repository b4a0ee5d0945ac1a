"""Minimal readers for checking the `course` CLI's output files (host tooling).

read_vti: VTK XML ImageData as written by course5_amd/csrc/host/vtk_io.cpp (appended raw,
UInt64 header) -> float64 array [res_y, res_x, n_components].  Array name, type and component
count are returned so tests can check the object2d.cpp:12-13 contract (ImageScalars, Float64 x 2).
"""
from __future__ import annotations

import base64
import re
import struct
import zlib

import numpy as np


def read_vti(path: str):
    raw = open(path, "rb").read()
    head_end = raw.index(b"<AppendedData")
    head = raw[:head_end].decode()
    ext = [int(v) for v in re.search(r'WholeExtent="([^"]+)"', head).group(1).split()]
    res_x, res_y = ext[1] - ext[0] + 1, ext[3] - ext[2] + 1
    arr = re.search(r'<DataArray type="(\w+)" Name="(\w+)" NumberOfComponents="(\d+)" format="appended" offset="(\d+)"', head)
    dtype, name, ncomp, offset = arr.group(1), arr.group(2), int(arr.group(3)), int(arr.group(4))
    assert re.search(r'header_type="UInt64"', head)
    np_type = {"Float64": "<f8", "Float32": "<f4"}[dtype]
    start = raw.index(b"_", head_end) + 1 + offset
    if b'encoding="raw"' in raw[head_end:head_end + 40]:
        (n_bytes,) = struct.unpack("<Q", raw[start:start + 8])
        data = np.frombuffer(raw, dtype=np_type, count=n_bytes // np.dtype(np_type).itemsize, offset=start + 8)
    else:  # base64 + vtkZLibDataCompressor blocks
        assert 'compressor="vtkZLibDataCompressor"' in head
        end = raw.index(b"<", start)
        text = raw[start:end].strip()
        first = base64.b64decode(text[:32])  # 3 x UInt64 = 24 bytes = exactly 32 base64 characters
        n_blocks, block, last = struct.unpack("<QQQ", first)
        hdr_len = -(-(3 + n_blocks) * 8 // 3) * 4  # base64 length of the whole header, padded
        sizes = struct.unpack(f"<{3 + n_blocks}Q", base64.b64decode(text[:hdr_len]))[3:]
        body = base64.b64decode(text[hdr_len:])
        out, pos = [], 0
        for sz in sizes:
            out.append(zlib.decompress(body[pos:pos + sz]))
            pos += sz
        blob = b"".join(out)
        assert len(blob) == (n_blocks - 1) * block + (last or block)
        data = np.frombuffer(blob, dtype=np_type)
    info = dict(name=name, type=dtype, components=ncomp, dims=(res_x, res_y, ext[5] - ext[4] + 1),
                origin=re.search(r'Origin="([^"]+)"', head).group(1), spacing=re.search(r'Spacing="([^"]+)"', head).group(1))
    return data.reshape(res_y, res_x, ncomp), info


def read_png(path: str) -> np.ndarray:
    """Decode an 8-bit RGB, non-interlaced PNG (what `course --png` writes) into uint8 [rows][cols][3], first
    row = top scanline.  All five scanline filters are undone; chunk CRCs and the zlib checksum are verified."""
    raw = open(path, "rb").read()
    if raw[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    off, idat, shape = 8, b"", None
    while off < len(raw):
        (n,), kind = struct.unpack(">I", raw[off:off + 4]), raw[off + 4:off + 8]
        body = raw[off + 8:off + 8 + n]
        (crc,) = struct.unpack(">I", raw[off + 8 + n:off + 12 + n])
        if zlib.crc32(kind + body) != crc:
            raise ValueError(f"PNG chunk {kind!r}: CRC mismatch")
        if kind == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, colour, interlace) != (8, 2, 0):
                raise ValueError("only 8-bit RGB without interlacing is read here")
            shape = (h, w)
        elif kind == b"IDAT":
            idat += body
        elif kind == b"IEND":
            break
        off += 12 + n
    h, w = shape
    data = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), dtype=np.uint8)
    prev = np.zeros(3 * w, dtype=np.int64)
    for r in range(h):
        f, line = int(data[r, 0]), data[r, 1:].astype(np.int64)
        if f == 0:
            cur = line
        elif f == 1:  # Sub: running sum per colour component
            cur = np.cumsum(line.reshape(w, 3), axis=0).reshape(-1) & 255
        elif f == 2:  # Up
            cur = (line + prev) & 255
        else:  # Average / Paeth: byte by byte
            cur = np.zeros(3 * w, dtype=np.int64)
            for k in range(3 * w):
                a = cur[k - 3] if k >= 3 else 0
                b = prev[k]
                c = prev[k - 3] if k >= 3 else 0
                if f == 3:
                    pred = (a + b) // 2
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[k] = (line[k] + pred) & 255
        out[r] = cur
        prev = cur
    return out.reshape(h, w, 3)
