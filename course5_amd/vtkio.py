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
