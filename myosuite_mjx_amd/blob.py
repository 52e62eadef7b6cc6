"""Versioned binary "model blob": the one artefact both the C oracle and the HIP
library load (`myo_model_load`, include/myo_hip.h).  Layout (little endian):

    char[4]  magic "MYOB"      uint32 version      uint32 narrays      uint32 pad
    narrays x { char[32] name; uint32 dtype (0=f64, 1=i32); uint32 ndim;
                uint32 shape[4]; uint64 nbytes; uint64 offset }         (72 bytes each)
    data, each array 8-byte aligned at its absolute offset

Arrays carry MuJoCo `mjModel` field names (see mjcf.py).  Name strings for
bodies/joints/... travel in a JSON side-car held by the Python wrapper only.
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"MYOB"
VERSION = 3
_REC = struct.Struct("<32sII4IQQ")


def pack(arrays: dict) -> bytes:
    recs, datas = [], []
    off = 16 + _REC.size * len(arrays)
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        if a.dtype.kind == "f":
            a, dt = a.astype("<f8"), 0
        elif a.dtype.kind in "iub":
            a, dt = a.astype("<i4"), 1
        else:
            raise TypeError(f"{name}: dtype {a.dtype}")
        if a.ndim > 4 or len(name) > 31:
            raise ValueError(name)
        off = (off + 7) & ~7
        shape = list(a.shape) + [1] * (4 - a.ndim)
        recs.append(_REC.pack(name.encode(), dt, a.ndim, *shape, a.nbytes, off))
        datas.append((off, a.tobytes()))
        off += a.nbytes
    out = bytearray(off)
    out[:16] = struct.pack("<4sIII", MAGIC, VERSION, len(arrays), 0)
    p = 16
    for r in recs:
        out[p:p + _REC.size] = r
        p += _REC.size
    for o, d in datas:
        out[o:o + len(d)] = d
    return bytes(out)


def unpack(buf: bytes) -> dict:
    magic, ver, n, _ = struct.unpack_from("<4sIII", buf, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError("not a MYOB v%d blob" % VERSION)
    out = {}
    for i in range(n):
        name, dt, nd, s0, s1, s2, s3, nb, off = _REC.unpack_from(buf, 16 + i * _REC.size)
        shape = (s0, s1, s2, s3)[:nd]
        a = np.frombuffer(buf, dtype="<f8" if dt == 0 else "<i4", count=nb // (8 if dt == 0 else 4), offset=off)
        out[name.rstrip(b"\0").decode()] = a.reshape(shape).copy()
    return out
