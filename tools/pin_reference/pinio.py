"""The kit's flat container for named arrays (".pin"): what tools/pin_reference/pin_harness.cpp reads and writes without any
library.  Layout (little endian): magic "RSDPIN01", uint32 count, then per array: uint32 name length, name bytes, uint32 dtype code
(0 = float64, 1 = int32, 2 = int64, 3 = uint8), uint32 ndim, ndim x uint64 dims, the data in C order."""
import struct

import numpy as np

MAGIC = b"RSDPIN01"
DTYPES = {0: np.float64, 1: np.int32, 2: np.int64, 3: np.uint8}
CODES = {np.dtype(v): k for k, v in DTYPES.items()}


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype not in CODES:
                a = a.astype(np.float64)
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<II", CODES[a.dtype], a.ndim))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        assert f.read(8) == MAGIC, "not a .pin file: " + path
        (count,) = struct.unpack("<I", f.read(4))
        for _ in range(count):
            (ln,) = struct.unpack("<I", f.read(4))
            name = f.read(ln).decode()
            code, ndim = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dQ" % ndim, f.read(8 * ndim)) if ndim else ()
            dt = np.dtype(DTYPES[code])
            n = int(np.prod(dims)) if ndim else 1
            out[name] = np.frombuffer(f.read(n * dt.itemsize), dtype=dt).reshape(dims).copy()
    return out
