"""SHA-256 of the DEVICE code of a built libmeao_hip.so (its .hip_fatbin section: the gfx950 code objects of every kernel).

Committed counter evidence (profiles/pmc_traffic.json: HBM bytes and VALU instructions per launch from rocprofv3 --pmc passes) is
only valid for the kernels it was collected from: tools/make_pmc_traffic.py stores this hash, bench.py compares it with the
library it loaded and marks the evidence stale when they differ.  Host-only changes (meao_api.cpp) do not change the hash."""
from __future__ import annotations

import hashlib
import struct


def fatbin_bytes(path: str) -> bytes:
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError(f"{path}: not a little-endian ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)

    def section(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = section(shstrndx)
    names = data[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = section(i)
        if names[name:names.index(b"\0", name)] == b".hip_fatbin":
            return data[off:off + size]
    raise ValueError(f"{path}: no .hip_fatbin section")


def device_code_sha256(path: str) -> str:
    return hashlib.sha256(fatbin_bytes(path)).hexdigest()


if __name__ == "__main__":
    import sys
    from . import _lib
    print(device_code_sha256(sys.argv[1] if len(sys.argv) > 1 else _lib.LIB_PATH))
