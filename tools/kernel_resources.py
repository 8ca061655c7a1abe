#!/usr/bin/env python3
"""VGPR / spill / scratch figures of the kernels in a HIP library or object: unpacks the clang offload bundles
of the file and reads the code-object metadata with llvm-readelf.
    python tools/kernel_resources.py noble-curves_amd/libncg.so [substring ...]"""
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def bundles(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "amdgcn" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += 24


def main():
    blob = open(sys.argv[1], "rb").read()
    pats = sys.argv[2:]
    for co in bundles(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            if pats and not any(p in dem for p in pats):
                continue
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]  # noqa: E731
            print("%-90s vgpr %3s spill %3s scratch %4s sgpr %3s" % (dem.split("(")[0][-90:], g("vgpr_count"), g("vgpr_spill_count"),
                                                                      g("private_segment_fixed_size"), g("sgpr_count")))


if __name__ == "__main__":
    main()
