"""Hash of the kernel sources libvrhip.so is built from (csrc/*.hip, csrc/*.h, include/vrhip.h).

The Makefile compiles it into the library (`vrhip_build_source_hash()`), bench.py stamps it on every profile it
writes and refuses to measure a library built from other sources than the tree holds.  Run as a script: prints it."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """sha256 (16 hex digits) over volumerenderercl_amd/csrc/*.hip, *.h and the C ABI header, names included."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "volumerenderercl_amd", "csrc")
    files = sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")))
    files.append(os.path.join(ROOT, "include", "vrhip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
