"""Hash of the kernel sources libvrhip.so is built from (csrc/*.hip, csrc/*.h, csrc/*.inc, include/vrhip.h).

The Makefile compiles it into the library (`vrhip_build_source_hash()`), bench.py stamps it on every profile it
writes and refuses to measure a library built from other sources than the tree holds.  A build with compile flags
beyond the Makefile's defaults (tools/mkvariant.sh: -D variants for A/B runs) carries "<source hash>+<flags hash>":
the same sources, but never mistaken for the product build -- its profiles match no product run.
Run as a script: prints the hash; arguments = the extra flags."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(extra_flags=()):
    """sha256 (16 hex digits) over volumerenderercl_amd/csrc/*.hip, *.h, *.inc and the C ABI header, names
    included; with extra compile flags, "+" and 8 hex digits of their hash are appended."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "volumerenderercl_amd", "csrc")
    files = sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")) +
                   glob.glob(os.path.join(base, "*.inc")))
    files.append(os.path.join(ROOT, "include", "vrhip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    out = h.hexdigest()[:16]
    flags = [f for f in extra_flags if f]
    if flags:
        out += "+" + hashlib.sha256(" ".join(flags).encode()).hexdigest()[:8]
    return out


if __name__ == "__main__":
    print(source_hash(sys.argv[1:]))
