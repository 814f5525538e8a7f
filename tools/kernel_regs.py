#!/usr/bin/env python3
"""tools/kernel_regs.py FILE.s [FILTER...] -- registers, spills, scratch and LDS of the kernels in the device assembly of
a .hip file (hipcc ... --cuda-device-only -S FILE.hip -o FILE.s), from the code object's metadata."""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
meta = t[t.index("amdhsa.kernels"):]
flt = sys.argv[2:] or [""]
for k in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    def g(key):
        m = re.search(r"\.%s:\s+(\S+)" % key, k)
        return m.group(1) if m else "?"
    n = g("name")
    if not any(f in n for f in flt):
        continue
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        d = n
    d = re.sub(r"\(VolView.*", "", d).replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-96s vgpr %3s agpr %3s spills %3s scratch %4s B  lds %6s B  sgpr %3s" % (
        d[:96], g("vgpr_count"), k.split()[0], g("vgpr_spill_count"), g("private_segment_fixed_size"),
        g("group_segment_fixed_size"), g("sgpr_count")))
