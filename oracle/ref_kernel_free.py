#!/usr/bin/env python3
"""Which functions of the REFERENCE's compiled kernel object (oracle/_ref/ref_kernel.o =
/root/reference/src/kernel/volumeraycast.cl through `clang -x cl`, `make -C oracle ref`) can be
executed without the OpenCL C builtin library?  Walks the object's relocations: a function is FREE
when neither it nor anything it calls references an undefined symbol.  Those are the only pieces of the
kernel file the oracle can be pinned to bit for bit (tests/test_oracle_golden.py); everything else needs
builtins this image lacks and stays unpinned.  TEST INFRASTRUCTURE ONLY."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def analyse(obj):
    dis = subprocess.run([OBJDUMP, "-dr", "--no-show-raw-insn", obj], capture_output=True, text=True,
                         check=True).stdout
    refs, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1)
            refs[cur] = set()
            continue
        m = re.search(r"R_X86_64_\w+\s+(\S+?)([-+]0x[0-9a-f]+)?$", line.strip())
        if m and cur and not m.group(1).startswith("."):
            refs[cur].add(m.group(1))
    out = {}
    for f in refs:
        seen, todo = set(), [f]
        while todo:
            for sym in refs.get(todo.pop(), ()):
                if sym not in seen:
                    seen.add(sym)
                    if sym in refs:
                        todo.append(sym)
        out[f] = sorted(x for x in seen if x not in refs)
    return out


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "_ref", "ref_kernel.o")
    for f, undefined in analyse(obj).items():
        print("%-40s %s" % (f, "FREE" if not undefined else "needs %d builtins" % len(undefined)))
