#!/usr/bin/env python3
"""Where a kernel's spilled registers are touched: every scratch load / store of the kernel's gfx950 ISA with the
loop depth of the block it stands in (LLVM's own loop comments in the assembly listing).

    python tools/spill_sites.py [source=audiality2_amd/csrc/a2amd_fast.hip] [kernel name prefix=_Z17k_leaf_oscfiltpan]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "audiality2_amd", "csrc", "a2amd_fast.hip")
    pref = sys.argv[2] if len(sys.argv) > 2 else "_Z17k_leaf_oscfiltpan"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S",
                        src, "-o", out], check=True, capture_output=True, cwd=os.path.dirname(src))
        L = open(out).read().split("\n")
    st = next(i for i, l in enumerate(L) if l.startswith(pref))
    en = next(i for i, l in enumerate(L[st:]) if l.startswith(".Lfunc_end")) + st
    body = L[st:en]
    depth, header = 0, None
    rows = []
    ninstr = 0
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            d = re.search(r"Depth[= ](\d+)", l)
            h = re.search(r"Header[=:]\s*(BB\d+_\d+)", l)
            if "Loop" in l and d:
                depth = int(d.group(1))
                header = h.group(1) if h else m.group(1)[2:]
            else:
                depth, header = 0, None
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ninstr += 1
            if t.startswith("scratch_"):
                rows.append((ninstr, depth, header, t.split(";")[0].strip(), "reload" if "load" in t.split()[0] else "spill"))
    print(f"{pref}: {ninstr} instructions, {len(rows)} scratch accesses")
    by = collections.Counter((r[1], r[4]) for r in rows)
    for (d, k), n in sorted(by.items()):
        print(f"  loop depth {d}: {n} {k}s")
    print("  instruction #, loop depth, innermost loop header, instruction")
    for r in rows:
        print(f"  {r[0]:6d}  depth {r[1]}  {r[2] or '-':12s}  {r[3]}")


if __name__ == "__main__":
    main()
