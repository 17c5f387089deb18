#!/usr/bin/env python3
"""Name the [object+offset] buckets of sprof.c with nm:
    python tools/ubench/sprof_resolve.py sprof.log  [dir with the objects ...]
"""
import bisect
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DIRS = sys.argv[2:] + [os.path.join(ROOT, "audiality2_amd"), os.path.join(ROOT, "oracle", "_ref"),
                       "/lib/x86_64-linux-gnu", "/opt/rocm/lib"]
tables = {}


def table(obj):
    if obj not in tables:
        syms = []
        for d in DIRS:
            p = os.path.join(d, obj)
            if os.path.exists(p):
                for flags in (["-n", "-C", "--defined-only"], ["-n", "-C", "-D", "--defined-only"]):
                    out = subprocess.run(["nm"] + flags + [p], capture_output=True, text=True).stdout
                    for ln in out.splitlines():
                        f = ln.split(None, 2)
                        if len(f) == 3 and f[1] in "tTwW":
                            syms.append((int(f[0], 16), f[2]))
                break
        tables[obj] = sorted(set(syms))
    return tables[obj]


agg = collections.Counter()
for ln in open(sys.argv[1]):
    m = re.match(r"sprof\s+([0-9.]+)%\s+\[(.+)\+([0-9a-f]+)\]", ln)
    if not m:
        m2 = re.match(r"sprof\s+([0-9.]+)%\s+(\S+)", ln)
        if m2:
            agg[m2.group(2)] += float(m2.group(1))
        continue
    pct, obj, off = float(m.group(1)), m.group(2), int(m.group(3), 16)
    t = table(obj)
    i = bisect.bisect_right(t, (off + 63, "\xff")) - 1
    agg[f"{obj}:{t[i][1][:70]}" if t and i >= 0 else f"{obj}+{off:x}"] += pct
for k, v in agg.most_common(45):
    print(f"{v:6.2f}%  {k}")
