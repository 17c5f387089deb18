#!/usr/bin/env python
"""Kernel-trace CSV -> per-buffer timeline.  A buffer starts at every k_vm_commit launch (one per a2_Run in the steady
scripted cells); prints the last N buffers: kernel, queue id, start / end in microseconds after the buffer's first kernel."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Queue_Id", "?")))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_vm_commit")]
# group commits that belong together (several classes): a new buffer when the gap to the previous commit is > 200 us
bufs = []
for i in starts:
    if not bufs or rows[i][0] - rows[bufs[-1]][0] > 200000:
        bufs.append(i)
print("buffers seen:", len(bufs), " period us (last 10):",
      [round((rows[b][0] - rows[a][0]) / 1e3) for a, b in zip(bufs[-11:-1], bufs[-10:])])
for k in range(max(0, len(bufs) - n - 1), len(bufs) - 1):
    a, b = bufs[k], bufs[k + 1]
    # include kernels that started shortly before the commit (the pass it waited for)
    t0 = rows[a][0]
    lo = a
    while lo > 0 and rows[lo - 1][1] > t0 - 50000:
        lo -= 1
    print("-- buffer", k)
    for s, e, name, q in rows[lo:b]:
        print("  %-28s q%-3s %8.1f -> %8.1f  (%7.1f)" % (name[:28], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
