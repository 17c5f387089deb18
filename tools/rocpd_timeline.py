#!/usr/bin/env python3
"""The last dispatches of a rocprofv3 run (default rocpd / sqlite output) in start order, with the gap to the kernel before:
    python tools/rocpd_timeline.py <output dir> [how many = 40]"""
import glob
import os
import sqlite3
import sys


def main():
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for dbf in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
        db = sqlite3.connect(dbf)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
        if not kd or not ks:
            continue
        cols = [r[1] for r in cur.execute(f"pragma table_info({kd[0]})")]
        q = (f"select s.kernel_name, d.start, d.end{', d.queue_id' if 'queue_id' in cols else ''} from {kd[0]} d "
             f"join {ks[0]} s on d.kernel_id=s.id order by d.start desc limit {n}")
        rows = list(cur.execute(q))[::-1]
        t0, last_end = rows[0][1], None
        for r in rows:
            name, st, en = r[0].split("(")[0][:44], r[1], r[2]
            gap = "" if last_end is None else "%+7.1f" % ((st - last_end) / 1e3)
            print("%9.1f %8.1f us  gap %8s  q%s  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, r[3] if len(r) > 3 else "", name))
            last_end = en if last_end is None else max(last_end, en)


if __name__ == "__main__":
    main()
