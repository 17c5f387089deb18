#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 run that wrote its default (rocpd / sqlite) output:
    python tools/rocpd_kernels.py <output dir> [label]   ->  one JSON line"""
import glob
import json
import os
import sqlite3
import sys


def kernels(path):
    out = {}
    for dbf in glob.glob(os.path.join(path, "**", "*.db"), recursive=True):
        db = sqlite3.connect(dbf)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
        if not kd or not ks:
            continue
        q = (f"select s.kernel_name, count(*), avg(d.end-d.start), max(d.end-d.start), min(d.end-d.start) from {kd[0]} d "
             f"join {ks[0]} s on d.kernel_id=s.id group by s.kernel_name")
        for name, n, avg, mx, mn in cur.execute(q):
            short = name.split("(")[0].replace(".kd", "")
            e = out.setdefault(short, {"calls": 0, "total_us": 0.0, "max_us": 0.0, "min_us": 1e30})
            e["calls"] += n
            e["total_us"] += avg * n / 1e3
            e["max_us"] = max(e["max_us"], mx / 1e3)
            e["min_us"] = min(e["min_us"], mn / 1e3)
    return out


if __name__ == "__main__":
    k = kernels(sys.argv[1])
    top = sorted(k.items(), key=lambda kv: -kv[1]["total_us"])[:12]
    print(json.dumps({"label": sys.argv[2] if len(sys.argv) > 2 else "", "kernels": {
        n: {"calls": v["calls"], "avg_us": round(v["total_us"] / v["calls"], 1), "max_us": round(v["max_us"], 1),
            "min_us": round(v["min_us"], 1)} for n, v in top}}))
