#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (median launch).

    python tools/pmc_summary.py <dir-with-rocprof-output> [label]

Prints one line per (kernel, counter); FETCH_SIZE / WRITE_SIZE (KB) are also
turned into HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 - the
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE
reports half of wide coalesced reads).
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


def main():
    root = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    # per (kernel, counter): dispatch id -> value summed over the rows of that dispatch
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row.get("Kernel_Name", ""))
                c = row.get("Counter_Name", "")
                try:
                    v = float(row.get("Counter_Value", "0"))
                except ValueError:
                    continue
                per[(k, c)][(fn, row.get("Dispatch_Id"))] += v
    # the kernel trace of the same run (--kernel-trace): total time per kernel, so that whoever reads the summary can
    # pick the DOMINANT kernel of a workload by time instead of guessing it from the workload's name (round 5's review)
    dur = collections.defaultdict(float)
    calls = collections.Counter()
    for fn in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                # (rocprofv3's kernel_trace.csv: Start_Timestamp / End_Timestamp, ns; other spellings tolerated - a trace
                # that cannot be read leaves the summary without times and pmc_to_json.py says so: "kernel_chosen")
                lo = {k.lower(): v for k, v in row.items() if k}
                try:
                    dt = float(lo.get("end_timestamp", lo.get("end", lo["endns"] if "endns" in lo else ""))) - \
                        float(lo.get("start_timestamp", lo.get("start", lo["startns"] if "startns" in lo else "")))
                except (KeyError, ValueError):
                    continue
                k = short(row.get("Kernel_Name", ""))
                dur[k] += dt
                calls[k] += 1
    out = {}
    for (k, c), d in sorted(per.items()):
        vals = sorted(d.values())
        # the MEDIAN launch: the set-up batch is an outlier both ways (the leaf kernels
        # find only voices with records there and do next to nothing; the general kernel
        # executes every voice's init records)
        mean = vals[len(vals) // 2] if vals else 0.0
        steady = vals
        out.setdefault(k, {})[c] = mean
        print(f"{label:>28s} {k:24s} {c:24s} {mean:16.6g} median launch of {len(vals)}")
    for k, d in out.items():
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            hbm = (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0
            d["hbm_bytes_per_launch"] = hbm
            print(f"{label:>28s} {k:24s} {'-> HBM bytes/launch':24s} {hbm:16.6g}")
    for k, d in out.items():
        if k in dur:
            d["trace_ns_total"] = dur[k]
            d["trace_launches"] = calls[k]
            print(f"{label:>28s} {k:24s} {'-> ns in this run':24s} {dur[k]:16.6g} over {calls[k]} launches")
    print("JSON " + json.dumps({label: out}))


if __name__ == "__main__":
    main()
