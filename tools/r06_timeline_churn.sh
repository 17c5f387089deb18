#!/bin/bash
# round 6: the churn cell's kernel timeline (configs[2]'s 16 384 sustained voices + 200 note births and deaths a second),
# every kernel and copy of the last buffers, and the drop-in's own split of a buffer's time
cd ${GRAFT_REPO_ROOT:-$(pwd)}
REPO=$PWD
export TMPDIR=/tmp
pre="$PWD/audiality2_amd/liba2amd_walk.so $PWD/audiality2_amd/liba2amd_units.so"
( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=4096 A2AMD_HOSTTIMING=1 timeout 120 ../../oracle/_ref/ref_bench bench.a2s OscFilterPanChurn 16384 6144 1 2>&1 | grep -v "^REC" | tail -12 | cut -c1-400 )
rm -rf /tmp/prof_t
( cd tests/a2s; LD_PRELOAD="$pre" A2REF_BUFFER=4096 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_t -- \
    ../../oracle/_ref/ref_bench bench.a2s OscFilterPanChurn 16384 3072 1 > /tmp/prof_t.log 2>&1 )
grep voice_samples /tmp/prof_t.log | cut -c1-300
f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
# buffers: a gap of more than 150 us without any kernel starts a new one (the engine's walk)
bufs = [0]
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 8):i]) > 150000:
        bufs.append(i)
print("kernels", len(rows), "buffers", len(bufs), "periods us (last 10):", [round((rows[b][0] - rows[a][0]) / 1e3) for a, b in zip(bufs[-11:-1], bufs[-10:])])
for k in range(max(0, len(bufs) - 3), len(bufs) - 1):
    a, b = bufs[k], bufs[k + 1]
    t0 = rows[a][0]
    print("-- buffer", k, "kernels", b - a)
    for s, e, name, q in rows[a:b]:
        print("  %-30s q%-3s %8.1f -> %8.1f  (%7.1f)" % (name[:30], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
