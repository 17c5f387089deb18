#!/bin/bash
# configs[1]: where a 69-74 us step goes (graph or plain launches; the stream's timeline), and the filter kernel's cycle counters
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
for e in "X=1" "A2AMD_NO_GRAPH=1"; do env $e python bench.py --config 1 --steps 400 --warmup 20 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['parity_vs_golden'])"; done
A2AMD_LIB=$PWD/tools/ubench/variants/liba2amd_prof.so python bench.py --config 2 --steps 12 --warmup 4 --no-extra --no-cpu-baseline --no-engine --no-realtime 2>&1 | grep "^block" | tail -n 34 > $O/filt_prof.txt; cat $O/filt_prof.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --config 1 --steps 60 --warmup 20 --no-extra --no-cpu-baseline --no-engine --no-realtime > /tmp/tr.log 2>&1
python - <<'PY'
import csv,glob
ev=[]
for f in glob.glob('/tmp/tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]))
for f in glob.glob('/tmp/tr/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY "+r.get("Direction","")[:20]))
ev.sort()
prev=None
for s,e,n in ev[-70:]:
    print(n.ljust(32), "dur %6.1f us"%((e-s)/1e3), "gap %6.1f us"%(((s-prev)/1e3) if prev else 0))
    prev=e
PY
