#!/bin/bash
# differential soak of the round's final library (the filter window along the lanes): random scripts in seed ranges no
# earlier soak used, 50 seeds per run so that a deadline leaves whole runs behind; walk + device VM, then units alone
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/final_soak_jfilt.txt; : > $O
END=$(( $(date +%s) + ${BUDGET:-760} ))
run() { [ $(date +%s) -ge $END ] && return; echo "== $*" >> $O; "$@" 2>&1 | tail -2 >> $O; }
for a in 1500 2000 3750 1550 2050 3800 1600 2100 3850; do A2FUZZ_WALK=1 run python tests/measure/fuzz_soak.py $a $((a + 50)); done
for a in 1650 3000 2150 1700 3050; do run python tests/measure/fuzz_soak.py $a $((a + 50)); done
cat $O
