#!/bin/bash
# one fuzz seed, the engine's CPU units vs the drop-in behind the walk with parts of it switched off
cd $GRAFT_REPO_ROOT
S=${SEED:-0}
python - <<PY
import sys; sys.path.insert(0,'tests')
from fuzz_scripts import make_script
open('/tmp/fuzz.a2s','w').write(make_script($S))
PY
cd /tmp
R=$GRAFT_REPO_ROOT/oracle/_ref/ref_render; A=$GRAFT_REPO_ROOT/audiality2_amd
run() { # label, env...
  local label=$1; shift
  env "$@" A2REF_SINK=1 A2REF_SOURCE=1 $R /tmp/fuzz.a2s Main 72000 64 48000 2 /tmp/o_$label.pcm 0.15 2>&1 | grep -v "^$" | tail -2 | sed "s/^/$label: /"
  md5sum /tmp/o_$label.pcm | cut -c1-12
}
run cpu X=1
run units LD_PRELOAD=$A/liba2amd_units.so
run walk LD_PRELOAD="$A/liba2amd_walk.so $A/liba2amd_units.so" A2AMD_WALK_STATS=1
run nohold LD_PRELOAD="$A/liba2amd_walk.so $A/liba2amd_units.so" A2AMD_WALK_NOHOLD=1
run nocache LD_PRELOAD="$A/liba2amd_walk.so $A/liba2amd_units.so" A2AMD_WALK_NOCACHE=1
run walkoff LD_PRELOAD="$A/liba2amd_walk.so $A/liba2amd_units.so" A2AMD_WALK_OFF=1
