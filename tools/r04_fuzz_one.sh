#!/bin/bash
# usage: r04_fuzz_one.sh seed rate buffer channels [env...]
seed=$1; rate=$2; buf=$3; ch=$4; shift 4
O=$PWD/gpurun_out/fz; mkdir -p $O
python tests/fuzz_scripts.py $seed > $O/fuzz$seed.a2s
frames=$(( (rate*3/2) / buf * buf ))
R=$PWD/oracle/_ref/ref_render
U=$PWD/audiality2_amd/liba2amd_units.so; W=$PWD/audiality2_amd/liba2amd_walk.so
cd $O
env "$@" $R fuzz$seed.a2s Main $frames $buf $rate $ch cpu.pcm 0.15 2>&1 | tail -3
env "$@" LD_PRELOAD=$U $R fuzz$seed.a2s Main $frames $buf $rate $ch units.pcm 0.15 2>&1 | tail -3
env "$@" A2AMD_WALK_STATS=1 LD_PRELOAD="$W $U" $R fuzz$seed.a2s Main $frames $buf $rate $ch walk.pcm 0.15 2>&1 | tail -3
python - <<PY
import numpy as np
a=np.fromfile('cpu.pcm',dtype='<i4')
for n in ('units','walk'):
    b=np.fromfile(n+'.pcm',dtype='<i4')
    bad=np.nonzero(a!=b)[0]
    print(n, len(bad), 'differ', bad[:5], (bad[:5]//($ch*$buf)) if len(bad) else '')
PY
rm -f *.pcm
