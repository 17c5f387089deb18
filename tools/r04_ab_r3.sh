#!/bin/bash
# one fuzz seed: CPU engine vs this round's units vs round 3's units (tools/_r3, not committed)
seed=$1; rate=$2; buf=$3; ch=$4; shift 4
O=/tmp/ab; mkdir -p $O
python tests/fuzz_scripts.py $seed > $O/f.a2s
frames=$(( (rate*3/2) / buf * buf ))
R=$PWD/oracle/_ref/ref_render
cd $O
env "$@" $R f.a2s Main $frames $buf $rate $ch cpu.pcm 0.15 >/dev/null 2>&1
env "$@" LD_PRELOAD=$GRAFT_REPO_ROOT/audiality2_amd/liba2amd_units.so $R f.a2s Main $frames $buf $rate $ch now.pcm 0.15 >/dev/null 2>&1
env "$@" LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_r3 LD_PRELOAD=$GRAFT_REPO_ROOT/tools/_r3/liba2amd_units.so $R f.a2s Main $frames $buf $rate $ch r3.pcm 0.15 2>&1 | tail -2
cmp cpu.pcm now.pcm && echo "now: same"; cmp cpu.pcm r3.pcm && echo "r3: same"
