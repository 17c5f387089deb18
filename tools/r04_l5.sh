#!/bin/bash
R=$PWD/oracle/_ref/ref_render; U=$PWD/audiality2_amd/liba2amd_units.so
cd tools/_l5
for f in *.a2s; do
  for cfg in "44100 37" "48000 64"; do
    set -- $cfg
    fr=$(( $1*3/2/$2*$2 ))
    A2REF_SINK=1 $R $f Main $fr $2 $1 2 /tmp/c.pcm 0.15 >/dev/null 2>&1
    A2REF_SINK=1 LD_PRELOAD=$U $R $f Main $fr $2 $1 2 /tmp/u.pcm 0.15 >/dev/null 2>&1
    echo "$f $cfg: $(cmp /tmp/c.pcm /tmp/u.pcm 2>&1 | head -1)"
  done
done
