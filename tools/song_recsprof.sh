# cycle counts inside k_leaf_recs_all on the song (backend built with -DRECS_PROF into tools/ubench/variants/)
REPO=$GRAFT_REPO_ROOT
cd $REPO/tests/a2s
P=$REPO/oracle/_ref/a2play
U=$REPO/tools/ubench/variants/liba2amd_units.so   # (build.py's two commands with -DRECS_PROF, outputs into that directory)

export LD_LIBRARY_PATH=$REPO/oracle/_ref:$LD_LIBRARY_PATH
LD_PRELOAD=$U timeout 120 $P -dbuffer -r44100 song.a2s -pSong -st${ST:-30} 2>&1 | grep "k_leaf_recs" | tail -${TAIL:-12}
