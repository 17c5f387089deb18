mkdir -p gpurun_out
cd tests/a2s
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/oracle/_ref:$LD_LIBRARY_PATH
{
for i in 1 2; do
A2AMD_HOSTTIMING=1 LD_PRELOAD=$GRAFT_REPO_ROOT/audiality2_amd/liba2amd_units.so bash -c 'time $GRAFT_REPO_ROOT/oracle/_ref/a2play -dbuffer -r44100 song.a2s -pSong -st200' 2>&1 | grep -v "^$" | tail -6
done
A2AMD_PROFILE=1 A2AMD_HOSTTIMING=1 LD_PRELOAD=$GRAFT_REPO_ROOT/audiality2_amd/liba2amd_units.so $GRAFT_REPO_ROOT/oracle/_ref/a2play -dbuffer -r44100 song.a2s -pSong -st60 2>&1 | tail -12
} > $GRAFT_REPO_ROOT/gpurun_out/song_host.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/song_host.txt
