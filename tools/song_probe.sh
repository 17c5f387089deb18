# where a song buffer's time goes: host split + kernel trace of the reference's player on the drop-in
REPO=$GRAFT_REPO_ROOT
cd $REPO/tests/a2s
P=$REPO/oracle/_ref/a2play
U=$REPO/audiality2_amd/liba2amd_units.so
export LD_LIBRARY_PATH=$REPO/oracle/_ref:$LD_LIBRARY_PATH
A2AMD_HOSTTIMING=1 LD_PRELOAD=$U $P -dbuffer -r44100 song.a2s -pSong -st100 2>&1 | grep a2amd
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_song
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_song -- bash -c "cd $REPO/tests/a2s; LD_PRELOAD=\"\$LD_PRELOAD:$U\" exec $P -dbuffer -r44100 song.a2s -pSong -st60" > /tmp/prof_song.log 2>&1
f=$(find /tmp/prof_song -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python3 -c "import csv,sys; [print(r[\"Name\"][:40].ljust(40), r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1), r[\"Percentage\"]) for r in csv.DictReader(open(sys.argv[1]))]" $f
