mkdir -p gpurun_out
{
echo "== song, cycle counters (RECS_PROF build)"
ST=${ST:-12} TAIL=${TAIL:-400} bash tools/song_recsprof.sh | grep "recs<2,1>\|recs<1,1>" | sort -t: -k2 | awk '{print}' | tail -60
} > gpurun_out/jfilt_prof2.txt 2>&1
tail -50 gpurun_out/jfilt_prof2.txt
