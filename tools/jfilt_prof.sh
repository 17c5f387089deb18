#!/bin/bash
# cycle counters of the song's filter voices (-DRECS_PROF build in tools/ubench/variants: tools/variant_build.sh prof
# -DRECS_PROF, the units library linked against it), then the song's timing and the record paths' parity on the product
mkdir -p gpurun_out
{
echo "== song, cycle counters (RECS_PROF build)"
ST=${ST:-12} TAIL=${TAIL:-400} bash tools/song_recsprof.sh | grep "recs<2,1>" | sort -t: -k2 | tail -12
echo "== parity of the record paths"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "records or traces or fuzz or partial or mixed_quiet" 2>&1 | tail -3
echo "== song"; timeout 600 python tests/measure/song_timing.py --seconds 500 2>&1 | tail -2
} > gpurun_out/jfilt_prof2.txt 2>&1
cat gpurun_out/jfilt_prof2.txt
