#!/bin/bash
# round 3, first GPU pass: new parity tests, the bench line with engine_in_loop, configs[4]
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "config4" > $O/t_cfg4.log 2>&1; echo "cfg4 tests rc $?" >> $O/summary
python -m pytest tests/test_dropin.py -q -x -k "engine_in_loop_at" > $O/t_engine.log 2>&1; echo "engine tests rc $?" >> $O/summary
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/summary
python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench4 rc $?" >> $O/summary
python bench.py --config 4 --voices 262144 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg4_all.json 2> $O/bench_cfg4_all.err; echo "bench4all rc $?" >> $O/summary
cat $O/summary
tail -3 $O/t_cfg4.log $O/t_engine.log
