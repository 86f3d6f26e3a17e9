#!/bin/bash
# GPU call 3: the whole -m gpu suite (new: full-length goldens, N1 end-to-end, paintbrush, empty shards), bench smoke of the new default
set -u
OUT=gpurun_out/r3c3; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
timeout 600 python bench.py --workload c2 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 1500 $OUT/bench_c2.json
timeout 900 python bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_c2_256.json 2> $OUT/bench_c2_256.err; tail -c 1500 $OUT/bench_c2_256.json
