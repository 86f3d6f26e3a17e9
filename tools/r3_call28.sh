#!/bin/bash
# last call of the round: build check on the box, smoke(), the whole -m gpu suite, the default bench line
set -u
OUT=gpurun_out/r3c28; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
( time timeout 1500 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; head -c 600 $OUT/bench_default.json; echo; tail -4 $OUT/bench_default.err
