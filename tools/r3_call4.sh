#!/bin/bash
set -u
OUT=gpurun_out/r3c4; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest.log 2>&1; tail -30 $OUT/pytest.log
