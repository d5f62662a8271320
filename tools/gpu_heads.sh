#!/bin/bash
TAG=${1:-heads}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q --tb=short > $OUT/pytest_heads.log 2>&1; echo "heads rc=$?"
grep -E "^(FAILED|PASSED|ERROR)|passed|failed|AssertionError|assert " $OUT/pytest_heads.log | cut -c1-400 | head -40
timeout 300 python tools/bench_heads.py 2>&1 | tail -3
timeout 300 python tools/bench_heads.py 32 20000 64 1 50 2>&1 | tail -2
