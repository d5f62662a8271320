#!/bin/bash
OUT=gpurun_out/r04d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -x -q --tb=short > $OUT/pytest_heads.log 2>&1; echo "heads+golden (ps3) rc=$?"
tail -6 $OUT/pytest_heads.log | cut -c1-400
for v in ps3 ps x3; do
  echo "== variant $v"
  DCAHIP_HEADS_VARIANT=$v COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 1 50 2>&1 | tail -2
  DCAHIP_HEADS_VARIANT=$v COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 3 50 2>&1 | tail -2 | head -1
done
COMPACT=1 timeout 300 python tools/timing_heads_ps.py 4096 20000 64 1 20 > $OUT/timing_ps3.txt 2>&1; tail -16 $OUT/timing_ps3.txt
