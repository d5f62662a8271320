#!/bin/bash
# round 4, first GPU call: the wave-specialised K-HEADS against the oracle and against the round-3 kernel
OUT=gpurun_out/r04a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -x -q --tb=short > $OUT/pytest_heads.log 2>&1; echo "heads+golden rc=$?"
tail -15 $OUT/pytest_heads.log | cut -c1-300
for v in ps x3; do
  echo "== variant $v"
  DCAHIP_HEADS_VARIANT=$v COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 1 50 2>&1 | tail -2
  DCAHIP_HEADS_VARIANT=$v COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 3 50 2>&1 | tail -2
  DCAHIP_HEADS_VARIANT=$v COMPACT=0 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 1 50 2>&1 | tail -2
done
COMPACT=1 timeout 300 python tools/timing_heads_ps.py 4096 20000 64 1 20 > $OUT/timing_ps.txt 2>&1; tail -16 $OUT/timing_ps.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json
DCAHIP_HEADS_VARIANT=x3 timeout 600 python bench.py > $OUT/bench_x3.json 2> $OUT/bench_x3.err; echo "bench x3 rc=$?"; cut -c1-600 $OUT/bench_x3.json
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -5 $OUT/pytest_gpu.log | cut -c1-300
