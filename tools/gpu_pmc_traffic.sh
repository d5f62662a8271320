#!/bin/bash
# HBM traffic of the dominant kernel from TCC counters (separate passes), plus a calibration run
TAG=${1:-traffic}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/calib_$c -o pmc --output-format csv -- ./tools/_dbg/pmc_calibration > $OUT/calib_$c.log 2>&1; echo "calib $c rc=$?"
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/heads_$c -o pmc --output-format csv -- python tools/bench_heads.py 4096 20000 64 1 3 > $OUT/heads_$c.log 2>&1; echo "heads $c rc=$?"
done
