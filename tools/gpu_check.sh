#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/<tag>/.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 3000 $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name '*kernel_stats*' | head
