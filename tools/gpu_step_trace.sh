#!/bin/bash
# The kernel sequence of one step of a workload (rocprofv3 kernel trace of a short eager bench run).  usage: gpu_step_trace.sh TAG "bench args"
TAG=${1:-trace}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o t --output-format csv -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --graph off $2 > $OUT/bench.json 2> $OUT/bench.err; echo "prof rc=$?"
f=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/step_trace.py $f 6 --big | tee $OUT/step_trace.txt
python tools/step_trace.py $f 6 --median | tee $OUT/step_trace_b32.txt
rm -rf $OUT/prof
