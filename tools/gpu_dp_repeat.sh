#!/bin/bash
# the data-parallel GPU tests several times in a row (capture of the RCCL exchanges is the timing-sensitive part)
OUT=gpurun_out/${1:-dprep}; mkdir -p $OUT
for i in 1 2 3; do
  timeout 400 python -m pytest tests/test_dp_gpu.py -q -m gpu < /dev/null > $OUT/dp_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $OUT/dp_$i.log | cut -c1-120)"
done
