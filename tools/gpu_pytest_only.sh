#!/bin/bash
# the parity suite alone (log under gpurun_out/<tag>/pytest_gpu.log)
OUT=gpurun_out/${1:-pt}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q < /dev/null > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log | cut -c1-200
