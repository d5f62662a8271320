#!/bin/bash
OUT=gpurun_out/r04g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q -s --tb=short > $OUT/pytest_heads.log 2>&1; echo "heads rc=$?"
tail -2 $OUT/pytest_heads.log
DCA_AMD_TEST_LIB=tools/_dbg/libdcahip_bwd3.so timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q -s --tb=line > $OUT/pytest_bwd3.log 2>&1; echo "bwd3 rc=$?"
tail -2 $OUT/pytest_bwd3.log
