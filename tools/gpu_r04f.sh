#!/bin/bash
OUT=gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -q -s --tb=short > $OUT/pytest_heads.log 2>&1; echo "heads+golden rc=$?"
grep -E "backward products|passed|failed|Error|assert" $OUT/pytest_heads.log | sort | uniq -c | sort -rn | head -40
echo "== three-product build against the same tests (must fail)"
DCA_AMD_TEST_LIB=tools/_dbg/libdcahip_bwd3.so timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q -s --tb=line -k "benchmark_shape or vs_oracle" > $OUT/pytest_bwd3.log 2>&1; echo "bwd3 rc=$?"
grep -E "passed|failed" $OUT/pytest_bwd3.log | tail -2
grep -E "AssertionError" $OUT/pytest_bwd3.log | cut -c1-120 | sort | uniq -c | sort -rn | head -30
echo "== whole epoch"
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -s --tb=short -k "whole_epoch" > $OUT/pytest_epoch.log 2>&1; echo "epoch rc=$?"
grep -E "fp32 realisations|seeds \(|engine, worst|passed|failed|assert|Error" $OUT/pytest_epoch.log | cut -c1-400
echo "== C4 shard"
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -s --tb=short -k "c4_rank" > $OUT/pytest_c4.log 2>&1; echo "c4 rc=$?"
grep -E "C4 rank|passed|failed|assert|Error" $OUT/pytest_c4.log | cut -c1-500
