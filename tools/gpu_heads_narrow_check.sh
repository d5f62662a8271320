#!/bin/bash
# Do the parity tests of K-HEADS notice a build that spends TWO fp16 products per fp32 product instead of three (the second
# piece of one operand dropped: 2^-11 relative instead of 2^-22)?  Every case must FAIL; the product library must pass.
OUT=gpurun_out/${1:-narrow}; mkdir -p $OUT
{
echo "== product library"
timeout 600 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -q --tb=no 2>&1 | tail -3
echo "== -DDCA_EXP_H2_TWO (tools/_dbg/libdcahip_h2two.so)"
DCA_AMD_TEST_LIB=tools/_dbg/libdcahip_h2two.so timeout 600 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -q --tb=no 2>&1 | tail -45
} | tee $OUT/heads_narrow_check.txt
