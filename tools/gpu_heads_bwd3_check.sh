#!/bin/bash
# Does the parity test of K-HEADS notice a build whose backward products use three bf16 products instead of six?
# (tests/test_heads_fused_gpu.py::check holds dW / dH to 1e-6 of sum|ab|; the three-product build must FAIL it.)
python - <<'PY'
import sys
sys.path.insert(0, '.')
from dca_amd import build as b
print(b.build_hip(defines=('DCA_EXP_BWD3',), out='tools/_dbg/libdcahip_bwd3.so', verbose=False))
PY
DCA_AMD_TEST_LIB=tools/_dbg/libdcahip_bwd3.so timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q -x --tb=line -k "benchmark_shape or vs_oracle" 2>&1 | tail -5
