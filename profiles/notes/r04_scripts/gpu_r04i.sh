#!/bin/bash
OUT=gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" tools/_dbg/libdcahip_DCA_EXP_PIN_Z.so; do
  echo "== lib ${lib:-product}"
  DCA_AMD_LIB=$lib COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 1 50 2>&1 | tail -2
  DCA_AMD_LIB=$lib COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/bench_heads.py 4096 20000 64 3 50 2>&1 | tail -2 | head -1
done
done
DCA_AMD_TEST_LIB=tools/_dbg/libdcahip_DCA_EXP_PIN_Z.so timeout 600 python -m pytest tests/test_heads_fused_gpu.py tests/test_golden_gpu.py -x -q --tb=short 2>&1 | tail -2
timeout 300 python tools/b32_probe.py > $OUT/b32_probe.txt 2>&1; grep -E "steps per graph|summary|cold|clocks at" $OUT/b32_probe.txt | cut -c1-260
