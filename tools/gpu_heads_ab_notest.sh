#!/bin/bash
# as gpu_heads_ab.sh without the parity tests (experiment builds that are WRONG on purpose: they price a latency)
OUT=gpurun_out/${1:-ab}; shift
mkdir -p $OUT
for r in 1 2; do
  COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed 's/^/product   /' | tee -a $OUT/heads_ab.txt
  for L in "$@"; do
    DCA_AMD_LIB=$L COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed "s|^|$(basename $L) |" | tee -a $OUT/heads_ab.txt
  done
done
