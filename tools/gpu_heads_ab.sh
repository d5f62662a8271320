#!/bin/bash
# K-HEADS alone (byte store, C3 shape): the product library against experiment builds, alternating on ONE box (boxes differ by
# +-4 %).  usage: gpu_heads_ab.sh TAG lib1.so [lib2.so ...]
OUT=gpurun_out/${1:-ab}; shift
mkdir -p $OUT
timeout 600 python -m pytest tests/test_heads_fused_gpu.py -q -x 2>&1 | tail -3
for r in 1 2 3; do
  COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed 's/^/product   /' | tee -a $OUT/heads_ab.txt
  for L in "$@"; do
    DCA_AMD_LIB=$L COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed "s|^|$(basename $L) |" | tee -a $OUT/heads_ab.txt
  done
done
