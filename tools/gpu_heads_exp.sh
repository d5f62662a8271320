#!/bin/bash
# K-HEADS alone (byte store, C3 shape): product library against an experiment build, alternating
OUT=gpurun_out/${1:-exp}; LIBX=$2
mkdir -p $OUT
for r in 1 2 3; do
  COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed 's/^/product   /' | tee -a $OUT/heads_exp.txt
  DCA_AMD_LIB=$LIBX COMPACT=1 ONLY_FUSED=1 timeout 120 python tools/bench_heads.py 4096 20000 64 1 30 < /dev/null 2>/dev/null | grep heads_fused | sed 's/^/experiment /' | tee -a $OUT/heads_exp.txt
done
