#!/bin/bash
OUT=gpurun_out/icache
mkdir -p $OUT
export TMPDIR=/tmp
for v in slow vc; do
  DCA_AMD_LIB=$PWD/tools/_dbg/$v.so timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_IFETCH -d $OUT/$v -o pmc --output-format csv -- python tools/bench_heads.py 4096 20000 64 1 2 > $OUT/$v.log 2>&1; echo "$v rc=$?"; tail -2 $OUT/$v.log
done
