#!/bin/bash
# PMC counters for the fused heads kernel (separate passes; no tracing flags besides kernel-trace)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
  i=$((i+1))
  ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python tools/bench_heads.py 4096 20000 64 1 3 > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
find $OUT -name "*.csv" | head -20
