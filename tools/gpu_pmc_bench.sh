#!/bin/bash
# SQ / GRBM counters of every kernel of the bench step (separate --pmc passes, kernel-trace only).
TAG=${1:-pmcb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py $OUT $OUT/sq_summary.csv $OUT/sq_pipe_counts.json
# keep the raw rows of our kernels only (the torch data-generation kernels are not of interest)
for f in $(find $OUT -name '*counter_collection.csv'); do
  (head -1 $f; grep -E "heads_fused|heads_reduce|gemm_|splitk_reduce|zinb_nll|transpose|bn_|col_moments|rmsprop|enc0_|stack_" $f) > $f.filtered; mv $f.filtered $f
done
find $OUT -name '*kernel_trace.csv' -delete
cut -c1-200 $OUT/sq_summary.csv | head -30
