#!/bin/bash
# SQ / GRBM counters of the first-layer weight-gradient kernels alone (tools/ab_enc0_dw.py: both forms, one batch size); separate
# --pmc passes, kernel-trace only.      gpurun -- 'bash tools/gpu_pmc_dw.sh <tag> [B]'
TAG=${1:-pmcdw}; B=${2:-4096}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python tools/ab_enc0_dw.py $B > $OUT/p$i.log 2>&1 < /dev/null
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py $OUT $OUT/sq_summary.csv > /dev/null
for f in $(find $OUT -name '*counter_collection.csv'); do
  (head -1 $f; grep -E "enc0_" $f) > $f.filtered; mv $f.filtered $f
done
find $OUT -name '*kernel_trace.csv' -delete
python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/sq_summary.csv')))
keys = [k for k in rows[0].keys() if k != 'kernel']
for r in rows:
    if 'enc0_dw' in r['kernel']:
        print(r['kernel'])
        for k in keys:
            if r[k] != '':
                print('   %-28s %s' % (k, r[k]))
PY
