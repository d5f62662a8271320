#!/bin/bash
# rocprofv3 kernel trace of the fused heads alone (bench_heads.py): per-kernel average durations
TAG=${1:-hprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o hb -- python tools/bench_heads.py 4096 20000 64 1 30 > $OUT/bench.log 2>&1
DB=$(find $OUT/prof -name '*.db' | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.csv >/dev/null 2>&1 || python tools/rocpd_stats.py $DB > $OUT/kernel_stats.csv
grep -E "heads_" $OUT/kernel_stats.csv | cut -c1-60,100-400
rm -rf $OUT/prof
