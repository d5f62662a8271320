#!/bin/bash
# rocprofv3 kernel trace of the first-layer weight gradient alone (tools/ab_enc0_dw.py, both forms): per-kernel average durations
# of the split / product / finish kernels.     gpurun -- 'bash tools/gpu_dw_prof.sh <tag> [B]'
OUT=gpurun_out/${1:-dw}/prof; B=${2:-4096}
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o ab --output-format csv -- python tools/ab_enc0_dw.py $B > $OUT/ab.log 2> $OUT/prof.err < /dev/null; echo "prof rc=$?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cut -d, -f1-7 "$f" | head -12; else echo "no kernel stats"; tail -5 $OUT/prof.err; fi
find $OUT -name "*kernel_trace.csv" -size +8M -delete
