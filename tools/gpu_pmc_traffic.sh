#!/bin/bash
# HBM-side traffic of K-HEADS from the TCC counters (separate --pmc passes, kernel trace only), stamped with the fingerprint
# of the kernel sources; fold the result into profiles/pmc_traffic.json with  python tools/pmc_traffic_update.py --merge <dir>
TAG=${1:-traffic}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  COMPACT=1 ONLY_FUSED=1 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/heads_$c -o pmc --output-format csv -- python tools/bench_heads.py 4096 20000 64 1 3 > $OUT/heads_$c.log 2>&1; echo "heads $c rc=$?"
done
python tools/pmc_traffic_update.py $OUT
