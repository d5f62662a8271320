#!/bin/bash
# Round-6 visit: fp16 x 2 plane GEMM tests, then the per-kernel trace of the configs[4] step (which launches the time goes to).
TAG=${1:-r06e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_h2_gpu.py -q 2>&1 | tail -8
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 72 --warmup 3 --no-cpu-baseline --graph off > $OUT/bench_c5_prof.json 2> $OUT/bench_c5_prof.err; echo "prof rc=$?"
f=$(find $OUT/prof_c5 -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:28]:
    print('%-90s n=%6s avg=%10.1f us tot=%9.2f ms %5s%%'%(r['Name'][:90],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,r['Percentage']))
PY
find $OUT/prof_c5 -name '*kernel_trace.csv' -delete
