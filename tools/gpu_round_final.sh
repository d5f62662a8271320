#!/bin/bash
# The measurements a round is judged on, in one visit: parity tests, the bench line (c3) and the two 8-GPU workloads as one
# rank sees them (c4, c5), rocprofv3 kernel stats of the bench command, SQ counters per kernel, HBM-side traffic of K-HEADS.
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log | cut -c1-200
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench c3 rc=$?"
timeout 900 python bench.py --workload c4 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 rc=$?"
timeout 900 python bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench c5 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- python bench.py --steps 48 --warmup 5 --no-cpu-baseline --graph off > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name '*kernel_trace*' -delete; find $OUT/prof -name '*domain_stats*' -delete
bash tools/gpu_pmc_bench.sh $TAG/sq > $OUT/sq.log 2>&1; tail -12 $OUT/sq.log | cut -c1-220
bash tools/gpu_pmc_traffic.sh $TAG/traffic > $OUT/traffic.log 2>&1; tail -3 $OUT/traffic.log | cut -c1-400
find $OUT -name '*kernel_trace.csv' -delete
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for f in ('bench','bench_c4','bench_c5'):
    try:
        d=json.loads([l for l in open('%s/%s.json'%(o,f)).read().splitlines() if l.startswith('{')][0])
        r=d['roofline']
        print('%-9s %.0f cells/s  %.4f ms/step  | %s %.4f of %s (%s) | batch32 %s | cpu %s' % (f, d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['peak'], r['unit'], (d['config'].get('batch32') or {}).get('ms_per_step'), (d.get('cpu_baseline') or {}).get('value')))
        for k in d['kernels'][:7]: print('      %-16s %.4f ms  frac %.3f'%(k['kernel'],k['mean_ms'],k.get('frac',0)))
    except Exception as e: print(f,'FAILED',e)
PY
