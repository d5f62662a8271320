#!/bin/bash
OUT=gpurun_out/r04h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/b32_probe.py > $OUT/b32_probe.txt 2>&1; tail -9 $OUT/b32_probe.txt | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/b32prof -o b32 --output-format csv -- python tools/b32_probe.py --rounds 3 --spin 0.2 > $OUT/b32_prof.log 2>&1; echo "prof rc=$?"
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r04h/b32prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows=[r for r in rows if int(r['Calls'])>=1000]
    for r in rows[:12]: print('%-70s calls %6s avg %8.0f ns  min %7s max %8s'%(r['Name'][:70].replace('void (anonymous namespace)::',''), r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench.json').read())
print('value %.0f ms/step %.4f'%(d['value'], d['ms_per_step']), 'batch32', d['config'].get('batch32'))
print('roofline', json.dumps(d['roofline'])[:700])
for k in d['kernels'][:6]: print('   ', k['kernel'], '%.4f ms'%k['mean_ms'], 'frac %.3f'%k.get('frac',0))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
