#!/bin/bash
# Round-6 first visit: parity suite on the cleaned library, bench line, per-phase cycles of K-HEADS (timing build).
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log | cut -c1-300
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench c3 rc=$?"
COMPACT=1 ONLY_FUSED=1 timeout 300 python tools/timing_heads.py 4096 20000 64 1 > $OUT/heads_timing.txt 2>&1; echo "timing rc=$?"; tail -16 $OUT/heads_timing.txt
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
d=json.loads([l for l in open('%s/bench.json'%o).read().splitlines() if l.startswith('{')][0])
r=d['roofline']
print('%.0f cells/s  %.4f ms/step | %s %.4f | batch32 %s | step %s' % (d['value'], d['ms_per_step'], r['kernel'], r['frac'], (d.get('batch32') or {}).get('ms_per_step'), {k:v for k,v in r['step'].items() if k in ('frac','frac_of_fp32_ridge')}))
for k in d['kernels'][:8]: print('      %-16s %.4f ms  frac %.3f'%(k['kernel'],k['mean_ms'],k.get('frac',0)))
PY
