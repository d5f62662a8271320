#!/bin/bash
# Round-6 visit: the fp16 x 2 plane family + K-HEADS tests first, then the whole parity suite, c3 and c5 bench lines.
TAG=${1:-r06d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_h2_gpu.py tests/test_heads_fused_gpu.py -q -x 2>&1 | tail -12
timeout 900 python -m pytest tests/test_engine_gpu.py -q -x -k "wide_network" 2>&1 | tail -12
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench c3 rc=$?"
timeout 900 python bench.py --workload c5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "bench c5 rc=$?"
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for f in ('bench','bench_c5'):
    try:
        d=json.loads([l for l in open('%s/%s.json'%(o,f)).read().splitlines() if l.startswith('{')][0])
        r=d['roofline']
        print('%-9s %.0f cells/s  %.4f ms/step  | %s %.4f | loss %.5f -> %.5f' % (f, d['value'], d['ms_per_step'], r['kernel'], r['frac'], d['loss_first'], d['loss_last']))
        for k in d['kernels'][:8]: print('      %-16s %.4f ms  frac %.3f'%(k['kernel'],k['mean_ms'],k.get('frac',0)))
    except Exception as e: print(f,'FAILED',e); print(open('%s/%s.err'%(o,f)).read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log | cut -c1-300
