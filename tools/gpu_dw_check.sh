#!/bin/bash
# first-layer weight gradient from the byte store: parity test + timing alone + the bench line
OUT=gpurun_out/${1:-dwc}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k first_layer < /dev/null 2>&1 | tail -3 | tee $OUT/test_sparse.log
for r in 1 2 3; do timeout 200 python tools/bench_enc0_dw.py 4096 30 < /dev/null 2>/dev/null | grep "dW from" | tee -a $OUT/dw.txt; done
timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline < /dev/null 2>/dev/null | grep '^{' > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
print('ms_per_step %.4f cells/s %.0f' % (j['ms_per_step'], j['value']), {k['kernel']: round(k['mean_ms'], 4) for k in j['kernels']})
PY
