#!/bin/bash
# first-layer forward on the matrix pipe: parity test, microbenchmark against the dense forms, bench line, kernel durations
OUT=gpurun_out/${1:-r04l}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_sparse_gpu.py -x -q -m gpu -k first_layer < /dev/null 2>&1 | tail -5 | tee $OUT/test_sparse.log
timeout 300 python tools/bench_enc0.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_enc0.txt
timeout 300 python bench.py --steps 96 --warmup 16 --no-cpu-baseline < /dev/null 2>/dev/null | grep '^{' > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
print('ms_per_step %.4f cells/s %.0f' % (j['ms_per_step'], j['value']), {k['kernel']: round(k['mean_ms'], 4) for k in j['kernels']})
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- python bench.py --steps 48 --warmup 5 --no-cpu-baseline --graph off > $OUT/bench_prof.json 2> $OUT/prof.err < /dev/null; echo "prof rc=$?"
find $OUT/prof -name '*kernel_trace*' -delete; find $OUT/prof -name '*domain_stats*' -delete
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then grep "enc0\|heads_fused_x3" "$f" | cut -c1-200; fi
