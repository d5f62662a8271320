#!/bin/bash
OUT=gpurun_out/r04j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_sparse_gpu.py tests/test_api_gpu.py -x -q --tb=short > $OUT/pytest_dp.log 2>&1; echo "dp+sparse+api rc=$?"; tail -4 $OUT/pytest_dp.log | cut -c1-300
for mode in single dp_graph dp_eager dp_graph_stackoff; do
  case $mode in
    single) env="";;
    dp_graph) env="DCA_AMD_DIST_FORCE=1";;
    dp_eager) env="DCA_AMD_DIST_FORCE=1 DCA_AMD_DP_GRAPH=0";;
    dp_graph_stackoff) env="DCA_AMD_DIST_FORCE=1 DCA_AMD_STACK=off";;
  esac
  env $env timeout 300 python bench.py --steps 48 --warmup 5 --no-cpu-baseline > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  python - "$mode" <<'PY'
import json,sys
m=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/r04j/bench_%s.json'%m).read())
    print('%-20s ms/step %.4f  cells/s %.0f  launch: %s'%(m, d['ms_per_step'], d['value'], d['config']['launch'][:60]), '| exposed comm', json.dumps(d['config'].get('exposed_comm_ms_per_step'))[:400])
except Exception as e:
    print(m,'FAILED',e, open('gpurun_out/r04j/bench_%s.err'%m).read()[-600:])
PY
done
