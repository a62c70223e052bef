#!/bin/bash
# K8 iteration loop: parity tests, split timing, bench knn section
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_hip_knn.py -x -q -m gpu > gpurun_out/r03/knn_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r03/knn_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/debug/knn_split.py 2>&1 | tee gpurun_out/r03/knn_split.log &&
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --sections knn > gpurun_out/r03/knn_bench.json 2> gpurun_out/r03/knn_bench.err &&
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/knn_bench.json').read().strip().splitlines()[-1])
k = d.get('knn', d)
for name in ('shell_points', 'rendered_view_geometry'):
    v = k[name]
    print(name, {x: v[x] for x in ('ms_per_view', 'grid_build_ms', 'candidates_examined_per_query', 'far_search_queries')})
PY
