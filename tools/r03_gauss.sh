#!/bin/bash
# gauss path loop: parity tests + the attack section of the bench (gauss path, per-kernel timings)
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_hip_gauss.py tests/test_hip_cfg.py -x -q -m gpu > gpurun_out/r03/gauss_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r03/gauss_tests.log
[ $rc -eq 0 ] || exit $rc
NERFAIL_BENCH_LIGHT=1 NERFAIL_BENCH_TUNE_VICTIM=0 timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --sections attack > gpurun_out/r03/gauss_bench.json 2> gpurun_out/r03/gauss_bench.err || exit 1
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/gauss_bench.json').read().strip().splitlines()[-1])
a = d.get('attack', d)
print('gauss path', a['gauss_path_deterministic']['ms_per_iter'], a['gauss_path_deterministic']['ms_per_iter_each_block'])
for k, v in a['gauss_kernels'].items():
    if isinstance(v, dict): print(k, round(v['ms_per_call'], 4))
PY
