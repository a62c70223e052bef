#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_train.py tests/test_hip_nerf.py tests/test_hip_f16x3.py -x -q -m gpu > $O/step4_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/step4_tests.log
timeout -k 10 200 python3 tools/microbench_mlp.py --only fwd_infer,fwd_train --sizes 1024x64,1024x192,2048x128 > $O/microbench_fwd2.log 2>&1; grep -v amdgpu.ids $O/microbench_fwd2.log
timeout -k 10 300 python3 bench.py --sections train --no-cpu-baseline > $O/bench_train2.json 2> $O/bench_train2.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_train2.json'))['train']
print({k:d[k] for k in ('ms_per_step','train_rays_per_sec_fwd_bwd','ms_per_step_mean_whole_loop','final_loss')}, d['roofline']['frac'])
print(d['ms_per_step_each'])
PY
