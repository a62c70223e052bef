#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_train.py tests/test_hip_nerf.py tests/test_hip_f16x3.py tests/test_hip_ops.py tests/test_hip_cfg.py tests/test_hip_pipeline.py tests/test_hip_adam.py -x -q -m gpu > $O/step3_tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/step3_tests.log
timeout -k 10 300 python3 bench.py --sections train --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_train.json'))['train']
print({k:d[k] for k in ('ms_per_step','train_rays_per_sec_fwd_bwd','ms_per_step_mean_whole_loop','final_loss')}, d['roofline']['frac'])
print(d['ms_per_step_each'])
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_new -o tl -- python3 tools/profile_train_step.py f32 > $O/tl_new.log 2>&1
python3 tools/train_timeline.py $O/tl_new 10 > $O/tl_new_summary.txt 2>&1
rm -f $O/tl_new/*/*.db
grep -v "^E2026\|^W2026" $O/tl_new.log | tail -3; head -40 $O/tl_new_summary.txt
