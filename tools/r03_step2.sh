#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_hip_train.py tests/test_hip_nerf.py tests/test_hip_f16x3.py tests/test_hip_ops.py -x -q -m gpu > $O/step2_tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/step2_tests.log
timeout -k 10 200 python3 tools/microbench_mlp.py --only fwd_infer,fwd_train --sizes 1024x64,1024x192,2048x128 > $O/microbench_fwd.log 2>&1; grep -v amdgpu.ids $O/microbench_fwd.log
NERFAIL_FWD_KERNEL=reg timeout -k 10 200 python3 tools/microbench_mlp.py --only fwd_train --sizes 1024x64,1024x192,2048x128 > $O/microbench_fwd_reg.log 2>&1; grep -v amdgpu.ids $O/microbench_fwd_reg.log
bash tools/r03_dw_ablate.sh
