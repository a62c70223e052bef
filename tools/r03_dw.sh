#!/bin/bash
# round-3: new weight-gradient kernel - parity tests, timings, balance fit
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_hip_train.py tests/test_hip_f16x3.py tests/test_hip_ops.py -x -q -m gpu > $O/dw_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/dw_tests.log
timeout -k 10 300 python3 tools/microbench_mlp.py --only bwd_data,bwd_weights,bwd_w_bf16x3 --sizes 1024x64,1024x192,2048x128 > $O/microbench_dw.log 2>&1 || exit 1
timeout -k 10 300 python3 tools/microbench_mlp.py --dual 1024x64+192 >> $O/microbench_dw.log 2>&1 || exit 1
cat $O/microbench_dw.log
timeout -k 10 300 python3 tools/dw_balance.py > $O/dw_balance.log 2>&1; tail -8 $O/dw_balance.log
