#!/bin/bash
# round-3 baseline: (1) a*rounds + b fit inputs for the four exact-f32 MLP kernels, (2) kernel timeline of the training step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python3 tools/microbench_mlp.py --only fwd_infer,fwd_train,bwd_data,bwd_weights --sizes 512x64,1024x64,1024x128,1024x192,2048x128,4096x128 > $O/microbench_base.log 2>&1 || exit 1
echo microbench done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_base -o tl -- python3 tools/profile_train_step.py f32 > $O/tl_base.log 2>&1 || exit 1
python3 tools/train_timeline.py $O/tl_base 10 > $O/tl_base_summary.txt 2>&1
rm -f $O/tl_base/*/*.db
tail -5 $O/tl_base.log; head -30 $O/tl_base_summary.txt
