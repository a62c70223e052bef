#!/bin/bash
# Round 6: every kernel of one free-running training step (rocprofv3 --kernel-trace of tools/profile_train_step.py f32).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o tl -- python3 tools/profile_train_step.py f32 > $O/train_profile${TAG}.log 2>&1
python3 tools/train_timeline.py $O/tl 10 > $O/train_timeline${TAG}.txt 2>&1
rm -rf $O/tl
cat $O/train_profile${TAG}.log | tail -3; cat $O/train_timeline${TAG}.txt
