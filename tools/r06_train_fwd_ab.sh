#!/bin/bash
# Round 6: training-forward step-form A/B (tools/experiment.py lds_train_* builds): kernel time at a LONG launch (24 rounds: the
# clock has ramped, see profiles/r06_launch_timeline_clocks.txt) + the bitwise test of every candidate that is meant to be kept.
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do
for v in "" $VARIANTS; do
  L=""; [ -n "$v" ] && L="--lib nerfail_amd/lib/libnerfail_hip_exp_$v.so"
  echo "== ${v:-product}"
  timeout -k 10 200 python3 tools/microbench_mlp.py $L --only fwd_train,fwd_infer --sizes ${SIZES:-4096x192} 2>&1 | grep "M="
done
done | tee $O/train_fwd_ab.log
for v in $CHECK; do
  echo "== bits $v" | tee -a $O/train_fwd_ab.log
  NERFAIL_HIP_LIB=$PWD/nerfail_amd/lib/libnerfail_hip_exp_$v.so timeout -k 10 300 python3 -m pytest tests/test_hip_nerf.py tests/test_hip_train.py -m gpu -q -x -p no:cacheprovider -k "lds_training_forward or points_formed or training_step_gradients" 2>&1 | tail -2 | tee -a $O/train_fwd_ab.log
done
