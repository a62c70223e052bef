#!/bin/bash
# Round 5: A/B of mlp_lds.hip step-schedule variants (tools/experiment.py builds) - kernel time at render size + per-step stamps.
O=gpurun_out/r05/mlp_ab; mkdir -p $O
for v in "" $VARIANTS; do
  L=""; [ -n "$v" ] && L="--lib nerfail_amd/lib/libnerfail_hip_exp_$v.so"
  echo "== ${v:-product}" | tee -a $O/ab.log
  timeout -k 10 200 python3 tools/microbench_mlp.py $L --only ${ONLY:-fwd_infer} --sizes ${SIZES:-65536x192} 2>&1 | grep "M=" | tee -a $O/ab.log
done
for v in $STEPS; do
  echo "== steps $v" | tee -a $O/ab.log
  timeout -k 10 200 python3 tools/lds_steps.py ${v%%:*} $( [ "${v#*:}" != "$v" ] && echo ${v#*:} ) 2>&1 | tail -3 | tee -a $O/ab.log
done
