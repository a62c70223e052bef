#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
: > $O/fwd_ablate.log
for v in "" _exp_lds_train_nont _exp_lds_train_nostore _exp_lds_train_nomask; do
  echo "== lib$v" >> $O/fwd_ablate.log
  timeout -k 10 120 python3 tools/microbench_mlp.py --lib "nerfail_amd/lib/libnerfail_hip$v.so" --only fwd_infer,fwd_train --sizes 2048x128 >> $O/fwd_ablate.log 2>&1 || exit 1
done
grep -v amdgpu.ids $O/fwd_ablate.log
