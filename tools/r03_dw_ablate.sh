#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
: > $O/dw_ablate.log
for v in "" _exp_dw_nodma _exp_dw_nobarrier _exp_dw_nofetch _exp_dw_norowsum "_exp_dw_nodma+dw_nobarrier"; do
  echo "== lib$v" >> $O/dw_ablate.log
  timeout -k 10 120 python3 tools/microbench_mlp.py --lib "nerfail_amd/lib/libnerfail_hip$v.so" --only bwd_weights --sizes 2048x128 >> $O/dw_ablate.log 2>&1 || exit 1
done
cat $O/dw_ablate.log | grep -v amdgpu.ids
