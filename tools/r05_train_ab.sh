#!/bin/bash
# Round 5: backward-data A/B (register-streamed vs LDS ring, and ring variants) at the training step's sizes, then the train
# section of bench.py.
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 240 python3 -m pytest tests/test_hip_train.py -m gpu -q -x -p no:cacheprovider -k "lds_ring_backward or reproducible or joint" 2>&1 | tail -2
for k in reg lds $VARIANTS reg lds $VARIANTS; do
  echo "== $k"
  if [ "$k" = reg ] || [ "$k" = lds ]; then NERFAIL_BWD_KERNEL=$k timeout -k 10 120 python3 tools/microbench_mlp.py --dual 1024x64+192 2>&1 | grep "bwd_data2"
  else timeout -k 10 120 python3 tools/microbench_mlp.py --lib nerfail_amd/lib/libnerfail_hip_exp_$k.so --dual 1024x64+192 2>&1 | grep "bwd_data2"; fi
done | tee $O/bwd_ab.log
timeout -k 10 200 python3 bench.py --child train 2>/dev/null > $O/train_child.jsonl
python3 - <<P
import json
for ln in open('$O/train_child.jsonl'):
    try: d = json.loads(ln)
    except ValueError: continue
    for k, v in d.items():
        if isinstance(v, dict) and "ms_per_step" in v: print(k, v["ms_per_step"], v.get("roofline", {}).get("frac"))
P
