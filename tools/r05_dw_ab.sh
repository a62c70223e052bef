#!/bin/bash
# Round 5: weight-gradient kernel A/B (product = LDS-DMA in the global form; experiment dw_dma_buf = the buffer form; fixed in round 6:
# the script named a variant that tools/experiment.py does not have) + the training tests + the train section.
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_hip_train.py tests/test_hip_f16x3.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for k in "" dw_dma_buf "" dw_dma_buf; do
  echo "== ${k:-product}"
  L=""; [ -n "$k" ] && L="--lib nerfail_amd/lib/libnerfail_hip_exp_$k.so"
  timeout -k 10 120 python3 tools/microbench_mlp.py $L --dual 1024x64+192 2>&1 | grep "bwd_weights2"
done | tee $O/dw_ab.log
timeout -k 10 200 python3 bench.py --child train 2>/dev/null > $O/train_child.jsonl
python3 - <<P
import json
for ln in open('$O/train_child.jsonl'):
    try: d = json.loads(ln)
    except ValueError: continue
    for k, v in d.items():
        if isinstance(v, dict) and "ms_per_step" in v: print(k, v["ms_per_step"], v.get("roofline", {}).get("frac"))
P
