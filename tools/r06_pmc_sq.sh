#!/bin/bash
# Round 6 (as round 5): matrix-pipe utilisation of the headline kernel FROM COUNTERS. One rocprofv3 --pmc pass (SQ + GRBM
# slots only: 7 SQ of 8, 1 GRBM of 2; kernel trace only beside it) of `bench.py --child render` (one process, no children).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06/pmc_sq; rm -rf $O; mkdir -p $O
G=${1:-render}
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"
NERFAIL_BENCH_LIGHT=1 timeout -k 10 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o $G -- python3 bench.py --child $G --steps 2 --warmup 1 > $O/$G.jsonl 2> $O/$G.log || { tail -20 $O/$G.log; exit 1; }
python3 tools/pmc_sq_summary.py $O "rocprofv3 --pmc $C --kernel-trace -- python3 bench.py --child $G --steps 2 --warmup 1" nerf_mlp composite > $O/pmc_sq_$G.json || exit 1
find $O -name "*.db" -delete
python3 - <<P
import json
d = json.load(open('$O/pmc_sq_$G.json'))
for k, v in d['kernels'].items():
    print(k[:70], {x: (round(y, 4) if isinstance(y, float) else y) for x, y in v.items() if x != 'counters'})
P
