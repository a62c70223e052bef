#!/bin/bash
# Round 6: the N > 1 forms of bench.py on the 1-GPU box: (a) under torch.distributed.run, two gloo ranks on GPU 0 (what the driver's
# launcher does, minus RCCL); (b) the 1-rank RCCL dry run (communicator + the attack leg's all-reduce really issued).
O=gpurun_out/r06; mkdir -p $O
NERFAIL_BENCH_DETAIL=$O/bench_detail_torchrun_2rank_gloo.json timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --dist-backend gloo --device 0 --steps 2 --warmup 1 > $O/torchrun_2rank_gloo.json 2> $O/torchrun_2rank_gloo.err; echo "torchrun 2-rank rc $?"
tail -c 1500 $O/torchrun_2rank_gloo.json; echo
NERFAIL_BENCH_DRYRUN_NCCL=1 NERFAIL_BENCH_DETAIL=$O/bench_detail_nccl_1rank_dryrun.json timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --sections render > $O/nccl_1rank_dryrun.json 2> $O/nccl_1rank_dryrun.err; echo "nccl dry run rc $?"
python3 - <<P
import json
d = json.load(open('$O/bench_detail_nccl_1rank_dryrun.json'))
a = d.get('attack_nccl_dryrun') or d.get('attack_nccl_dryrun_error')
print('attack_nccl_dryrun:', {k: a[k] for k in ('iters_per_sec', 'allreduce_ms', 'allreduce_bytes', 'allreduce_backend', 'perturbation_identical_on_all_ranks')} if isinstance(a, dict) else a)
P
