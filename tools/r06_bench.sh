#!/bin/bash
# Round 6: the driver's command, then the self-launched 2-rank rehearsal (gloo, both ranks on GPU 0).
O=gpurun_out/r06; mkdir -p $O
export NERFAIL_BENCH_DETAIL=$O/bench_detail_driver_cmd.json
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.json 2> $O/driver_cmd.err; echo "driver cmd rc $?" | tee $O/driver_cmd.rc
wc -c $O/driver_cmd.json; cat $O/driver_cmd.json
export NERFAIL_BENCH_DETAIL=$O/bench_detail_2rank_gloo.json
timeout -k 10 500 python3 bench.py --gpus 2 --dist-backend gloo --device 0 --steps 2 --warmup 1 > $O/selfspawn_2rank_gloo.json 2> $O/selfspawn_2rank_gloo.err; echo "2-rank rc $?" | tee $O/selfspawn_2rank_gloo.rc
cat $O/selfspawn_2rank_gloo.json
