#!/bin/bash
# kernel stats + FETCH_SIZE pass of the gauss kernels (bench.py --sections attack); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NERFAIL_BENCH_TUNE_VICTIM=0
rm -rf gpurun_out/prof_attack gpurun_out/pmc_attack_fetch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_attack -o attack -- python3 bench.py --sections attack > gpurun_out/prof_attack.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_attack_fetch -o f -- python3 bench.py --sections attack > gpurun_out/pmc_attack_fetch.log 2>&1 || exit 1
