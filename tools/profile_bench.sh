#!/bin/bash
# Everything profiles/r03_* is made from, in one GPU-box call:  gpurun --timeout 1200 -- 'bash tools/profile_bench.sh'
#   1. the default bench line                                  -> gpurun_out/r03/bench_line.json
#   2. rocprofv3 --kernel-trace --stats of the same workload   -> gpurun_out/r03/stats/bench_kernel_stats.csv
#   3. two counter passes (FETCH_SIZE, WRITE_SIZE; their own runs, no trace domains besides --kernel-trace) reduced by
#      tools/pmc_summary.py to HBM bytes per launch and kernel -> gpurun_out/r03/pmc_hbm_traffic.json
# The program stands directly behind `--` (no env / bash -c hop: the profiler has initialised the GPU by then).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O && rm -rf $O/stats $O/pmc_fetch $O/pmc_write
export NERFAIL_BENCH_TUNE_VICTIM=${NERFAIL_BENCH_TUNE_VICTIM:-1}
timeout -k 10 420 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err || exit 1
echo "bench line done" && tail -c 300 $O/bench_line.json
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/rocprof_stats.log || exit 1
echo "stats pass done"
export NERFAIL_BENCH_TUNE_VICTIM=0
export NERFAIL_BENCH_LIGHT=1
PMC_CMD="bench.py --steps 1 --warmup 0 --no-cpu-baseline --sections render,train,attack"
timeout -k 10 560 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $PMC_CMD > $O/pmc_fetch.json 2> $O/pmc_fetch.log || exit 1
echo "fetch pass done"
timeout -k 10 560 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $PMC_CMD > $O/pmc_write.json 2> $O/pmc_write.log || exit 1
echo "write pass done"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 $PMC_CMD" > $O/pmc_hbm_traffic.json || exit 1
rm -f $O/stats/*.db $O/pmc_fetch/*_kernel_trace.csv $O/pmc_write/*_kernel_trace.csv
ls -la $O $O/stats
