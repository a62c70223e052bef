#!/bin/bash
# part B: two counter passes (FETCH_SIZE, WRITE_SIZE; their own runs, kernel trace only) -> HBM bytes per launch and kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O && rm -rf $O/pmc_fetch $O/pmc_write
export NERFAIL_BENCH_TUNE_VICTIM=0
export NERFAIL_BENCH_LIGHT=1
PMC_CMD="bench.py --steps 1 --warmup 0 --no-cpu-baseline --sections render,train,attack"
timeout -k 10 520 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $PMC_CMD > $O/pmc_fetch.json 2> $O/pmc_fetch.log || exit 1
echo "fetch pass done"
timeout -k 10 520 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $PMC_CMD > $O/pmc_write.json 2> $O/pmc_write.log || exit 1
echo "write pass done"
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 $PMC_CMD (NERFAIL_BENCH_LIGHT=1)" > $O/pmc_hbm_traffic.json || exit 1
rm -f $O/pmc_fetch/*_kernel_trace.csv $O/pmc_write/*_kernel_trace.csv $O/pmc_fetch/*/*_kernel_trace.csv $O/pmc_write/*/*_kernel_trace.csv $O/pmc_*/*.db $O/pmc_*/*/*.db
head -c 1500 $O/pmc_hbm_traffic.json
