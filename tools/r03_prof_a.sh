#!/bin/bash
# round-3 judged artefacts, part A: default bench line + rocprofv3 kernel stats (part B = the two counter passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O && rm -rf $O/stats
timeout -k 10 420 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err || exit 1
echo "bench line done" && tail -c 200 $O/bench_line.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/rocprof_stats.log || exit 1
echo "stats pass done"
rm -f $O/stats/*.db $O/stats/*/*.db
ls -la $O/stats $O/stats/* | head -20
