#!/bin/bash
# Round-4 judged artefacts from the FINAL binary, in this order: (1) pytest -m gpu, (2) rocprofv3 kernel stats of every bench
# group, (3) the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: their own runs, kernel trace only), (4) the bench line.
# The bench parent starts children, which a profiler-preloaded process must not do: every group is profiled as
# `bench.py --child <group>` (one process).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/final
rm -rf $O && mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
  tail -n 3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
fi
for g in render train attack extras; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $g -- python3 bench.py --child $g --steps 2 --warmup 1 > $O/stats_$g.jsonl 2> $O/stats_$g.log || { tail -5 $O/stats_$g.log; exit 1; }
done
echo "stats passes done"
export NERFAIL_BENCH_LIGHT=1
for c in FETCH_SIZE WRITE_SIZE; do
  for g in render train attack; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o $g -- python3 bench.py --child $g --steps 1 --warmup 0 > /dev/null 2> $O/pmc_${c}_$g.log || { tail -5 $O/pmc_${c}_$g.log; exit 1; }
  done
done
unset NERFAIL_BENCH_LIGHT
echo "counter passes done"
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python3 bench.py --child {render|train|attack} --steps 1 --warmup 0 (NERFAIL_BENCH_LIGHT=1)" > $O/pmc_hbm_traffic.json || exit 1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*counter_collection.csv" -delete
cp $O/pmc_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json
timeout -k 10 420 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
tail -c 300 $O/bench_default.json
