#!/bin/bash
# Round-6 judged artefacts from the FINAL binary, in this order: (1) pytest -m gpu, (2) rocprofv3 kernel stats of every bench
# group, (3) the HBM counter passes (FETCH_SIZE, WRITE_SIZE: their own runs, kernel trace only), (4) the SQ / GRBM pass of the
# render group (matrix-pipe utilisation). Every group is profiled as `bench.py --child <group>` (one process: a
# profiler-preloaded process must not start children). Then copy the summaries to profiles/ and run tools/r06_bench.sh.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06/final
rm -rf $O && mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; rc=$?
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
bash tools/r06_pmc_sq.sh render || exit 1
cp gpurun_out/r06/pmc_sq/pmc_sq_render.json $O/pmc_sq_render.json
find gpurun_out/r06/pmc_sq -name "*counter_collection.csv" -delete; find gpurun_out/r06/pmc_sq -name "*kernel_trace.csv" -delete
echo "profiles done"
timeout -k 10 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
