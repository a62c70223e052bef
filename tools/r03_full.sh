#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
echo skip pytest
echo skip smoke
T0=$(date +%s); timeout -k 10 900 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; tail -c 600 $O/bench_full.json
