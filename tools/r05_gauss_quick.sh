#!/bin/bash
# quick K10/K11/K12 timing: the attack group's light form, plain + rocprofv3 kernel stats only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${ROUND:-r05}/${TAG:-k11q}
mkdir -p $O
export NERFAIL_BENCH_LIGHT=1
timeout -k 10 300 python3 bench.py --child attack > $O/plain.jsonl 2> $O/plain.err || { tail -5 $O/plain.err; exit 1; }
python3 - <<PY
import sys
sys.path.insert(0,'.')
import bench
r = bench._read_results('$O/plain.jsonl')['attack']
print('gauss path ms', r['gauss_path_deterministic']['ms_per_iter'])
for k, v in r['gauss_kernels'].items():
    if isinstance(v, dict): print(k, round(v['ms_per_call'], 4), round(v['roofline']['frac'],3))
PY
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o a -- python3 bench.py --child attack > /dev/null 2> $O/stats.log || exit 1
python3 - <<PY
import csv,glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if ('gauss' in r['Name'] or 'igsm' in r['Name']) and 'bwd_kernel' not in r['Name'] and 'weight' not in r['Name']:
        print('%-66s calls %-5s avg_us %.1f' % (r['Name'][:66], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -f $O/stats/*.db $O/stats/*/*.db $O/stats/*/*kernel_trace.csv $O/stats/*kernel_trace.csv
