#!/bin/bash
# Round 4: where K10 / K11 spend their time: kernel stats + counter passes of the attack group's light form (kernels only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04/${TAG:-k11}
mkdir -p $O
export NERFAIL_BENCH_LIGHT=1
CMD="bench.py --child attack"
timeout -k 10 300 python3 $CMD > $O/plain.jsonl 2> $O/plain.err || exit 1
python3 - <<PY
import json,sys
sys.path.insert(0,'.')
import bench
r = bench._read_results('$O/plain.jsonl')['attack']
print('gauss path ms', r['gauss_path_deterministic']['ms_per_iter'], r['gauss_path_deterministic']['ms_per_iter_each_block'])
for k, v in r['gauss_kernels'].items():
    if isinstance(v, dict): print(k, round(v['ms_per_call'], 4), v['roofline']['frac'])
PY
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o a -- python3 $CMD > /dev/null 2> $O/stats.log || exit 1
python3 - <<PY
import csv,glob
f = glob.glob('$O/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gauss' in r['Name'] or 'igsm' in r['Name'] or 'seg' in r['Name']:
        print('%-70s calls %-5s avg_us %.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -f $O/stats/*.db $O/stats/*/*.db $O/stats/*/*kernel_trace.csv
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf $O/pmc$i
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -o p -- python3 $CMD > /dev/null 2> $O/pmc$i.log || { tail -5 $O/pmc$i.log; exit 1; }
  rm -f $O/pmc$i/*.db $O/pmc$i/*/*.db $O/pmc$i/*/*kernel_trace.csv
done
python3 tools/pmc_table.py $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 --grep=gauss > $O/pmc_table.txt
grep -A12 "seg_reduce_views_kernel<false>\|pixel_grad_rgb\|rows_sum3\|fwd_views_kernel<true>" $O/pmc_table.txt | head -120
