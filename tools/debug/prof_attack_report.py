import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open('gpurun_out/prof_attack/attack_kernel_stats.csv')))
t = {r['Name'].split('(')[0]: (int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3) for r in rows if 'nerfail' in r['Name']}
f = defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_attack_fetch/f_counter_collection.csv')):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'nerfail' in r['Kernel_Name']:
        f[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']) * 2048 / 1e6)
for k, (c, a, m) in sorted(t.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if 'gauss' in k or 'igsm' in k:
        print(k[-45:].ljust(45), str(c).rjust(4), '%8.1f us avg %8.1f min   fetch %7.1f MB max %7.1f' % (a, m, sum(f[k]) / max(1, len(f[k])), max(f[k] or [0])))
