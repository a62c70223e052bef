#!/usr/bin/env python3
"""profiles/r04_driver_cmd_runs.txt from gpurun_out/r04/driver_cmd_*.json (the driver's command, N runs in a row on one lease)."""
import glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r04'
status = {l.split()[1]: l.strip() for l in open(os.path.join(d, 'driver_cmd.status')) if l.startswith('run ')}
print('python3 bench.py --gpus 1 --steps 20 --warmup 5, consecutive runs on one MI355X lease (tools/r04_driver_cmd.sh)')
for f in sorted(glob.glob(os.path.join(d, 'driver_cmd_*.json')), key=lambda p: int(p.split('_')[-1].split('.')[0])):
    k = f.split('_')[-1].split('.')[0]
    x = json.loads(open(f).read().strip().splitlines()[-1])
    errs = [e for e in x if e.endswith('_error')]
    print('%s | value %.0f rays/s, %.1f ms/step, roofline.frac %.4f (traffic %s), composite %.3f, cpu_baseline %.0f rays/s on %d cores | '
          'train %.3f ms (frac %.3f) | gauss path %.4f ms (frac %.3f) | knn %.2f ms | wall %.1f s, sections %s | errors %s'
          % (status.get(k, 'run ' + k), x['value'], x['ms_per_step'], x['roofline']['frac'], x['roofline']['traffic'], x['composite_scan']['frac'],
             x['cpu_baseline']['value'], x['cpu_baseline']['cores'], x['train']['ms_per_step'], x['roofline_fwd_bwd']['frac'],
             x['attack']['gauss_path_deterministic']['ms_per_iter'], x['roofline_attack']['frac'], x['knn']['rendered_view_geometry']['ms_per_view'],
             x['wall_seconds'], {g: (s['rc'], s['seconds']) for g, s in x['sections'].items()}, errs))
