#!/usr/bin/env python3
"""Reads a `rocprofv3 --kernel-trace --output-format csv` trace of tools/profile_train_step.py and prints, for the
free-running training steps at the end of that script, every kernel of one median step with its duration and the idle
gap in front of it, plus totals: time inside the big MLP kernels, inside all other kernels, and idle between kernels.
Usage: python tools/train_timeline.py <dir with *_kernel_trace.csv> [n_steps_to_average]"""
import csv
import glob
import os
import sys

BIG = ('nerf_mlp_fwd', 'nerf_mlp_bwd_data', 'nerf_mlp_bwd_weights', 'nerf_mlp_dw_lds')


def main():
    d = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        raise SystemExit('no *kernel_trace.csv under ' + d)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # a step ends with the Adam kernel; take the last `nsteps` complete steps
    ends = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r[2]]
    if len(ends) < nsteps + 1:
        raise SystemExit('only %d adam steps in the trace' % len(ends))
    ends = ends[-(nsteps + 1):]
    tot = {'big': 0.0, 'small': 0.0, 'gap': 0.0, 'wall': 0.0, 'n_small': 0}
    per_kernel = {}
    for a, b in zip(ends[:-1], ends[1:]):
        seg = rows[a + 1:b + 1]
        prev_end = rows[a][1]
        tot['wall'] += (seg[-1][1] - rows[a][1]) / 1e3
        for s, e, name in seg:
            dur, gap = (e - s) / 1e3, max(0.0, (s - prev_end) / 1e3)
            prev_end = max(prev_end, e)
            short = name.split('(')[0].replace('void ', '')[:70]
            big = any(k in name for k in BIG)
            tot['big' if big else 'small'] += dur
            tot['gap'] += gap
            tot['n_small'] += 0 if big else 1
            k = per_kernel.setdefault(short, [0, 0.0, 0.0])
            k[0] += 1
            k[1] += dur
            k[2] += gap
    n = float(nsteps)
    print('per step (mean of %d): wall %.1f us, big MLP kernels %.1f us, %d other kernels %.1f us, idle gaps %.1f us'
          % (nsteps, tot['wall'] / n, tot['big'] / n, tot['n_small'] / n, tot['small'] / n, tot['gap'] / n))
    print('%-72s %6s %10s %10s' % ('kernel', 'calls', 'us/step', 'gap us/step'))
    for k, (c, dur, gap) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        print('%-72s %6.1f %10.1f %10.1f' % (k, c / n, dur / n, gap / n))
    # one step, in order
    a, b = ends[-2], ends[-1]
    prev_end = rows[a][1]
    print('--- last step in launch order (us: gap before, duration)')
    for s, e, name in rows[a + 1:b + 1]:
        print('%8.1f %9.1f  %s' % (max(0.0, (s - prev_end) / 1e3), (e - s) / 1e3, name.split('(')[0].replace('void ', '')[:90]))
        prev_end = max(prev_end, e)


if __name__ == '__main__':
    main()
