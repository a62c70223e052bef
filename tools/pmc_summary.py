#!/usr/bin/env python3
"""Reduce two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of the SAME command) to
HBM(+Infinity Cache) bytes per launch for every nerfail kernel, with the gfx950 corrections of MI355X_MICROARCH.md
(FETCH_SIZE counts 128-byte requests at 64 B: x2; both counters are in KB).

    python tools/pmc_summary.py <fetch_dir> <write_dir> "<command that was profiled>" > profiles/<name>.json
"""
import csv, glob, json, os, sys
from collections import defaultdict


def read(d, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter and 'nerfail' in r['Kernel_Name']:
                per[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return per


def main():
    fetch, write = read(sys.argv[1], 'FETCH_SIZE'), read(sys.argv[2], 'WRITE_SIZE')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    try:                                       # ties the measurement to the kernel sources it was taken from:
        import bench                           # bench.py reports a kernel's traffic only while the files it is compiled from still hash the same
        sha, files = bench.kernel_source_hash(), bench.kernel_source_hashes()
    except Exception:
        sha, files = None, {}
    out = {'command': sys.argv[3] if len(sys.argv) > 3 else '', 'csrc_sha16': sha, 'csrc_files': files,
           'correction': 'FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), KB -> bytes x1024; WRITE_SIZE KB -> bytes',
           'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        out['kernels'][k] = {'launches': max(len(f), len(w)),
                             'fetch_bytes_per_launch_corrected': (sum(f) / len(f) * 2048.0) if f else None,
                             'write_bytes_per_launch': (sum(w) / len(w) * 1024.0) if w else None,
                             'fetch_bytes_max_launch_corrected': (max(f) * 2048.0) if f else None,
                             'write_bytes_max_launch': (max(w) * 1024.0) if w else None}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
