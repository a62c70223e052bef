#!/usr/bin/env python3
"""Reduce one rocprofv3 --pmc pass of SQ / GRBM counters (+ --kernel-trace of the same run) to per-kernel means and the
derived figures quoted in DESIGN.md:

  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)    share of SIMD-cycles with the matrix pipe busy
                (GRBM_GUI_ACTIVE is summed over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'; MFMA_BUSY over all SIMDs)
  clock_GHz   = GRBM_GUI_ACTIVE / 8 / kernel duration                             the clock the chip held during the dispatch
  wait_any / wait_inst / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES

    python tools/pmc_sq_summary.py <dir> "<command>" [kernel substring ...] > profiles/<name>.json
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d, cmd, want = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values per dispatch
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'nerfail' in k and (not want or any(w in k for w in want)):
                per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k in per:
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    try:                                       # ties the measurement to the kernel sources it was taken from (as pmc_summary.py does)
        import bench_sections as B
        files = B.kernel_source_hashes()
    except Exception:
        files = {}
    out = {'command': cmd, 'csrc_files': files, 'units': 'means per dispatch; SQ_* wave counters in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles summed '
                                    'over all SIMDs, GRBM_GUI_ACTIVE summed over the 8 XCDs', 'kernels': {}}
    for k, cs in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {'dispatches': max(len(v) for v in cs.values()), 'counters': m}
        if dur.get(k):
            e['seconds_per_dispatch_under_pmc'] = sum(dur[k]) / len(dur[k])
        gui = m.get('GRBM_GUI_ACTIVE')
        if gui and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            e['mfma_busy'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * gui / 8.0)
        if gui and dur.get(k):
            e['clock_GHz'] = gui / 8.0 / e['seconds_per_dispatch_under_pmc'] / 1e9
        wc = m.get('SQ_WAVE_CYCLES')
        if wc:
            for name, c in (('wait_any', 'SQ_WAIT_ANY'), ('wait_inst', 'SQ_WAIT_INST_ANY'), ('active', 'SQ_ACTIVE_INST_ANY')):
                if c in m:
                    e[name] = m[c] / wc
        out['kernels'][k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
