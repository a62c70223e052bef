"""Wave-level counters of the K8 far search on view geometry (needs `python tools/experiment.py knn_profile`):
candidates, parked-insert flushes, cells / coarse cells / blocks opened and clock ticks per far wave."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
os.environ['NERFAIL_HIP_LIB'] = os.path.join(ROOT, 'nerfail_amd', 'lib', 'libnerfail_hip_exp_knn_profile.so')
import synth
from nerfail_amd import _lib
from nerfail_amd.create_index_and_dist import index_and_dist
dev = torch.device('cuda:0')
H = W = 800
S = torch.from_numpy(np.stack([synth.sphere_view_points(H, W, th) for th in (-120., 0., 120.)]).reshape(-1, 3)).to(dev)
Q = torch.from_numpy(synth.sphere_view_points(H, W, 45.)).to(dev)
lib = _lib.load()
index_and_dist(Q, S)
NW = (H // 8) * (W // 8)
stats = torch.zeros((16 + 9 * NW,), dtype=torch.int64, device=dev)
lib.nerfail_knn8_grid_stats(_lib.dev(stats))
index_and_dist(Q, S)
torch.cuda.synchronize()
lib.nerfail_knn8_grid_stats(None)
v = stats.cpu().tolist()
nw = max(v[9], 1)
print('far waves %d; per far wave: points scanned %.0f, flushes %.0f, cells %.1f, coarse cells %.1f, blocks %.1f' %
      (v[9], v[2] / nw, v[3] / nw, v[4] / nw, v[5] / nw, v[6] / nw))
print('clock ticks (100 MHz) per far wave: whole kernel %.0f, inside the point scans %.0f  -> %.1f ns per scanned point'
      % (v[8] / nw, v[7] / nw, v[7] * 10.0 / max(v[2], 1)))
print('per far wave: %.1f point loads waiting %.1f us in all (%.2f us each); %.0f slow trips (%.0f points survive the box prefilter) taking %.1f us in all (%.2f us each)'
      % (v[11] / nw, v[10] / nw / 100.0, v[10] / max(v[11], 1) / 100.0, v[12] / nw, v[14] / nw, v[13] / nw / 100.0, v[13] / max(v[12], 1) / 100.0))
a = np.array(v[16:16 + 5 * NW], dtype=np.float64).reshape(5, NW)
hw = np.array(v[16 + 4 * NW:16 + 5 * NW], dtype=np.int64)
t, pts, nfar = a[0] / 100.0, a[1], a[2]          # us
m = nfar > 0
print('per far wave us: median %.0f  p90 %.0f  p99 %.0f  max %.0f;  points: median %.0f p99 %.0f max %.0f' %
      (np.median(t[m]), np.percentile(t[m], 90), np.percentile(t[m], 99), t[m].max(), np.median(pts[m]), np.percentile(pts[m], 99), pts[m].max()))
order = np.argsort(-t)[:12]
for w in order:
    print('  wave %5d (row %3d, col %3d..)  far lanes %2d  points %7d  %.0f us' % (w, (w // (W // 8)) * 8, (w % (W // 8)) * 8, nfar[w], pts[w], t[w]))
print('sum of far-wave time %.1f ms-waves; by far-lane count: ' % (t[m].sum() / 1e3) +
      ', '.join('%d-%d lanes: %d waves %.0f us avg' % (lo, hi, ((nfar >= lo) & (nfar <= hi)).sum(), t[(nfar >= lo) & (nfar <= hi)].mean())
                for lo, hi in ((1, 8), (9, 32), (33, 63), (64, 64)) if ((nfar >= lo) & (nfar <= hi)).any()))
t0 = a[3]
base = t0[m].min()
st, en = (t0[m] - base) / 100.0, (t0[m] - base) / 100.0 + t[m]
print('far waves start between %.0f and %.0f us after the first, the last ends at %.0f us' % (st.min(), st.max(), en.max()))
for lo in range(0, int(en.max()) + 1, 250):
    mid = lo + 125.0
    print('   t = %5.0f us: %4d far waves in flight' % (mid, int(((st <= mid) & (en > mid)).sum())))
hwm = hw[m]
xcc, hid = (hwm >> 32) & 0xf, hwm & 0xffffffff
cu, sh, se, simd = (hid >> 8) & 0xf, (hid >> 12) & 1, (hid >> 13) & 7, (hid >> 4) & 3
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print('distinct CUs seen by far waves: %d; distinct (xcc) %d, (se) %d, (sh) %d, (cu) %d' % (len(np.unique(cuid)), len(np.unique(xcc)), len(np.unique(se)), len(np.unique(sh)), len(np.unique(cu))))
for mid in (125., 625., 1125., 2375.):
    fl = (st <= mid) & (en > mid)
    c = np.bincount(np.unique(cuid, return_inverse=True)[1][fl], minlength=len(np.unique(cuid)))
    print('   t = %5.0f us: far waves in flight per CU: min %d median %d max %d' % (mid, c.min(), np.median(c), c.max()))
started = np.sort(st)
print('far waves started by 50 us: %d, 250 us: %d, 1000 us: %d, 2000 us: %d' % tuple(int((started <= x).sum()) for x in (50, 250, 1000, 2000)))
ex = np.array(v[16 + 5 * NW:16 + 6 * NW], dtype=np.float64)
nm = ~m
print('waves WITHOUT far lanes: %d; us: median %.0f p90 %.0f p99 %.0f max %.0f; sum %.1f ms-waves; candidates per lane: median %.0f p99 %.0f max %.0f'
      % (nm.sum(), np.median(t[nm]), np.percentile(t[nm], 90), np.percentile(t[nm], 99), t[nm].max(), t[nm].sum() / 1e3,
         np.median(ex[nm]) / 64, np.percentile(ex[nm], 99) / 64, ex[nm].max() / 64))
for w in np.argsort(-np.where(nm, t, 0))[:8]:
    print('  wave %5d (row %3d, col %3d..)  candidates per lane %7.0f  %.0f us' % (w, (w // (W // 8)) * 8, (w % (W // 8)) * 8, ex[w] / 64, t[w]))
allst, allen = (t0 - t0.min()) / 100.0, (t0 - t0.min()) / 100.0 + t
for mid in (125., 625., 1125., 1625., 2375., 3125.):
    fl = (allst <= mid) & (allen > mid)
    print('   t = %5.0f us: %4d waves in flight (%d far)' % (mid, fl.sum(), (fl & m).sum()))
sh = np.array(v[16 + 6 * NW:16 + 7 * NW], dtype=np.float64) / 100.0
print('shell phase per wave us: all waves median %.0f p90 %.0f p99 %.0f max %.0f, sum %.1f ms-waves; far waves: median %.0f p99 %.0f sum %.1f ms-waves'
      % (np.median(sh), np.percentile(sh, 90), np.percentile(sh, 99), sh.max(), sh.sum() / 1e3, np.median(sh[m]), np.percentile(sh[m], 99), sh[m].sum() / 1e3))
for w in np.argsort(-sh)[:8]:
    print('  wave %5d (row %3d, col %3d..)  far lanes %2d  candidates per lane %7.0f  shell phase %.0f us of %.0f' % (w, (w // (W // 8)) * 8, (w % (W // 8)) * 8, nfar[w], ex[w] / 64, sh[w], t[w]))
c7 = np.array(v[16 + 7 * NW:16 + 8 * NW], dtype=np.int64)
c8 = np.array(v[16 + 8 * NW:16 + 9 * NW], dtype=np.int64)
cells, coarse, blocks, flushes, scan_us = c7 & 0xfffff, (c7 >> 20) & 0xfffff, c7 >> 40, c8 & 0xffffff, (c8 >> 24) / 100.0
print('slowest waves in detail:')
for w in np.argsort(-t)[:10]:
    print('  wave %5d (row %3d, col %3d): %5.0f us, %5.0f in scans; points %6d in %4d cells (%3d coarse, %2d blocks), %4d flushes; started at %.0f us'
          % (w, (w // (W // 8)) * 8, (w % (W // 8)) * 8, t[w], scan_us[w], pts[w], cells[w], coarse[w], blocks[w], flushes[w], allst[w]))
k = np.argsort(t)[len(t) // 2 - 3:len(t) // 2 + 3]
print('median waves:')
for w in k:
    print('  wave %5d (row %3d, col %3d): %5.0f us, %5.0f in scans; points %6d in %4d cells (%3d coarse, %2d blocks), %4d flushes; started at %.0f us'
          % (w, (w // (W // 8)) * 8, (w % (W // 8)) * 8, t[w], scan_us[w], pts[w], cells[w], coarse[w], blocks[w], flushes[w], allst[w]))
