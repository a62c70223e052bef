import os
import sys


def _usable_cpus():
    """CPUs this process may use: cgroup quota (cpu.max) if set, else the affinity mask (same rule as bench.py)."""
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


# size the OpenMP / OpenBLAS pools to the cgroup quota before numpy / torch create them (the GPU boxes report 256 CPUs
# and grant 16: oversubscribed pools get the whole process throttled)
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, str(_usable_cpus()))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        return cache[name]
    return load


def rel_err(a, b, floor=1e-6):
    """max |a-b| / max(|b|, floor-scaled magnitude): the 'relative fp32' measure used in all parity tests."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), 'NaN positions differ'
    m = ~nan_b
    if not m.any():
        return 0.0
    scale = np.maximum(np.abs(b[m]), floor + 1e-3 * np.abs(b[m]).max())
    return float((np.abs(a[m] - b[m]) / scale).max())


def l2_err(a, b):
    """||a-b|| / ||b|| over the whole tensor (float64): the measure used for parameter gradients."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
