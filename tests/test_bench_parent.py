"""bench.py's parent process: it never imports torch, runs every section group in a fresh child, keeps what a child emitted
before it died, prints ONE short contract line on stdout (< 1800 bytes: round 4's 20 KB line was truncated by the driver's
capture and never parsed) and writes everything else to the detail file. `--gpus N` without a launcher spawns its own N
ranks. CPU only: the `selftest` group has no GPU work, and NERFAIL_BENCH_FAKE_RENDER stands in for the render child."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
FULL = os.path.join(ROOT, 'tests', 'data', 'bench_detail_full.json')     # a fully populated run (round 4, every section)
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def run(tmp_path, extra_env=None, args=('--sections', 'selftest'), launcher=None):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    detail = str(tmp_path / 'detail.json')
    env.update({'NERFAIL_BENCH_DETAIL': detail})
    env.update(extra_env or {})
    cmd = (launcher or [sys.executable]) + [BENCH] + list(args)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=180)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) <= 1, r.stdout + r.stderr              # the contract: ONE JSON line on stdout
    line = json.loads(lines[0]) if lines else None
    if line is not None:
        assert len(lines[0].encode()) < 1800, len(lines[0])
    return r, line, (json.load(open(detail)) if os.path.exists(detail) else None)


def test_parent_collects_a_healthy_child(tmp_path):
    r, line, detail = run(tmp_path)
    assert r.returncode == 0
    assert detail['selftest'] == {'before': 1, 'after': 2}
    assert detail['sections']['selftest']['rc'] == 0 and 'selftest_error' not in detail
    for k in CONTRACT:
        assert k in line
    assert line['detail'].endswith('detail.json')


def test_a_killed_child_keeps_its_partial_results_and_becomes_an_error_key(tmp_path):
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_SELFTEST_DIE': '1'})      # the child SIGKILLs itself after its first object
    assert detail['selftest'] == {'before': 1}
    assert 'rc -9' in detail['selftest_error'] and detail['sections']['selftest']['rc'] == -9
    assert any('selftest_error' in e for e in line['errors'])
    assert r.returncode == 0                                 # no render section was asked for: nothing of the contract is missing


def test_fully_populated_run_gives_a_short_line_with_every_contract_key(tmp_path):
    """Every section's output present (the stored round-4 run, 20 KB as one object): the stdout line stays under the
    limit and still carries roofline, cpu_baseline and the flat fwd+bwd / attack scalars."""
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_FAKE_RENDER': FULL}, args=('--sections', 'render', '--steps', '20', '--warmup', '5'))
    assert r.returncode == 0, r.stderr
    stored = json.load(open(FULL))
    assert len(json.dumps(stored)) > 15000                   # the thing that did not fit last round
    for k in CONTRACT:
        assert line[k] is not None or k == 'vs_baseline', k
    assert line['value'] == pytest.approx(stored['value'], rel=1e-5) and line['steps'] == 20 and line['warmup'] == 5
    assert line['roofline']['bound'] == 'mfma' and line['roofline']['frac'] == pytest.approx(stored['roofline']['frac'], rel=1e-5)
    assert set(line['roofline']) >= {'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
    for k in ('fwd_bwd_rays_per_sec', 'fwd_bwd_ms_per_step', 'fwd_bwd_frac', 'attack_iters_per_sec', 'attack_ms_per_iter', 'attack_frac'):
        assert isinstance(line[k], float), k
    assert detail['attack']['gauss_kernels'] == stored['attack']['gauss_kernels']       # nothing is lost: it is in the file


def test_line_builder_always_fits(tmp_path):
    """contract_line() drops optional keys rather than exceed the limit, whatever the children reported."""
    code = ("import json, sys, types; sys.argv=['bench.py']; import importlib.util as u\n"
            "spec = u.spec_from_file_location('bench_parent_only', %r); src = open(%r).read().split(\"if __name__ == '__main__':\")[0]\n"
            "m = types.ModuleType('bench_parent_only'); m.__file__ = %r; exec(compile(src, %r, 'exec'), m.__dict__)\n"
            "d = json.load(open(%r)); d['x_error'] = 'e' * 5000; d['y_error'] = 'f' * 5000\n"
            "d['cpu_baseline']['sample'] = 's' * 9000; d['roofline']['kernel'] = 'k' * 300\n"
            "a = types.SimpleNamespace(steps=20, warmup=5)\n"
            "ln = m.contract_line(d, 1, a, '/some/long/path/' + 'p' * 200 + '/bench_detail.json')\n"
            "assert len(json.dumps(ln)) < m.LINE_LIMIT, len(json.dumps(ln)); assert ln['value'] and ln['roofline'] and ln['cpu_baseline']\n"
            "assert 'torch' not in sys.modules\n" % (BENCH, BENCH, BENCH, BENCH, FULL))
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


def test_dead_render_child_is_a_failed_run(tmp_path):
    """the parent used to exit 0 with "value": null when the render child died before the timed region ended."""
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_FAKE_RENDER': FULL, 'NERFAIL_BENCH_FAKE_RENDER_DIE': '0'}, args=('--sections', 'render'))
    assert r.returncode != 0
    assert line['value'] is None and 'rc -9' in detail['render_error']


def test_gpus_2_without_a_launcher_spawns_two_ranks(tmp_path):
    """`python bench.py --gpus 2` (WORLD_SIZE unset) starts its own ranks with torchrun's environment; the
    fake render children meet in a gloo all-reduce, so n_gpus is the number of ranks that really rendezvoused."""
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_FAKE_RENDER': FULL}, args=('--gpus', '2', '--sections', 'render'))
    assert r.returncode == 0, r.stderr
    assert line['n_gpus'] == 2 and detail['sections']['render']['ranks'] == [0, 0] and detail['local_rank_seen'] == 0


def test_gpus_2_with_a_dead_rank_is_a_failed_run(tmp_path):
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_FAKE_RENDER': FULL, 'NERFAIL_BENCH_FAKE_RENDER_DIE': '1'},
                          args=('--gpus', '2', '--sections', 'render'))
    assert 'rank 1' in detail['render_error']                # rank 0 delivered its line; the dead peer is reported, not hidden
    assert detail['sections']['render']['ranks'][1] == -9


def test_gpus_must_match_the_launchers_world_size(tmp_path):
    """Asked for 8, launched as 1: refuse (exit 2, nothing on stdout) instead of a line that says n_gpus 1."""
    r, line, _ = run(tmp_path, {'WORLD_SIZE': '1', 'RANK': '0', 'NERFAIL_BENCH_FAKE_RENDER': FULL}, args=('--gpus', '8', '--sections', 'render'))
    assert r.returncode == 2 and line is None and 'WORLD_SIZE' in r.stderr


def test_under_torch_distributed_run(tmp_path):
    """The driver's N > 1 form: one parent + one render child per rank, rank 0 prints the line."""
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                '--master-port', '29633']
    r, line, detail = run(tmp_path, {'NERFAIL_BENCH_FAKE_RENDER': FULL}, args=('--gpus', '2', '--sections', 'render'), launcher=launcher)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2


def test_parent_never_imports_torch():
    code = ("import sys, runpy; sys.argv = ['bench.py', '--sections', 'selftest'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules, 'the parent must stay GPU-free'\n" % BENCH)
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_unknown_section_is_refused():
    r = subprocess.run([sys.executable, BENCH, '--sections', 'nonsense'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode != 0 and 'unknown sections' in r.stderr
