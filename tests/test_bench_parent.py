"""bench.py's parent process: it never imports torch, runs every section group in a fresh child, keeps
what a child emitted before it died and always prints ONE parseable JSON line. CPU only (the `selftest` group has no GPU work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run(extra_env=None, args=('--sections', 'selftest')):
    env = dict(os.environ)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, BENCH] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout + r.stderr              # the contract: ONE JSON line on stdout
    return r.returncode, json.loads(lines[0])


def test_parent_collects_a_healthy_child():
    rc, line = run()
    assert rc == 0
    assert line['selftest'] == {'before': 1, 'after': 2}
    assert line['sections']['selftest']['rc'] == 0 and 'selftest_error' not in line
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'metrics'):
        assert k in line


def test_a_killed_child_keeps_its_partial_results_and_becomes_an_error_key():
    rc, line = run({'NERFAIL_BENCH_SELFTEST_DIE': '1'})      # the child SIGKILLs itself after its first object
    assert line['selftest'] == {'before': 1}
    assert 'rc -9' in line['selftest_error'] and line['sections']['selftest']['rc'] == -9
    assert rc == 0                                           # no render section was asked for: nothing of the contract is missing


def test_parent_never_imports_torch():
    code = ("import sys, runpy; sys.argv = ['bench.py', '--sections', 'selftest'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules, 'the parent must stay GPU-free'\n" % BENCH)
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr


def test_unknown_section_is_refused():
    r = subprocess.run([sys.executable, BENCH, '--sections', 'nonsense'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode != 0 and 'unknown sections' in r.stderr
