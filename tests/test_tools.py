"""The timing experiments live OUTSIDE the shipped sources (tools/experiment.py patches a temp copy of one csrc file):
every pattern they substitute must still match the current source exactly once, or the instrument has rotted."""
import importlib.util
import os

from conftest import ROOT


def test_experiment_patches_still_apply():
    spec = importlib.util.spec_from_file_location('nf_experiment', os.path.join(ROOT, 'tools', 'experiment.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert len(mod.EXPERIMENTS) >= 10
    for name, (src, patches, flags) in mod.EXPERIMENTS.items():
        text = open(os.path.join(ROOT, 'nerfail_amd', 'csrc', src)).read()
        for old, new in patches:
            assert text.count(old) == 1, 'experiment %s: pattern no longer matches %s exactly once:\n%s' % (name, src, old)


def test_shipped_sources_hold_no_experiment_switches():
    """no NF_*_ABLATE macros or clock probes in the translation units that ship."""
    d = os.path.join(ROOT, 'nerfail_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        text = open(os.path.join(d, f)).read()
        assert '_ABLATE' not in text and 'clock64' not in text, f
