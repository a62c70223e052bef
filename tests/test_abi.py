"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/nerfail_hip.h
declares, validates arguments with the documented error codes, and the package fails loudly without a GPU."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'nerfail_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nerfail_[A-Za-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from nerfail_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.nerfail_abi_version() == _lib.ABI_VERSION


def test_argument_validation_without_gpu():
    from nerfail_amd import _lib
    lib = _lib.load()
    assert lib.nerfail_mlp_packed_floats(8, 256, 4) == 601600     # 2320 + 16 pad weight pieces, 10 bias pieces, 2 + 2 head pieces
    assert lib.nerfail_mlp_packed_floats(4, 64, 4) == lib.nerfail_mlp_packed_floats(4, 64, -1)
    assert lib.nerfail_mlp_packed_floats(8, 100, 4) == 0            # unsupported width
    assert lib.nerfail_composite(None, None, None, None, 5, 0, 0, None, None, None, None, None, None, None, None) == 1
    assert b'n_samples' in lib.nerfail_last_error()
    assert lib.nerfail_knn8(None, 4, None, 3, None, None, None, None) == 1
    assert b'8 points' in lib.nerfail_last_error()
    assert lib.nerfail_pack_rays(None, None, 0, 2.0, 6.0, None, None) == 0   # empty input is a no-op
    assert lib.nerfail_igsm_step(None, None, None, 10, 2.0, 32.0, 0, None, None) == 1
    with pytest.raises(_lib.NerfailError):
        _lib.check(lib.nerfail_gauss_weight(None, 1, 1, -1.0, None, None))
    # sizing helpers are pure host functions
    assert lib.nerfail_mlp_packed_T_floats(8, 256, 4) == 9 * 32 * 8 * 256 - 16 * 8 * 256    # 8 full layers + views (half the quads)
    # per 32-sample tile: E0 E1 V + 8 layers and feature x 8 slots + 4 hv slots + 3 slots of ReLU bit masks
    assert lib.nerfail_mlp_train_acts_floats(8, 256, 64) == 2 * (3 + 9 * 8 + 4 + 3) * 1024
    # multi-RHS gauss backward and the DeepFool step kernels: sizes and NULL pointers are rejected before any launch
    assert lib.nerfail_gauss_bwd_csr_multi(None, None, None, 8, None, None, None, None, 10, 1, 10, -1.0, None, None, None) == 1
    assert lib.nerfail_deepfool_norms_scratch_bytes(8, 1000) == ((1000 + 2047) // 2048) * 7 * 8
    assert lib.nerfail_deepfool_norms_scratch_bytes(1, 1000) == 0 and lib.nerfail_deepfool_norms_scratch_bytes(9, 1000) == 0
    assert lib.nerfail_deepfool_norms(None, 8, 1000, None, 0, None, None) == 1
    assert lib.nerfail_deepfool_norms(None, 1, 1000, None, 0, None, None) == 1
    assert lib.nerfail_deepfool_apply(None, 8, 1000, None, None, 0.02, None, None, None, None) == 1
    assert lib.nerfail_deepfool_apply(None, 8, 0, None, None, 0.02, None, None, None, None) == 1
    # optimizer step: nothing to do / NULL table / NULL tensor pointers are rejected before any launch
    assert lib.nerfail_adam_step(None, 0, 0.9, 0.999, 1e-8, None) == 0
    assert lib.nerfail_adam_step(None, 2, 0.9, 0.999, 1e-8, None) == 1
    bad = _lib.AdamTensor()
    bad.numel, bad.bias_correction2_sqrt = 8, 1.0
    assert lib.nerfail_adam_step((_lib.AdamTensor * 1)(bad), 1, 0.9, 0.999, 1e-8, None) == 1
    assert lib.nerfail_adam_step((_lib.AdamTensor * 1)(bad), 1, 1.5, 0.999, 1e-8, None) == 1
    assert lib.nerfail_mlp_train_dz_floats(8, 256, 33) == 2 * (8 * 8 + 8 + 4 + 1) * 1024
    assert lib.nerfail_mlp_train_acts_floats(8, 100, 64) == 0
    assert lib.nerfail_knn8_grid_workspace_bytes(7) == 0 and lib.nerfail_knn8_grid_workspace_bytes(1 << 24) == 0
    assert lib.nerfail_knn8_grid_workspace_bytes(1920000) > 1920000 * 32
    assert lib.nerfail_gauss_csr_workspace_bytes(1920000, 8, 640000) >= 2 * 4 * 8 * 640000 * 8
    assert lib.nerfail_gauss_bwd_scratch_floats(8, 640000, 1) >= 8 * 640000 * 4 and lib.nerfail_gauss_bwd_scratch_floats(8, 640000, 9) == 0
    assert lib.nerfail_gauss_csr_workspace_bytes(10, 1 << 20, 1 << 20) == 0                 # ids would overflow 32 bits
    assert lib.nerfail_knn8_grid(None, 4, None, 100, None, None, None, None, 0, None) == 1
    assert lib.nerfail_composite_bwd(None, None, None, None, 3, 1, 1, None, None, None, None, None, None, None) == 1
    assert lib.nerfail_mlp_bwd_data(None, None, 8, 100, 4, None, None, 10, None, None) == 1
    assert b'unsupported' in lib.nerfail_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only behaviour')
def test_fails_loudly_without_gpu():
    from nerfail_amd.run_nerf import raw2outputs
    from nerfail_amd.GaussNet import create_gauss_w
    with pytest.raises(RuntimeError, match='no CPU path'):
        raw2outputs(torch.zeros(2, 64, 4), torch.zeros(2, 64), torch.ones(2, 3))
    with pytest.raises(RuntimeError, match='no CPU path'):
        create_gauss_w('cpu', 0.02)(torch.zeros(1, 2, 4, 4, 8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nerfail_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_mirror_signatures_match_reference():
    """Fixture g19 = inspect.signature of the reference's boundary functions (tests/golden/make_golden.py, reference imported
    there). Every mirror must start with exactly the reference's parameters (names, kinds, defaults); what a mirror adds
    (explicit random draws, view ids) must come AFTER them and be optional, so reference call sites keep working."""
    import inspect
    import json
    from nerfail_amd import run_nerf, run_nerf_helpers, nerf_to_coord, GaussNet, deepfool, load_blender
    mods = {'run_nerf': run_nerf, 'run_nerf_helpers': run_nerf_helpers, 'nerf_to_coord': nerf_to_coord, 'GaussNet': GaussNet,
            'deepfool': deepfool, 'load_blender': load_blender}
    table = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'g19_signatures.json')))
    assert len(table) >= 25
    for name, ref in sorted(table.items()):
        obj = mods[name.split('.')[0]]
        for part in name.split('.')[1:]:
            obj = getattr(obj, part)
        mine = [[p.name, str(p.kind), None if p.default is inspect.Parameter.empty else repr(p.default)]
                for p in inspect.signature(obj).parameters.values()]
        assert mine[:len(ref)] == ref, (name, ref, mine)
        for extra in mine[len(ref):]:
            assert extra[2] is not None or extra[1] in ('VAR_KEYWORD', 'VAR_POSITIONAL'), (name, extra)
