"""ctypes binding of libnerfail_hip.so (the C ABI declared in include/nerfail_hip.h).

There is NO CPU fallback anywhere in this package: if the library is missing, or a tensor is not a
contiguous float32 HIP tensor, the call raises. PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: libnerfail_hip.so binds to torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NERFAIL_HIP_LIB') or os.path.join(_HERE, 'lib', 'libnerfail_hip.so')   # override: A/B builds (tools/ablate.py)

ABI_VERSION = 7
MAX_DEPTH = 16
DW_BF16X3, DW_ACCUMULATE = 1, 2          # flags of nerfail_mlp_bwd_weights
RAY_FLOATS = 11

c_f = ctypes.c_float
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_p = ctypes.c_void_p


class NerfailError(RuntimeError):
    """A libnerfail_hip call returned a NERFAIL_E* code."""


class MlpParams(ctypes.Structure):
    """struct nerfail_mlp_params (include/nerfail_hip.h)."""
    _fields_ = [('D', ctypes.c_int32), ('W', ctypes.c_int32), ('input_ch', ctypes.c_int32),
                ('input_ch_views', ctypes.c_int32), ('skip', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('pts_w', c_p * MAX_DEPTH), ('pts_b', c_p * MAX_DEPTH),
                ('views_w', c_p), ('views_b', c_p), ('feature_w', c_p), ('feature_b', c_p),
                ('alpha_w', c_p), ('alpha_b', c_p), ('rgb_w', c_p), ('rgb_b', c_p)]


# name -> (restype, argtypes); every symbol include/nerfail_hip.h declares
SIGNATURES = {
    'nerfail_abi_version': (c_i, []),
    'nerfail_last_error': (ctypes.c_char_p, []),
    'nerfail_device_name': (c_i, [ctypes.c_char_p, ctypes.c_size_t]),
    'nerfail_get_rays': (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'nerfail_pack_rays': (c_i, [c_p, c_p, c_i64, c_f, c_f, c_p, c_p]),
    'nerfail_ray_gen': (c_i, [c_i, c_i, c_p, c_p, c_f, c_f, c_i64, c_i64, c_p, c_p]),
    'nerfail_sample_coarse': (c_i, [c_p, c_i64, c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    'nerfail_sample_pdf': (c_i, [c_p, c_p, c_i64, c_i, c_p, c_i, c_i, c_p, c_p]),
    'nerfail_sample_fine': (c_i, [c_p, c_i64, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    'nerfail_embed': (c_i, [c_p, c_i64, c_i, c_p, c_p]),
    'nerfail_mlp_packed_floats': (ctypes.c_size_t, [c_i, c_i, c_i]),
    'nerfail_mlp_pack': (c_i, [ctypes.POINTER(MlpParams), c_p, c_p]),
    'nerfail_mlp_fwd': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p]),
    'nerfail_mlp_fwd_rays': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p, c_p]),
    'nerfail_mlp_fwd_select': (c_i, [c_i]),
    'nerfail_mlp_bwd_select': (c_i, [c_i]),
    'nerfail_mlp_fwd_embedded': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i64, c_p, c_p]),
    'nerfail_mlp_f16_image_bytes': (ctypes.c_size_t, [c_i, c_i, c_i]),
    'nerfail_mlp_pack_f16': (c_i, [ctypes.POINTER(MlpParams), c_p, c_p]),
    'nerfail_mlp_fwd_f16': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p]),
    'nerfail_mlp_fwd_f16_train': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p, c_p]),
    'nerfail_mlp_f16_image_T_bytes': (ctypes.c_size_t, [c_i, c_i, c_i]),
    'nerfail_mlp_pack_f16_T': (c_i, [ctypes.POINTER(MlpParams), c_p, c_p]),
    'nerfail_mlp_bwd_data_f16': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_p, c_p]),
    'nerfail_mlp_train_acts_floats': (ctypes.c_size_t, [c_i, c_i, c_i64]),
    'nerfail_mlp_train_dz_floats': (ctypes.c_size_t, [c_i, c_i, c_i64]),
    'nerfail_mlp_fwd_train': (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p, c_p]),
    'nerfail_mlp_packed_T_floats': (ctypes.c_size_t, [c_i, c_i, c_i]),
    'nerfail_mlp_pack_T': (c_i, [ctypes.POINTER(MlpParams), c_p, c_p]),
    'nerfail_mlp_pack_train': (c_i, [ctypes.POINTER(MlpParams), c_p, c_p, c_p]),
    'nerfail_mlp_bwd_data': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_i64, c_p, c_p]),
    'nerfail_mlp_bwd_data2': (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    'nerfail_mlp_bwd_weights_scratch_bytes': (ctypes.c_size_t, [c_i, c_i, c_i, c_i64, c_i64, c_i]),
    'nerfail_mlp_bwd_weights': (c_i, [c_i, c_i, c_i, c_p, c_p, c_i64, ctypes.POINTER(MlpParams), c_i64, ctypes.POINTER(MlpParams),
                                      c_i, c_p, ctypes.c_size_t, c_p]),
    'nerfail_composite': (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nerfail_composite_select': (c_i, [c_i]),
    'nerfail_composite_bwd': (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nerfail_knn8': (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p]),
    'nerfail_knn8_grid_workspace_bytes': (ctypes.c_size_t, [c_i64]),
    'nerfail_knn8_grid': (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    'nerfail_knn8_grid_build': (c_i, [c_p, c_i64, c_p, ctypes.c_size_t, c_p]),
    'nerfail_knn8_grid_stats': (c_i, [c_p]),
    'nerfail_knn8_grid_search': (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    'nerfail_knn8_grid_search_view': (c_i, [c_p, c_i, c_i, c_i64, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    'nerfail_gauss_weight': (c_i, [c_p, c_i64, c_i64, c_f, c_p, c_p]),
    'nerfail_gauss_compose': (c_i, [c_p, c_p, c_i64, c_p, c_p]),
    'nerfail_gauss_fwd': (c_i, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_f, c_p, c_p, c_p, c_p]),
    'nerfail_gauss_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_p, c_p]),
    'nerfail_gauss_fwd_views': (c_i, [c_p, c_i64, c_p, c_i, c_i64, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p]),
    'nerfail_gauss_bwd_views_rgb': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i64, c_p, c_p, c_p]),
    'nerfail_igsm_step_rgb': (c_i, [c_p, c_p, c_p, c_i64, c_f, c_f, c_i, c_p, c_p]),
    'nerfail_gauss_bwd_views_rgb_step': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i64, c_i64, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p]),
    'nerfail_fingerprint': (c_i, [c_p, c_i64, c_i64, c_p, c_p]),
    'nerfail_gauss_csr_workspace_bytes': (ctypes.c_size_t, [c_i64, c_i64, c_i64]),
    'nerfail_gauss_csr_build': (c_i, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    'nerfail_gauss_bwd_scratch_floats': (ctypes.c_size_t, [c_i64, c_i64, c_i]),
    'nerfail_gauss_view_pack_workspace_bytes': (ctypes.c_size_t, [c_i64]),
    'nerfail_gauss_view_chunks': (c_i64, [c_i64]),
    'nerfail_gauss_view_pack': (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    'nerfail_gauss_bwd_view_multi': (c_i, [c_p, c_p, c_p, c_i, c_p, c_i64, c_i64, c_f, c_p, c_p, c_p]),
    'nerfail_gauss_bwd_views_scratch_floats': (ctypes.c_size_t, [c_p, c_i, c_i64, c_i]),
    'nerfail_gauss_bwd_csr': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_p, c_i, c_p, c_p]),
    'nerfail_gauss_bwd_views': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i64, c_f, c_p, c_p, c_p]),
    'nerfail_gauss_bwd_csr_multi': (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_f, c_p, c_p, c_p]),
    'nerfail_deepfool_norms_scratch_bytes': (ctypes.c_size_t, [c_i, c_i64]),
    'nerfail_deepfool_norms': (c_i, [c_p, c_i, c_i64, c_p, ctypes.c_size_t, c_p, c_p]),
    'nerfail_deepfool_apply': (c_i, [c_p, c_i, c_i64, c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    'nerfail_igsm_step': (c_i, [c_p, c_p, c_p, c_i64, c_f, c_f, c_i, c_p, c_p]),
    'nerfail_adam_step': (c_i, [c_p, c_i, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_p]),
    'nerfail_mse': (c_i, [c_p, c_p, c_i64, c_p, c_p, c_p]),
}



class ViewIndexStruct(ctypes.Structure):
    """struct nerfail_view_index (include/nerfail_hip.h)"""
    _fields_ = [('packed', c_p), ('w_sorted', c_p), ('chunk_ord', c_p), ('pos', c_p), ('n_entries', c_i64), ('n_rows', c_i64)]


class ViewFwdStruct(ctypes.Structure):
    """struct nerfail_view_fwd (include/nerfail_hip.h)"""
    _fields_ = [('weight_and_index', c_p), ('ori_img', c_p)]


class AdamTensor(ctypes.Structure):
    """struct nerfail_adam_tensor (include/nerfail_hip.h)"""
    _fields_ = [('param', c_p), ('grad', c_p), ('exp_avg', c_p), ('exp_avg_sq', c_p), ('numel', c_i64),
                ('step_size', c_f), ('bias_correction2_sqrt', c_f)]


_lib = None


def load():
    """Load the library once; raise (never fall back) when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('nerfail_amd: %s is missing - build it with `python -m nerfail_amd.build` '
                           '(hipcc, gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = the .so is stale: rebuild
        fn.restype = res
        fn.argtypes = args
    v = lib.nerfail_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError('nerfail_amd: ABI version %d, expected %d - rebuild the library' % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().nerfail_last_error().decode('utf-8', 'replace')
        raise NerfailError('libnerfail_hip error %d: %s' % (rc, msg))


def dev(t, name='tensor'):
    """Pointer of a dense float32 HIP tensor; anything else is an error (no silent copies / fallbacks)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s is on %s: nerfail_amd runs on the MI355X only (no CPU path)' % (name, t.device))
    if t.dtype not in (torch.float32, torch.int32, torch.uint8, torch.int64):
        raise TypeError('%s must be float32 (got %s)' % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return c_p(t.data_ptr())


def host_floats(values):
    arr = (c_f * len(values))(*[float(v) for v in values])
    return arr


def stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


def f32c(t, device=None):
    """Detached contiguous float32 view/copy of t on `device` (the reference's .float() / .to(device))."""
    t = t.detach()
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def device_name():
    buf = ctypes.create_string_buffer(256)
    check(load().nerfail_device_name(buf, 256))
    return buf.value.decode()
