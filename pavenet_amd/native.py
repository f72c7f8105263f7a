"""ctypes binding of libpave_hip.so (C ABI declared in include/pave_hip.h).

The library is built in-tree by ``pavenet_amd.build_native`` (hipcc,
--offload-arch=gfx950).  There is NO fallback: if the shared object is missing
or a symbol is absent, importing the product ops raises.
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libpave_hip.so')
DIAG_LIB_PATH = os.path.join(_HERE, 'lib', 'libpave_hip_diag.so')   # -DPAVE_DIAG build (tests/, tools/)

_c_int = ctypes.c_int
_vp = ctypes.c_void_p

# name -> argtypes; every entry point returns int (0 = PAVE_OK)
SIGNATURES = {
    'pave_ms_deform_attn_forward_f32': [_vp] * 6 + [_c_int] * 8 + [_vp],
    'pave_ms_deform_attn_forward_f64': [_vp] * 6 + [_c_int] * 8 + [_vp],
    'pave_deform_attn_grid_fused_f32': [_vp] * 10 + [_c_int] * 8 + [_vp, _c_int, _c_int, _vp],
    'pave_deform_attn_pose_fused_f32': [_vp] * 8 + [_c_int] * 7 + [_vp, _c_int, _c_int, _vp],
    'pave_fuse_sum_nhwc_f32': [_vp, _c_int] * 4 + [_vp] + [_c_int] * 5 + [_vp],
    'pave_bias_act_rows_f32': [_vp] * 4 + [ctypes.c_longlong, _c_int, _c_int, _vp],
    'pave_fill_rows_f32': [_vp, ctypes.c_longlong, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, _c_int, _vp],
    'pave_bias_add_layernorm_f32': [_vp] * 6 + [ctypes.c_longlong, _c_int, ctypes.c_float, _vp],
    'pave_bias_add_layernorm_pos_f32': [_vp] * 7 + [ctypes.c_longlong, _vp, ctypes.c_longlong, _c_int,
                                        ctypes.c_float, _vp],
    'pave_enc_deform_attn_tile_f32': [_vp] * 4 + [_c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp],
    'pave_ms_deform_attn_backward_f32': [_vp] * 9 + [_c_int] * 8 + [_vp],
    'pave_ms_deform_attn_backward_f64': [_vp] * 9 + [_c_int] * 8 + [_vp],
    'pave_preprocess_frames': [_vp, _c_int, _vp] + [_c_int] * 7 + [_vp, _vp, _c_int, _vp],
    'pave_conv3x3_nhwc_f32': [_vp] * 4 + [_c_int] * 7 + [_vp],
    'pave_rows_gemm_bias_res_act_f32': [_vp] * 7 + [ctypes.c_longlong] + [_c_int] * 4 + [_vp],
    'pave_bias_relu_maxpool_nhwc_f32': [_vp] * 3 + [_c_int] * 4 + [_vp],
    'pave_gemm_bf16x3_f32': [_vp] * 6 + [ctypes.c_longlong] + [_c_int] * 4 + [_vp],
    'pave_gemm_fp16_act_f32': [_vp, _c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _c_int,
                               ctypes.c_longlong, _c_int, _c_int, _c_int, _vp],
    'pave_gemm_bf16x3_ex_f32': [_vp] * 5 + [ctypes.c_longlong, _vp, _vp, _c_int, ctypes.c_longlong]
                               + [_c_int] * 4 + [_vp],
    'pave_gemm_bf16x3_cat_f32': [_vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _c_int, _c_int,
                                 _c_int, _c_int, _vp],
    'pave_gemm_bf16x3_grouped_f32': [_vp, ctypes.c_longlong, _vp, _vp, _vp, ctypes.c_longlong, _c_int, _c_int,
                                     _c_int, _c_int, _c_int, _vp],
    'pave_gemm_bf16x3_ln_f32': [_vp] * 6 + [ctypes.c_float, _vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _vp],
    'pave_groupnorm_nhwc_f32': [_vp] * 4 + [ctypes.c_longlong] + [_c_int] * 4 + [ctypes.c_float, _vp,
                                _c_int, _vp, _vp],
    'pave_groupnorm_levels_nhwc_f32': [_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp],
    'pave_ref_update_f32': [_vp, _vp, _vp, ctypes.c_longlong, ctypes.c_float, _vp],
    'pave_conv7x7s2_nchw_split_f32': [_vp] * 4 + [_c_int] * 7 + [_vp],
    'pave_repitch_rows_f32': [_vp, _vp, ctypes.c_longlong, _c_int, _c_int, _vp],
    'pave_conv1x1_strided_split_f32': [_vp] * 4 + [_c_int] * 8 + [_vp],
    'pave_split_bf16x3_f32': [_vp, _vp, ctypes.c_longlong, _c_int, _vp],
    'pave_conv3x3_split_f32': [_vp] * 5 + [_c_int] * 8 + [_vp],
    'pave_bottleneck_chain_f32': [_vp] * 8 + [_c_int, _vp, _vp, _vp, _vp] + [_c_int] * 5 + [_vp],
    'pave_conv3x3_splitk_f32': [_vp] * 5 + [_c_int] * 7 + [_vp, ctypes.c_longlong, _c_int, _vp],
    'pave_gemm_bf16x3_encproj_f32': [_vp, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp,
                                     ctypes.c_longlong, _c_int, _c_int, _vp],
    'pave_conv3x3s2_c3_nchw_f32': [_vp] * 4 + [_c_int] * 4 + [_vp],
    'pave_mha_core_f32': [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    'pave_topk_rows_f32': [_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp],
    'pave_gather_frame_poses_f32': [_vp, _vp, _vp] + [_c_int] * 5 + [_vp],
    'pave_pose_finalize_f32': [_vp] * 7 + [_c_int] * 5 + [_vp],
    'pave_ref_update_frames_f32': [_vp, _vp, _vp] + [_c_int] * 5 + [ctypes.c_float, _vp],
    'pave_swin_window_attn_f32': [_vp] * 4 + [_c_int] * 7 + [ctypes.c_float, _vp],
    'pave_merge_softmax_partials_f32': [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp],
    'pave_gemm_bf16x3_splitk_f32': [_vp] * 5 + [ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int, _vp,
                                    ctypes.c_longlong, _vp],
    'pave_gather_rows_add_f32': [_vp, ctypes.c_longlong, _vp, _vp, _vp, _vp] + [_c_int] * 4 + [_vp],
    'pave_proposal_refs_f32': [_vp, _c_int, _vp, ctypes.c_longlong, _vp, _vp] + [_c_int] * 5 + [_vp],
    'pave_oks_nms_f32': [_vp] * 3 + [ctypes.c_double] + [_vp] * 2 + [_c_int] * 3 + [_vp],
}
# every symbol include/pave_hip.h declares
EXPORTED = tuple(SIGNATURES) + ('pave_abi_version', 'pave_last_error', 'pave_conv3x3_splitk_workspace_bytes',
                                'pave_gemm_splitk_workspace_bytes')



class GnLevel(ctypes.Structure):
    """`pave_gn_level` of include/pave_hip.h (one map of pave_groupnorm_levels_nhwc_f32)."""
    _fields_ = [('x', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('y', ctypes.c_void_p),
                ('y_batch_stride', ctypes.c_longlong), ('HW', ctypes.c_int), ('nchunks', ctypes.c_int),
                ('eps', ctypes.c_float)]


_lib = None
ABI_VERSION = 20  # == PAVE_ABI_VERSION of include/pave_hip.h this file's SIGNATURES were written for


class NativeLibraryError(RuntimeError):
    pass


def _open(path):
    if not os.path.exists(path):
        raise NativeLibraryError(
            f'{path} not found: run `python -c "import __graft_entry__ as g; '
            f'g.build()"` (hipcc --offload-arch=gfx950). pavenet_amd has no '
            f'CPU or eager fallback for its HIP kernels.')
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _c_int
    lib.pave_abi_version.restype = _c_int
    lib.pave_abi_version.argtypes = []
    lib.pave_last_error.restype = ctypes.c_char_p
    lib.pave_last_error.argtypes = []
    lib.pave_conv3x3_splitk_workspace_bytes.restype = ctypes.c_longlong
    lib.pave_conv3x3_splitk_workspace_bytes.argtypes = [_c_int] * 6
    lib.pave_gemm_splitk_workspace_bytes.restype = ctypes.c_longlong
    lib.pave_gemm_splitk_workspace_bytes.argtypes = [ctypes.c_longlong, _c_int, _c_int]
    have = lib.pave_abi_version()
    if have != ABI_VERSION:   # a stale .so called with the wrong argument list corrupts memory
        raise NativeLibraryError(
            f'{path} has ABI version {have}, this package expects {ABI_VERSION}: rebuild it '
            f'(`python -m pavenet_amd.build_native`)')
    return lib


def load():
    """Load libpave_hip.so (once) and type its entry points."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


_diag_lib = None


@contextlib.contextmanager
def diag_build(variant=0):
    """tests/ and tools/ only: inside the block every op of this package runs on the -DPAVE_DIAG
    build of the same sources with kernel-form override `variant` (pave_gemm_split.hip lists the
    values); yields that library (it also has pave_diag_enc_tile_ablate).  The shipped library
    has no such switch, and is back in place when the block ends."""
    global _lib, _diag_lib
    if _diag_lib is None:
        _diag_lib = _open(DIAG_LIB_PATH)
        _diag_lib.pave_diag_gemm_variant.argtypes = [_c_int]
        _diag_lib.pave_diag_gemm_variant.restype = None
    product = load()
    _diag_lib.pave_diag_gemm_variant(int(variant))
    _lib = _diag_lib
    try:
        yield _diag_lib
    finally:
        _diag_lib.pave_diag_gemm_variant(0)
        _lib = product


def use_diag_build(variant=0):
    """tools/ only: run the rest of this process on the -DPAVE_DIAG build with override `variant`."""
    global _lib
    with diag_build(variant) as lib:
        pass
    lib.pave_diag_gemm_variant(int(variant))
    _lib = lib
    return lib


_SYNC_DEBUG = os.environ.get('PAVE_SYNC_DEBUG', '0') == '1'


def check(status, what):
    if status != 0:
        msg = load().pave_last_error().decode()
        raise RuntimeError(f'{what} failed with status {status}: {msg}')
    if _SYNC_DEBUG:   # fault localisation: every native launch is drained and reported
        import sys

        import torch
        torch.cuda.synchronize()
        print(f'[pave] {what} ok', file=sys.stderr, flush=True)
