"""ctypes binding of libprismer_hip.so (C ABI: include/prismer_hip.h).

Loading never falls back to anything: if the shared library is missing or does not export every symbol the
header declares, importing this module raises.  All entry points take raw device pointers; `ptr(t)` turns a
torch tensor into one (torch is used for device memory and streams only).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be imported BEFORE the library: torch wheels bundle their own libamdhip64; loading
#                        ours first would bring up a second HIP runtime that cannot see torch's device/streams.

from . import build as _build

c_void_p, c_int, c_float, c_i64, c_u32, c_u64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint32, C.c_uint64

PH_OK = 0
ACT_NONE, ACT_QUICKGELU, ACT_RELU2, ACT_GELU, ACT_RELU, ACT_SAVED_GRAD = 0, 1, 2, 3, 4, 5


class RowMap(C.Structure):
    _fields_ = [('seg_in', c_int), ('seg_out', c_int), ('seg_off', c_int)]


IDENT = RowMap(0, 0, 0)


class ConvGather(C.Structure):
    _fields_ = [('B', c_int), ('H', c_int), ('W', c_int), ('C', c_int), ('ks', c_int), ('stride', c_int),
                ('kh', c_int), ('kw', c_int), ('off_y', c_int), ('off_x', c_int), ('Ho', c_int), ('Wo', c_int)]


class GemmArgs(C.Structure):
    _fields_ = [('A', c_void_p), ('B', c_void_p), ('C', c_void_p),
                ('M', c_int), ('N', c_int), ('K', c_int),
                ('lda', c_int), ('ldb', c_int), ('ldc', c_int),
                ('trans_a', c_int), ('trans_b', c_int),
                ('bias', c_void_p), ('act', c_int),
                ('pre_out', c_void_p),
                ('act_in', c_void_p), ('ld_act', c_int),
                ('residual', c_void_p), ('ldr', c_int),
                ('drop_p', c_float), ('drop_seed', c_void_p), ('drop_stream', c_u32),
                ('out_f32', c_int), ('accumulate', c_int), ('alpha', c_float), ('split_k', c_int),
                ('residual_f32', c_int), ('workspace', c_void_p), ('workspace_bytes', c_i64), ('pre_grad', c_int),
                ('conv', C.POINTER(ConvGather)), ('col_stats', c_void_p),
                ('rowmap_wo', c_int), ('rowmap_mul', c_int), ('rowmap_sub', c_int), ('rowmap_add', c_int), ('defer_reduce', c_int)]


class ConvDgradItem(C.Structure):
    _fields_ = [('w', c_void_p), ('dst', c_void_p), ('Cout', c_int), ('Cin', c_int), ('stride', c_int)]


class LnReduceItem(C.Structure):
    _fields_ = [('ws', c_void_p), ('blocks', c_int), ('D', c_int), ('dgamma', c_void_p), ('dbeta', c_void_p)]


class ColsumItem(C.Structure):
    _fields_ = [('x', c_void_p), ('out', c_void_p), ('M', c_int), ('N', c_int), ('ld', c_int)]


GEMM_GROUP_MAX = 16      # PH_GEMM_GROUP_MAX


class ConvLayoutItem(C.Structure):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('Cout', c_int), ('Cin', c_int), ('ks', c_int), ('Kp', c_int)]


CONV_GROUP_MAX = 32      # PH_CONV_GROUP_MAX


class BnItem(C.Structure):
    _fields_ = [('y', c_void_p), ('a', c_void_p), ('dy', c_void_p), ('M', c_i64), ('C', c_int),
                ('gamma', c_void_p), ('beta', c_void_p), ('running_mean', c_void_p), ('running_var', c_void_p),
                ('stats', c_void_p), ('sums', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p)]


BN_GROUP_MAX = 8         # PH_BN_GROUP_MAX
COLSTAT_SLABS = 8        # PH_COLSTAT_SLABS


class LayerNormFwdArgs(C.Structure):
    _fields_ = [('x', c_void_p), ('gamma', c_void_p), ('beta', c_void_p),
                ('y', c_void_p), ('y_map', RowMap), ('y2', c_void_p), ('y2_map', RowMap),
                ('mean', c_void_p), ('rstd', c_void_p), ('M', c_int), ('D', c_int), ('eps', c_float),
                ('x_f32', c_int), ('y_f32', c_void_p)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [('dy', c_void_p), ('dy_map', RowMap), ('dy2', c_void_p), ('dy2_map', RowMap),
                ('x', c_void_p), ('mean', c_void_p), ('rstd', c_void_p), ('gamma', c_void_p),
                ('dskip', c_void_p), ('dx', c_void_p), ('dx_drop', c_void_p),
                ('drop_p', c_float), ('drop_seed', c_void_p), ('drop_stream', c_u32),
                ('dgamma', c_void_p), ('dbeta', c_void_p), ('M', c_int), ('D', c_int),
                ('x_f32', c_int), ('partial_ws', c_void_p), ('partial_ws_bytes', c_i64), ('defer_reduce', c_int)]


class AttnFwdArgs(C.Structure):
    _fields_ = [('q', c_void_p), ('k', c_void_p), ('v', c_void_p), ('o', c_void_p),
                ('q_bs', c_i64), ('q_ts', c_i64), ('k_bs', c_i64), ('k_ts', c_i64),
                ('v_bs', c_i64), ('v_ts', c_i64), ('o_bs', c_i64), ('o_ts', c_i64),
                ('B', c_int), ('H', c_int), ('Sq', c_int), ('Sk', c_int), ('dh', c_int),
                ('scale', c_float), ('key_mask', c_void_p), ('causal', c_int),
                ('drop_p', c_float), ('drop_seed', c_void_p), ('drop_stream', c_u32),
                ('lse', c_void_p)]


class AttnBwdArgs(C.Structure):
    _fields_ = [('f', AttnFwdArgs), ('d_o', c_void_p), ('do_bs', c_i64), ('do_ts', c_i64),
                ('dq', c_void_p), ('dk', c_void_p), ('dv', c_void_p),
                ('dq_bs', c_i64), ('dq_ts', c_i64), ('dk_bs', c_i64), ('dk_ts', c_i64),
                ('dv_bs', c_i64), ('dv_ts', c_i64), ('delta', c_void_p)]


class EmbedFwdArgs(C.Structure):
    _fields_ = [('ids', c_void_p), ('B', c_int), ('T', c_int), ('H', c_int), ('pad_id', c_int),
                ('word', c_void_p), ('pos', c_void_p), ('type', c_void_p),
                ('gamma', c_void_p), ('beta', c_void_p), ('eps', c_float),
                ('out', c_void_p), ('xhat', c_void_p), ('rstd', c_void_p),
                ('drop_p', c_float), ('drop_seed', c_void_p), ('drop_stream', c_u32), ('out_f32', c_void_p)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [('f', EmbedFwdArgs), ('dout', c_void_p), ('dword', c_void_p), ('dpos', c_void_p),
                ('dtype', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p)]


_SIGS = {
    'ph_version': (c_int, []),
    'ph_last_error': (C.c_char_p, []),
    'ph_gemm_bf16': (c_int, [C.POINTER(GemmArgs), c_void_p]),
    'ph_layernorm_fwd': (c_int, [C.POINTER(LayerNormFwdArgs), c_void_p]),
    'ph_layernorm_bwd': (c_int, [C.POINTER(LayerNormBwdArgs), c_void_p]),
    'ph_attention_fwd': (c_int, [C.POINTER(AttnFwdArgs), c_void_p]),
    'ph_attention_bwd': (c_int, [C.POINTER(AttnBwdArgs), c_void_p]),
    'ph_patchify': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_resize_bilinear_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_gather_rows_bf16': (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    'ph_dense_minmax_partial': (c_int, [c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p]),
    'ph_resize_remap_nchw_to_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_inpaint_resize_nhwc': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_im2col_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'ph_col2im_nhwc': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_bn_stats': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'ph_bn_apply_relu_grouped': (c_int, [c_void_p, c_int, c_float, c_float, c_int, c_void_p]),
    'ph_bn_relu_bwd_grouped': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_bn_relu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'ph_tokens_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    'ph_tokens_finalize_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p]),
    'ph_gather_taps': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ph_scatter_taps': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ph_embed_fwd': (c_int, [C.POINTER(EmbedFwdArgs), c_void_p]),
    'ph_embed_bwd': (c_int, [C.POINTER(EmbedBwdArgs), c_void_p]),
    'ph_softmax_gather_bf16': (c_int, [c_void_p, c_i64, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'ph_ce_fwd': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ph_ce_bwd': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'ph_adamw': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_float, c_float, c_float,
                         c_float, c_float, c_int, c_void_p]),
    'ph_adamw_keep': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_float, c_float, c_float,
                              c_float, c_float, c_int, c_void_p, c_void_p]),
    'ph_cast_f32_to_bf16': (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    'ph_cast_bf16_to_f32': (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    'ph_scale_cast_f32_to_bf16': (c_int, [c_void_p, c_void_p, c_i64, c_float, c_void_p]),
    'ph_colsum_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'ph_colsum_grouped_bf16': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_layernorm_bwd_blocks': (c_int, [c_int]),
    'ph_ln_param_reduce_grouped': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_conv_weight_to_shadow_grouped': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_conv_grad_from_shadow_grouped': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_gemm_grouped_bf16': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_gemm_grouped_capped_bf16': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'ph_gemm_tuning': (c_int, [c_int, c_int]),
    'ph_attention_tuning': (c_int, [c_int]),
    'ph_layernorm_tuning': (c_int, [c_int]),
    'ph_gemm_flush_deferred': (c_int, [c_void_p]),
    'ph_gemm_dispatch_counts': (c_int, [c_void_p, c_int, c_int]),
    'ph_query_workspace': (c_i64, [c_int, c_void_p, c_int]),
    'ph_conv_dgrad_shadow_grouped': (c_int, [c_void_p, c_int, c_void_p]),
    'ph_add_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    'ph_act_bwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    'ph_copy_rows_bf16': (c_int, [c_void_p, c_int, RowMap, c_void_p, c_int, RowMap, c_int, c_int, c_int, c_void_p]),
    'ph_conv_weight_to_shadow': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_conv_grad_from_shadow': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ph_advance_seed': (c_int, [c_void_p, c_void_p]),
    'ph_store_words': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'ph_add_i64': (c_int, [c_void_p, c_int, c_i64, c_void_p]),
    'ph_fill_zero': (c_int, [c_void_p, c_i64, c_void_p]),
    'ph_copy_bytes': (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    'ph_weighted_sum_f32': (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    'ph_prof_enable': (c_int, [c_int]),
    'ph_prof_collect': (c_int, [c_void_p]),
    'ph_prof_dump': (c_int, [C.c_char_p]),
    'ph_probe_layouts': (c_int, [c_void_p, c_void_p, c_void_p]),
}

EXPORTS = tuple(_SIGS)


def _load():
    path = os.environ.get('PRISMER_HIP_LIB') or _build.LIB
    if not os.path.isfile(path):
        raise ImportError(f'libprismer_hip.so not found at {path}: run `python -m prismer_amd.build` (hipcc, gfx950). '
                          'There is no fallback path.')
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f'{path} does not export {name} (stale build?)') from e
        fn.restype = res
        fn.argtypes = args
    return lib, path


lib, LIB_PATH = _load()
GEMM_BIG_DEFAULT = (7, 128)    # ph_gemm_tuning(mode, min_tiles) values of the product dispatch (tests / probes restore them after an override)
ABI_VERSION = 105          # PH_VERSION of include/prismer_hip.h these ctypes structures mirror
if lib.ph_version() != ABI_VERSION:
    raise ImportError(f'{LIB_PATH} reports ABI revision {lib.ph_version()}, this binding was written for {ABI_VERSION} '
                      '(stale build? run `python -m prismer_amd.build --force`)')


class PrismerHipError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != PH_OK:
        raise PrismerHipError(f'{what} failed ({rc}): {lib.ph_last_error().decode()}')


def ptr(t):
    """raw device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
