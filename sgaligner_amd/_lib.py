"""ctypes binding of the C-ABI HIP library (include/sgaligner_hip.h).

There is NO fallback: if csrc/libsga_hip.so is missing or fails to load, every product op raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SGA_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libsga_hip.so')   # override: kernel experiments
_lib = None

P = c_void_p
I = c_int
F = c_float

# name -> (restype, argtypes).  Mirrors include/sgaligner_hip.h one to one.
SIGNATURES = {
    'sga_version': (I, []),
    'sga_last_error': (c_char_p, []),
    'sga_device_cus': (I, []),
    'sga_pointnet_fwd': (I, [P, P, P, P, P, P, P, P, P, I, I, I, P]),
    'sga_pointnet_fwd_ws_bytes': (c_size_t, [I, I]),
    'sga_loss_neg_grad_wide_floats': (c_size_t, [I, I, I]),
    'sga_loss_neg_grad_wide': (I, [P, I, I, I, I, c_float, c_float, P, P, P, c_size_t, P]),
    'sga_wide16_ldt': (c_long, [I, I, I]),
    'sga_wide16_prepare': (I, [P, I, I, I, I, P, P, P]),
    'sga_loss_neg_sums_f16': (I, [P, I, I, I, I, c_float, c_float, P, I, I, P]),
    'sga_loss_neg_grad_f16_bytes': (c_size_t, [I, I, I]),
    'sga_loss_neg_grad_f16': (I, [P, P, I, I, I, I, c_float, c_float, P, P, P, c_size_t, I, I, P]),
    'sga_loss_head_fwd': (I, [P, I, P, P, I, c_double, c_double, c_double, c_double, P, P]),
    'sga_loss_head_bwd': (I, [P, P, I, P, P, I, c_double, c_double, c_double, c_double, P, P, P, P]),
    'sga_pointnet_fwd_ws': (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, c_size_t, I, P]),
    'sga_pointnet_fwd_bn_ws_bytes': (c_size_t, [I, I]),
    'sga_pointnet_fwd_bn': (I, [P, P, P, P, P, P, P, P, P, I, I, I, P, c_size_t, P, c_size_t, P, I, P]),
    'sga_pointnet_bwd': (I, [P] * 15 + [I, I, I, I, P]),
    'sga_gat_complete_flags': (I, [P, P, P, I, P, P]),
    'sga_gat_attn_fwd': (I, [P, P, P, P, P, P, P, I, I, P, P, P, P]),
    'sga_gat_attn_bwd': (I, [P, P, P, P, P, P, P, I, I, P, P, P, P, P]),
    'sga_elu_fwd': (I, [P, P, c_size_t, P]),
    'sga_elu_bwd': (I, [P, P, P, c_size_t, P]),
    'sga_simrank_workspace_bytes': (c_size_t, [I]),
    'sga_simrank_workspace_bytes_f16': (c_size_t, [I, I]),
    'sga_simrank': (I, [P, I, I, P, P, P, I, I, I, P, P, I, I, P, P, P, I, P, c_size_t, P]),
    'sga_pair_metrics': (I, [P, P, P, I, P, P, P, I, P, P]),
    'sga_gemm_ex': (I, [I, I, I, I, I, P, c_long, P, c_long, P, c_long, P, I, P, c_long, P]),
    'sga_pct_attention': (I, [P, c_long, P, c_long, I, I, P, P, c_long, P]),
    'sga_pct_attention_bwd': (I, [P, c_long, P, c_long, P, c_long, I, I, P, P, P, c_long, P, c_long, P]),
    'sga_segment_max': (I, [P, c_long, I, I, I, P, P, P]),
    'sga_segment_max_bwd': (I, [P, P, I, I, I, P, c_long, P]),
    'sga_pct_head_prep': (I, [P, P, P, P, P, I, I, c_long, I, c_float, P, P, P]),
    'sga_pct_head_dw': (I, [P, P, P, P, P, P, P, c_long, I, I, I, I, P, P, P, P]),
    'sga_pct_head_scatter': (I, [P, P, P, I, I, I, I, P, c_long, P]),
    'sga_segment_max_affine': (I, [P, c_long, I, I, I, P, P, c_float, P, P, P]),
    'sga_bn_stats': (I, [P, c_long, I, I, P, P]),
    'sga_bn_finalize': (I, [P, I, I, P, P, P, P, P, c_float, c_float, I, P, P]),
    'sga_bn_bwd_finalize': (I, [P, I, I, P, P]),
    'sga_bn_apply': (I, [P, c_long, I, I, P, P, I, P, c_long, P, c_long, P]),
    'sga_bn_bwd_stats': (I, [P, c_long, P, c_long, I, I, P, P, P, P, I, P, P]),
    'sga_bn_bwd_apply': (I, [P, c_long, P, c_long, I, I, P, P, P, P, P, P, I, P, c_long, P]),
    'sga_fps_scratch_floats': (c_size_t, [I]),
    'sga_fps': (I, [P, P, I, P, I, P, I, P, I, P, I, P, P, P]),
    'sga_hull_candidates': (I, [P, P, I, P, P, P]),
    'sga_hull_max_candidates': (I, []),
    'sga_hull_vertices': (I, [P, P, I, P, P, P]),
    'sga_gemm': (I, [I, I, I, I, I, P, c_long, I, P, c_long, P, c_long, P, I, P]),
    'sga_colsum': (I, [P, c_long, I, I, P, I, P]),
    'sga_gemm_bnstats': (I, [I, I, I, P, c_long, P, c_long, P, c_long, P, P, P]),
    'sga_cast_f64_f32': (I, [P, P, c_size_t, P]),
    'sga_loss_gather': (I, [P, I, I, P, I, P, I, P, P]),
    'sga_loss_scatter': (I, [P, P, P, P, I, I, I, P, P]),
    'sga_loss_neg_sums': (I, [P, I, I, I, I, F, F, P, P]),
    'sga_loss_neg_grad': (I, [P, I, I, I, I, F, F, P, P, P]),
    'sga_loss_neg_sums_shard': (I, [P, I, I, I, I, F, F, P, I, I, P]),
    'sga_loss_neg_grad_shard': (I, [P, I, I, I, I, F, F, P, P, I, I, P]),
    'sga_loss_anchor_fwd': (I, [P, P, I, I, P, F, F, F, P, I, I, P]),
    'sga_loss_anchor_bwd': (I, [P, P, I, I, P, F, F, F, P, P, P, I, I, P]),
    'sga_loss_anchor_f16_ws_bytes': (c_size_t, [I, I, I]),
    'sga_loss_anchor_fwd_f16': (I, [P, P, P, I, I, P, F, F, F, P, I, I, P, c_size_t, P]),
    'sga_loss_anchor_bwd_f16': (I, [P, P, P, I, I, P, F, F, F, P, P, P, I, I, P, c_size_t, P]),
    'sga_loss_stash_grad_f16_bytes': (c_size_t, [I, I]),
    'sga_loss_stash_grad_f16': (I, [P, P, I, I, I, I, P, I, I, P, c_size_t, P]),
    'sga_loss_multi_sums': (I, [P, I, I, P, I, I, I, F, F, P, I, I, P]),
    'sga_loss_multi_grad': (I, [P, I, I, P, I, I, I, F, F, P, P, P, I, I, P]),
    'sga_loss_centre_bytes': (c_size_t, []),
    'sga_loss_centre_tables': (I, [P, I, I, I, P, P, P]),
    'sga_loss_multi_sums_centred': (I, [P, I, P, I, I, I, F, F, P, I, I, P]),
    'sga_loss_multi_grad_centred': (I, [P, I, P, I, I, I, F, F, P, P, P, I, I, P]),
    'sga_loss_scatter_tangent_stat': (I, [P, P, P, P, I, I, P, P, P]),
    'sga_loss_build_joint': (I, [P, I, P, I, P, P]),
    'sga_loss_fold_joint': (I, [P, I, P, P, I, P, P, P]),
    'sga_loss_check_norms': (I, [P, I, P, P]),
    'sga_loss_slots': (I, []),
    'sga_loss_anchor_multi_fwd': (I, [P, I, P, I, P, F, F, F, P, I, I, P]),
    'sga_loss_anchor_multi_bwd': (I, [P, I, P, I, P, F, F, F, P, P, P, P, I, I, P, P]),
    'sga_loss_stash_grad': (I, [P, P, I, I, P, I, I, P]),
    'sga_loss_anchor_multi_bwd_sym': (I, [P, I, P, I, P, F, F, F, P, P, P, P, P, I, I, P, P]),
    'sga_loss_stash_grad_sym': (I, [P, P, P, I, I, P, I, I, P]),
    'sga_loss_anchor_multi_bwd_symx': (I, [P, I, P, I, P, F, F, F, P, P, P, P, P, I, I, I, I, I, P, P]),
    'sga_loss_stash_grad_symx': (I, [P, P, P, I, I, P, I, I, I, I, I, P]),
    'sga_loss_split3_bytes': (c_size_t, [I, I, I]),
    'sga_loss_split3_tables': (I, [P, I, I, I, P, P, P]),
    'sga_loss_scatter_tangent': (I, [P, P, P, P, I, I, I, I, P, P, P]),
    'sga_loss_stash_grad_symx_bf16x6': (I, [P, P, P, I, I, I, P, I, I, I, I, I, P]),
    'sga_loss_multi_sums_bf16x6': (I, [P, I, P, I, I, I, F, F, P, I, I, I, P]),
    'sga_loss_multi_grad_bf16x6': (I, [P, I, P, I, I, I, F, F, P, P, P, I, I, P]),
    'sga_group_loss_fwd': (I, [P, I, P, I, I, P, I, P, c_int64, F, F, F, P, P, P, I, P]),
    'sga_group_loss_bwd': (I, [P, I, P, I, I, P, I, P, c_int64, F, F, F, P, P, P, P, P, I, P]),
    'sga_fusion_fwd': (I, [P, I, P, P, I, I, P]),
    'sga_fusion_bwd_workspace_bytes': (c_size_t, [I]),
    'sga_fusion_bwd': (I, [P, I, P, P, P, P, I, I, P, c_size_t, P]),
}


class SgaLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgaLibraryError(
                f'HIP library not built: {LIB_PATH} is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `python -m sgaligner_amd._build`). sgaligner_amd has no CPU fallback.')
        # torch first: its bundled HIP runtime must be the one this library binds to.  Loaded the other way round (e.g. build() and then a
        # training step in one process) the library gets /opt/rocm's libamdhip64 as a SECOND runtime instance, whose memory calls
        # (hipMemsetAsync on a torch allocation) fail with "no ROCm-capable device is detected" while kernel launches still work.
        import torch  # noqa: F401
        try:
            l = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise SgaLibraryError(f'cannot load {LIB_PATH}: {e}') from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise SgaLibraryError(f'{LIB_PATH} does not export {name}; rebuild the library') from e
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().sga_last_error()
        raise RuntimeError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')
