"""ctypes binding of libdmet_hip.so (the C ABI declared in include/dmet.h).

There is NO CPU fallback: if the shared library is missing or no ROCm device tensor is supplied the ops raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# DMET_HIP_LIB lets a developer A/B an experimental build of the same ABI (tools/); the default is the in-tree .so
LIB_PATH = os.environ.get("DMET_HIP_LIB") or os.path.join(_PKG_DIR, "libdmet_hip.so")

_lock = threading.Lock()
_lib = None

_vp, _i, _i64, _sz, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); must list every symbol declared in include/dmet.h
SIGNATURES = {
    "dmet_version": (_i, []),
    "dmet_last_error": (C.c_char_p, []),
    "dmet_device_available": (_i, []),
    "dmet_knn_workspace_bytes": (_sz, [_i64, _i, _i, _i]),
    "dmet_knn_f32": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dmet_knn_local_f32": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_adamw_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _vp]),
    "dmet_adamw_lr_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _d, _d, _d, _d, _vp]),
    "dmet_bn_knn_local_dense_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _i,
                                         _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_knn_local_dense_f32": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_knn_fallback_stats": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "dmet_knn_size_hint": (_i, [_i, _i]),
    "dmet_finalize_defer_begin": (_i, []),
    "dmet_finalize_pending": (_i, []),
    "dmet_finalize_flush": (_i, [_vp]),
    "dmet_radius_f32": (_i, [_vp, _vp, _i, _i64, _i, _f, _i, _i, _vp, _vp, _vp]),
    "dmet_radius_counted_f32": (_i, [_vp, _vp, _i, _i64, _i, _f, _i, _i, _vp, _vp, _vp]),
    "dmet_radius_workspace_bytes": (_sz, [_i64]),
    "dmet_radius_windowed_f32": (_i, [_vp, _vp, _i, _i64, _i, _f, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "dmet_radius_windowed_local_f32": (_i, [_vp, _vp, _i, _i64, _i, _f, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "dmet_edgeconv_linear_workspace_bytes": (_sz, [_i64, _i]),
    "dmet_edgeconv_linear_max_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_edgeconv_fused_lds_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmet_node_linear_split_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmet_node_linear_split_sliced_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmet_bn_node_linear_split_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "dmet_gather_max_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_counted_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_lds_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_counted_lds_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_lds_sliced_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_lds_sliced_cap_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _i64, _vp]),
    "dmet_gather_max_lds16_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_edge_mlp2_supported": (_i, [_i, _i, _i, _i]),
    "dmet_edge_mlp2_bf16": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "dmet_edge_mlp2_bn_workspace_bytes": (_sz, [_i64, _i]),
    "dmet_edge_mlp2_bn_bf16": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp,
                                    _i, _vp, _vp, _sz, _vp]),
    "dmet_table_order_by_count": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "dmet_gather_max_counted_lds_j16_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_local_j16_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_bwd_j16_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp]),
    "dmet_gather_max_mixed_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_node_linear_split_bf16": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmet_gather_max_bf16q": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "dmet_gather_max_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp]),
    "dmet_reverse_index_workspace_bytes": (_sz, [_i64, _i64]),
    "dmet_reverse_index": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dmet_edge_features_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_segment_max_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "dmet_segment_sum_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_segment_max_bwd_f32": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_segment_sum_bwd_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_edge_features_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_met_reduce_f32": (_i, [_vp, _vp, _i64, _vp, _i, _vp, _vp]),
    "dmet_met_reduce_bwd_f32": (_i, [_vp, _vp, _i64, _vp, _i, _i64, _vp, _vp]),
    "dmet_met_loss_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "dmet_met_loss_strided_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "dmet_met_reduce_bwd_scaled_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i64, _vp, _vp]),
    "dmet_segment_sum_1d_f32": (_i, [_vp, _vp, _i, _vp, _vp]),
    "dmet_table_degree": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "dmet_table_edges": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dmet_batch_to_ptr": (_i, [_vp, _i64, _i, _vp, _vp]),
    "dmet_xty_workspace_bytes": (_sz, [_i64, _i, _i]),
    "dmet_xty_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "dmet_onehot_xty_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _sz, _vp]),
    "dmet_gather_max_bwd_lds_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "dmet_gather_max_bwd_lds16_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "dmet_gather_max_bwd_lds16_cap_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _i64, _vp]),
    "dmet_gather_max_bwd_j16_cap_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _i64, _vp]),
    "dmet_gather_max_bwd_sliced_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _i64, _vp]),
    "dmet_gather_max_bwd_j16_sliced_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _i64, _vp]),
    "dmet_edgeconv_linear_bwd_sliced_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_edgeconv_linear_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "dmet_edgeconv_linear_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_edgeconv_linear_bwd_add_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_edgeconv_linear_bwd_add_j16_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_head_fwd_f32": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmet_head_bwd_workspace_bytes": (_sz, [_i64]),
    "dmet_head_bwd_f32": (_i, [_vp, _i64] + [_vp] * 10 + [_vp, _sz, _vp]),
    "dmet_bn_workspace_bytes": (_sz, [_i64, _i]),
    "dmet_bn_fwd_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_bn_fwd_tracked_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_bn_head_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmet_bn_bwd_stats_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_encode_bn_bwd_f32": (_i, [_vp, _i64, _vp, _i64] + [_vp] * 9 + [_vp, _vp] + [_vp] * 5 + [_vp] * 9 + [_vp, _vp, _sz, _vp]),
    "dmet_bn_eval_stats_f32": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp]),
    "dmet_bn_apply_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dmet_bn_stats_f32": (_i, [_vp, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_bn_bwd_f32": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dmet_encode_fwd_f32": (_i, [_vp, _i64, _vp, _i64] + [_vp] * 9 + [_vp, _vp]),
    "dmet_encode_bwd_workspace_bytes": (_sz, [_i64]),
    "dmet_encode_bwd_f32": (_i, [_vp, _i64, _vp, _i64] + [_vp] * 9 + [_vp, _vp] + [_vp] * 9 + [_vp, _sz, _vp]),
}


class DmetLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libdmet_hip.so (built by `python -m deepmetv2_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DmetLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m deepmetv2_amd.build` "
                "(needs hipcc; gfx950). deepmetv2_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here means header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dmet_last_error()
        raise RuntimeError(f"{what or 'dmet call'} failed (rc={rc}): {msg.decode(errors='replace') if msg else ''}")
