"""ctypes binding of libamc3d_hip.so (C-ABI: include/amc3d.h).

There is NO fallback: if the shared library is missing or a symbol is absent,
importing the operators raises.  The product path never routes through oracle/.
"""
import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.environ.get("AMC3D_LIB") or os.path.join(_CSRC, "libamc3d_hip.so")  # AMC3D_LIB: diagnostic builds

_vp, _i, _f, _sz, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long
_ll = ctypes.c_longlong

# name -> (restype, argtypes); must list every symbol include/amc3d.h declares
SIGNATURES = {
    "amc3d_version": (ctypes.c_char_p, []),
    "amc3d_last_error": (ctypes.c_char_p, []),
    "amc3d_stream_create_dedicated": (_i, [ctypes.POINTER(ctypes.c_void_p)]),
    "amc3d_stream_create_masked": (_i, [_vp, _i, _i]),
    "amc3d_stream_create_cu_mask": (_i, [_vp, _vp, _i]),
    "amc3d_probe_xcc_ids": (_i, [_i, _vp, _vp]),
    "amc3d_reserve_scratch": (_i, [_i, _vp, _vp]),
    "amc3d_stream_destroy": (_i, [_vp]),
    "amc3d_grid_search_workspace_bytes": (_sz, [_i, _i, _i]),
    "amc3d_ball_query": (_i, [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_group_points": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_scatter_workspace_bytes": (_sz, [_i, _i, _i]),
    "amc3d_group_points_grad": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_gather_points": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_gather_points_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_fps_workspace_bytes": (_sz, [_i, _i]),
    "amc3d_furthest_point_sampling": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_three_nn": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_three_interpolate": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_three_interpolate_add": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_three_interpolate_grad": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_knnquery_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amc3d_knnquery_uses_grid": (_i, [_i, _i, _i, _i]),
    "amc3d_knnquery": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "amc3d_vote_labels": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_posmask": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_ambiguity_workspace_bytes": (_sz, [_i]),
    "amc3d_ambiguity": (_i, [_i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_contrast_csr_workspace_bytes": (_sz, [_i]),
    "amc3d_contrast_csr": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_contrast_backward_csr_supported": (_i, [_i]),
    "amc3d_contrast_backward_csr": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp,
                                         _vp, _vp]),
    "amc3d_contrast_mutual_workspace_bytes": (_sz, [_i]),
    "amc3d_contrast_mutual": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_contrast_backward_mutual_workspace_bytes": (_sz, [_i]),
    "amc3d_contrast_backward_mutual": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz,
                                            _vp, _vp]),
    "amc3d_adamw_chunk": (_i, []),
    "amc3d_adamw_step": (_i, [_vp, _vp, _i, ctypes.c_double, ctypes.c_double, _f, _f, _vp, _vp, _vp]),
    "amc3d_select_anchors_ints": (_sz, [_i]),
    "amc3d_select_anchors": (_i, [_i, _vp, _vp, _sz, _vp]),
    "amc3d_contrast_forward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_contrast_forward_cm": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_contrast_backward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_grouped_conv_supported": (_i, [_i, _i]),
    "amc3d_transpose_cn": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "amc3d_grouped_conv_forward": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_grouped_conv_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "amc3d_grouped_conv_backward": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_confusion_update": (_i, [_i, _i, _l, _vp, _vp, _ll, _i, _vp, _vp, _vp]),
    "amc3d_cross_entropy_workspace_bytes": (_sz, [_i, _l]),
    "amc3d_cross_entropy_forward": (_i, [_i, _i, _l, _vp, _vp, _ll, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_cross_entropy_backward": (_i, [_i, _i, _l, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_pointwise_conv_forward": (_i, [_i, _i, _i, _l, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_pointwise_conv_forward_workspace_bytes": (_sz, [_i, _i, _i, _l, _i]),
    "amc3d_pointwise_conv_forward_ws": (_i, [_i, _i, _i, _l, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_pointwise_conv_workspace_bytes": (_sz, [_i, _i, _i, _l]),
    "amc3d_pointwise_conv_backward": (_i, [_i, _i, _i, _l, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_residual_forward": (_i, [_i, _i, _l, _f, _f] + [_vp] * 11 + [_vp, _sz, _vp]),
    "amc3d_bn_residual_backward": (_i, [_i, _i, _l] + [_vp] * 11 + [_vp, _sz, _vp]),
    "amc3d_split_columns": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_join_columns": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_masked_refine_workspace_ints": (_sz, [_i]),
    "amc3d_masked_refine_forward": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_masked_refine_backward": (_i, [_i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_bn_sigmoid_forward": (_i, [_i, _i, _l, _f, _f] + [_vp] * 10 + [_vp, _sz, _vp]),
    "amc3d_bn_sigmoid_backward": (_i, [_i, _i, _l] + [_vp] * 10 + [_vp, _sz, _vp]),
    "amc3d_bias_grad": (_i, [_i, _i, _l, _vp, _vp, _vp]),
    "amc3d_sa_residual_forward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 8),
    "amc3d_sa_residual_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amc3d_sa_residual_backward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 11 + [_sz, _vp]),
    "amc3d_index_duplicates_workspace_bytes": (_sz, [_i, _i]),
    "amc3d_index_duplicates": (_i, [_i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_augment_workspace_bytes": (_sz, [_i]),
    "amc3d_augment_clouds": (_i, [_i, _i, _i, _f, _f] + [_vp] * 10 + [_sz, _vp]),
    "amc3d_voxelize_workspace_bytes": (_sz, [_i]),
    "amc3d_voxelize": (_i, [_i, _vp, ctypes.c_double] + [_vp] * 7 + [_sz, _vp]),
    "amc3d_voxel_select": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_crop_nearest_workspace_bytes": (_sz, [_i]),
    "amc3d_crop_nearest": (_i, [_i, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_local_aggregation_supported": (_i, [_i, _i]),
    "amc3d_group_moments_bytes": (_sz, [_i, _i]),
    "amc3d_group_moments": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_local_aggregation_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amc3d_local_aggregation_forward": (_i, [_i] * 7 + [_f, _f] + [_vp] * 18 + [_i, _vp, _vp, _sz, _vp]),
    "amc3d_local_aggregation_backward": (_i, [_i] * 6 + [_vp] * 17 + [_i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_pointwise_conv_forward_bf16": (_i, [_i, _i, _i, _l, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_pointwise_conv_workspace_bytes_bf16": (_sz, [_i, _i, _i, _l]),
    "amc3d_pointwise_conv_backward_bf16": (_i, [_i, _i, _i, _l, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_grouped_conv_bn_supported": (_i, [_i, _i]),
    "amc3d_grouped_conv_bn_forward": (_i, [_i] * 7 + [_f, _f] + [_vp] * 16 + [_i, _vp, _vp, _sz, _vp]),
    "amc3d_grouped_conv_bn_backward": (_i, [_i] * 6 + [_vp] * 15 + [_i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_group_csr_workspace_bytes": (_sz, [_i, _i, _i]),
    "amc3d_group_csr": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_group_moments_csr": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_grouped_conv_bn_csr_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "amc3d_grouped_conv_bn_backward_csr": (_i, [_i] * 6 + [_vp, _i] + [_vp] * 16 + [_i, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_group_csr_dp": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "amc3d_sa_tail_supported": (_i, [_i, _i, _i]),
    "amc3d_sa_tail_pays": (_i, [_i, _i]),
    "amc3d_sa_tail_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "amc3d_sa_tail_forward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 8 + [_f, _f, _i] + [_vp] * 10 + [_sz, _vp]),
    "amc3d_sa_tail_backward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 10 + [_i, _vp, _vp, _vp, _vp, _i] + [_vp] * 5 + [_sz, _vp]),
    "amc3d_bn_workspace_bytes": (_sz, [_i]),
    "amc3d_bn_stats": (_i, [_i, _i, _l, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_forward": (_i, [_i, _i, _l, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_update_running": (_i, [_i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_bn_act": (_i, [_i, _i, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_bn_max": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_bn_backward": (_i, [_i, _i, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_sums": (_i, [_i, _i, _l, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_forward_synced": (_i, [_i, _i, _l, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amc3d_bn_backward_sums": (_i, [_i, _i, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "amc3d_bn_backward_synced": (_i, [_i, _i, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def build(force=False):
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", _CSRC, "-j8"] + (["-B"] if force else [])
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building libamc3d_hip.so failed:\n" + proc.stdout)
    return _SO


_lib = None


def load():
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {_CSRC}`); there is no CPU fallback")
    # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64; import it first so this library binds
    # to the SAME runtime instance (loading /opt/rocm's copy first leaves two HIP runtimes in one process
    # and the kernels here then see "no ROCm-capable device")
    import torch  # noqa: F401
    lib = ctypes.CDLL(_SO)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed (hipError {status}): {load().amc3d_last_error().decode()}")
