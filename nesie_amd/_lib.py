"""ctypes binding of ``include/nesie_ops.h`` (libnesie_hip.so).

This is the binding a maintainer of the reference would write in place of its
pybind11 shims: plain pointers, ints and a stream handle.
The library is looked up in-tree only (``nesie_amd/libnesie_hip.so``) and a
missing or unloadable library is an ImportError -- never a silent fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NESIE_LIB") or os.path.join(_HERE, "libnesie_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float

# name -> argtypes, in the order of include/nesie_ops.h (stream last)
SIGNATURES = {
    "nesie_furthest_point_sampling_wrapper": [_I, _I, _I, _P, _P, _P, _P],
    "nesie_furthest_point_sampling_ws": [_I, _I, _I, _P, _P, _P, _P, ctypes.c_size_t, _P],
    "nesie_furthest_point_sampling_with_dist_wrapper": [_I, _I, _I, _P, _P, _P, _P],
    "nesie_ball_query_wrapper": [_I, _I, _I, _F, _F, _I, _P, _P, _P, _P],
    "nesie_ball_query_indexed": [_I, _I, _I, _F, _F, _I, _P, _P, ctypes.c_size_t, _P, _P],
    "nesie_group_points_forward": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nesie_group_points_backward": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nesie_gather_points_wrapper": [_I, _I, _I, _I, _P, _P, _P, _P],
    "nesie_gather_points_grad_wrapper": [_I, _I, _I, _I, _P, _P, _P, _P],
    "nesie_three_nn_wrapper": [_I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_three_interpolate_wrapper": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_three_interpolate_grad_wrapper": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_blend_conv_bn_forward": [_I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                                    _F, _P, _P, _P, _P, _P, ctypes.c_size_t, _I, _I, _P],
    "nesie_blend_conv_bn_backward": [_I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, _P, _P, _P, ctypes.c_size_t, _I, _I, _P],
    "nesie_side_decode_forward": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "nesie_side_decode_backward": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nesie_bn_eval_coef": [_I, _P, _P, _P, _P, _F, _P, _P],
    "nesie_affine_relu_forward": [_I, _I, ctypes.c_longlong, _P, _P, _I, _P, _I, _P, _P],
    "nesie_affine_relu_maxpool_forward": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_aligned_3d_nms": [_I, _I, _P, _P, _P, _P, _F, _P, _P, _P],
    "nesie_points_in_boxes_count": [_I, _I, _I, _P, _P, _P, _P],
    "nesie_boxes_overlap_bev": [_I, _P, _I, _P, _P, _P],
    "nesie_scene_assemble": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _P],
    "nesie_grid_taps": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nesie_blend_conv_forward": [_I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I,
                                 _P, _P],
    "nesie_blend_conv_backward": [_I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    "nesie_blend_conv_backward_bn": [_I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    "nesie_blend_conv_backward_staged": [_I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P,
                                         ctypes.c_size_t, _P],
    "nesie_pw_bnb_coef": [_I, _I, ctypes.c_double, _P, _P, _P, _P, _P, _P, _P],
    "nesie_sort_vertices_forward": [_I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_points_in_boxes_batch": [_I, _I, _I, _P, _P, _P, _P],
    "nesie_vote_targets": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "nesie_group_max_pool_forward": [ctypes.c_longlong, _I, _P, _P, _P, _P],
    "nesie_group_max_pool_backward": [ctypes.c_longlong, _I, _P, _P, _P, _P],
    "nesie_query_and_group_forward": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P],
    "nesie_query_and_group_backward": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nesie_three_interpolate_grad_csr": [_I, _I, _I, _I, _P, ctypes.c_longlong, _P, _P, _P, _P, _P],
    "nesie_inverted_index": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P],
    "nesie_query_and_group_backward_csr": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_group_points_backward_csr": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nesie_gather_rows3": [_I, _I, _I, _P, _P, _P, _P],
    "nesie_vote_finish_forward": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "nesie_vote_finish_backward": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "nesie_query_and_group_backward_xyz": [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P],
    "nesie_group_max_pool_backward_add": [ctypes.c_longlong, _I, _P, _P, _P, _P],
    "nesie_channel_sum": [_I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P],
    "nesie_iou3d_forward": [_I, _P, _P, _P, _P, _P],
    "nesie_lhs_nms_samecls": [_I, _I, _P, _F, _P, _P],
    "nesie_bn_relu_maxpool_forward": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P,
                                      _P, _P, ctypes.c_size_t, _P],
    "nesie_bn_relu_maxpool_backward": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _P, ctypes.c_size_t, _P],
    "nesie_conv_wgrad": [_I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, ctypes.c_longlong,
                         _P, _I, _P, _P, ctypes.c_size_t, _P],
    "nesie_conv_wgrad_bn": [_I, _I, _I, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _P, _P,
                            ctypes.c_longlong, _P, _I, _P, _P, ctypes.c_size_t, _P],
    "nesie_mlp_layer_forward_stream": [_I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P,
                                       _I, _P, _P, _P],
    "nesie_mlp_stat_finalize": [_I, ctypes.c_longlong, ctypes.c_double, _P, _P, _P, _P, _P, _F,
                                _F, _P, _I, _P],
    "nesie_pw_layer_forward": [_I, _I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                               ctypes.c_longlong, _I, _I, _P, _I, _P, _I, _P, _P, ctypes.c_longlong,
                               _P, _I, _I, _P, _P, _P, _P, _P],
    "nesie_pw_dgrad_bn_reduce": [_I, _I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                                 ctypes.c_longlong, _I, _I, _P, ctypes.c_longlong, _P,
                                 ctypes.c_longlong, _P, _P, _P],
    "nesie_bn_relu_backward_apply": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _I, _P,
                                     _P, _P, _I, _P, _P],
    "nesie_head_targets": [_I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nesie_head_loss_forward": [_I, _I, _I, _I] + [_P] * 30,
    "nesie_head_loss_forward_unsup": [_I, _I, _I, _I] + [_P] * 7 + [_I] + [_P] * 23,
    "nesie_head_loss_forward_sigma": [_I, _I, _I, _I, _I] + [_P] * 30,
    "nesie_saqe_extra_loss_forward": [_I, _I, _I, _I] + [_P] * 21,
    "nesie_saqe_extra_loss_backward": [_I, _I, _I] + [_P] * 11,
    "nesie_head_loss_backward": [_I, _I, _I] + [_P] * 18,
    "nesie_vote_loss_forward": [_I, _I, ctypes.c_longlong, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                _P, _P],
    "nesie_vote_loss_backward": [ctypes.c_longlong, _P, _P, _P, _P, _P],
    "nesie_proposal_jitter": [_I, _I, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P],
    "nesie_side_prob_stats": [_I, _I, _I, _I, _P, _P, _P],
    "nesie_flat_adamw_step": [ctypes.c_longlong, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _P,
                              ctypes.c_size_t, _P],
    "nesie_flat_adamw_step_dev": [ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P,
                                  ctypes.c_size_t, _P],
    "nesie_pw_wgrad": [_I, _I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                       ctypes.c_longlong, _P, _I, _P, _P, ctypes.c_size_t, _P],
    "nesie_pw_wgrad_deferred": [_I, _I, _I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P,
                       ctypes.c_longlong, _P, _I, _P, _P, ctypes.c_size_t, _P],
    "nesie_pw_wgrad_flush_deferred": [_P],
    "nesie_pw_wgrad_drop_deferred": [],
    "nesie_pw_wgrad_bn_backward": [_I, _I, _I, _I, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _P, _P,
                                   _P, _I, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P,
                                   ctypes.c_size_t, _P],
    "nesie_pw_wgrad_bn_backward_deferred": [_I, _I, _I, _I, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _P, _P,
                                   _P, _I, _P, ctypes.c_longlong, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P,
                                   ctypes.c_size_t, _P],
    "nesie_pw_layer_forward_k4": [_I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _I, _I, _P, _P,
                                  ctypes.c_longlong, _P, _P],
    "nesie_pw_dgrad_bn_reduce_k4": [_I, _I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _I, _I, _P,
                                    ctypes.c_longlong, _P, _P, _P, _P, _P],
    "nesie_pw_wgrad_bn_backward_k4": [_I, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _P, _P, _P, _I, _P,
                                      ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _I, _P],
    "nesie_pw_wgrad_bn_backward_k4_fused": [_I, ctypes.c_longlong, _P, _P, ctypes.c_longlong, _P, _P, _P, _I, _P,
                                            ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                            ctypes.c_size_t, _I, _P],
    "nesie_k4_moments": [_I, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P],
    "nesie_k4_stat_finalize": [ctypes.c_double, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P],
    "nesie_k4_first_layer_wgrad": [_P, _P, _P, _P, _I, _P, _P],
    "nesie_pw_stats_finalize": [_I, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P],
    "nesie_pw_pool_finish": [_I, _I, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P,
                             _P],
    "nesie_pw_pool_finish_z": [_I, _I, _I, ctypes.c_longlong, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P,
                               _P, _P],
    "nesie_pool_tail_sizes": [_I, _I, _I, ctypes.c_longlong, _P],
    "nesie_pool_tail_prepare": [_I, _I, _I, _I, _I] + [_P] * 14,
    "nesie_pool_tail_dgrad": [_I, _I, _I, ctypes.c_longlong, _I, _P, ctypes.c_longlong, _P, _P, _P, _P,
                              _P, ctypes.c_longlong, _P, _P],
    "nesie_pool_tail_wgrad": [_I, _I, _I, ctypes.c_longlong, _I, _P, ctypes.c_longlong] + [_P] * 10,
    "nesie_bn_relu_forward": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P,
                              _P, _P, _P, _I, _P, _I, _P, ctypes.c_size_t, _P],
    "nesie_bn_relu_backward": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P,
                               _P, _P, _P, _I, _P, _P, ctypes.c_size_t, _P],
}

_lib = None


def load():
    """Load libnesie_hip.so once and attach argtypes.  Raises ImportError."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C nesie_amd/csrc`). nesie_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _I
    lib.nesie_fps_workspace_bytes.argtypes = [_I, _I]
    lib.nesie_fps_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_fps_leaves_index.argtypes = [_I, _I]
    lib.nesie_fps_leaves_index.restype = _I
    lib.nesie_bn_workspace_bytes.argtypes = [_I, _I, ctypes.c_longlong]
    lib.nesie_bn_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_conv_wgrad_workspace_bytes.argtypes = [_I, _I, _I, ctypes.c_longlong]
    lib.nesie_conv_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_blend_conv_bn_workspace_bytes.argtypes = [_I, _I, _I, _I]
    lib.nesie_blend_conv_bn_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_blend_conv_backward_workspace_bytes.argtypes = [_I, _I, _I, _I]
    lib.nesie_blend_conv_backward_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_blend_conv_runs.argtypes = [_I, _I]
    lib.nesie_blend_conv_runs.restype = _I
    lib.nesie_mlp_stream_partials.argtypes = [_I, ctypes.c_longlong]
    lib.nesie_mlp_stream_partials.restype = ctypes.c_longlong
    lib.nesie_pw_wgrad_supported.argtypes = [_I, _I, ctypes.c_longlong]
    lib.nesie_pw_wgrad_supported.restype = _I
    lib.nesie_pw_wgrad_tiled.argtypes = [_I, _I, _I, _I, ctypes.c_longlong]
    lib.nesie_pw_wgrad_tiled.restype = _I
    lib.nesie_pw_wgrad_bn_supported.argtypes = [_I, _I, ctypes.c_longlong]
    lib.nesie_pw_wgrad_bn_supported.restype = _I
    lib.nesie_pw_wgrad_workspace_bytes.argtypes = [_I, _I, _I, _I, ctypes.c_longlong]
    lib.nesie_pw_wgrad_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_k4_moments_bytes.argtypes = []
    lib.nesie_k4_moments_bytes.restype = ctypes.c_size_t
    lib.nesie_pw_wgrad_bn_backward_k4_slots.argtypes = [_I, ctypes.c_longlong]
    lib.nesie_pw_wgrad_bn_backward_k4_slots.restype = _I
    lib.nesie_pw_wgrad_pending.argtypes = []
    lib.nesie_pw_wgrad_pending.restype = _I
    lib.nesie_flat_adamw_workspace_bytes.argtypes = []
    lib.nesie_flat_adamw_workspace_bytes.restype = ctypes.c_size_t
    lib.nesie_pw_supported.argtypes = [_I, _I, ctypes.c_longlong]
    lib.nesie_pw_supported.restype = _I
    lib.nesie_pw_stat_slots.argtypes = [_I, _I, _I, _I, ctypes.c_longlong]
    lib.nesie_pw_stat_slots.restype = _I
    lib.nesie_pool_tail_supported.argtypes = [_I, _I, ctypes.c_longlong, _I]
    lib.nesie_pool_tail_supported.restype = _I
    lib.nesie_abi_version.restype = _I
    lib.nesie_set_distance_form.argtypes = [_I]
    lib.nesie_set_distance_form.restype = _I
    lib.nesie_get_distance_form.argtypes = []
    lib.nesie_get_distance_form.restype = _I
    lib.nesie_set_cu_count.argtypes = [_I]
    lib.nesie_set_cu_count.restype = _I
    lib.nesie_get_cu_count.argtypes = []
    lib.nesie_get_cu_count.restype = _I
    lib.nesie_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def call(name, *args):
    """Call an entry point; non-zero status becomes a RuntimeError."""
    lib = load()
    status = getattr(lib, name)(*args)
    if status != 0:
        msg = lib.nesie_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (status {status}): {msg}")


def library_sha256():
    """sha256 of the library file ``load()`` maps (bench.py ties the PMC traffic figures under
    profiles/ to the build that produced them with it)."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, 'rb') as f:
        for chunk in iter(lambda: f.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()
