"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU)."""
import ctypes
import glob
import os
import re

from nesie_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(nesie_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_reference_entry_points():
    names = declared_symbols()
    for stem in ["furthest_point_sampling_wrapper", "furthest_point_sampling_with_dist_wrapper",
                 "ball_query_wrapper", "group_points_forward", "group_points_backward",
                 "gather_points_wrapper", "gather_points_grad_wrapper", "three_nn_wrapper",
                 "three_interpolate_wrapper", "three_interpolate_grad_wrapper",
                 "sort_vertices_forward", "points_in_boxes_batch"]:
        assert "nesie_" + stem in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_binding_covers_every_compute_entry_point():
    compute = [n for n in declared_symbols()
               if n not in ("nesie_abi_version", "nesie_last_error",
                            "nesie_set_distance_form", "nesie_get_distance_form", "nesie_set_cu_count", "nesie_get_cu_count",
                            "nesie_fps_workspace_bytes", "nesie_fps_leaves_index",
                            "nesie_bn_workspace_bytes",
                            "nesie_pw_supported", "nesie_pw_stat_slots", "nesie_pw_wgrad_supported", "nesie_pw_wgrad_tiled", "nesie_pool_tail_supported",
                            "nesie_pw_wgrad_workspace_bytes", "nesie_pw_wgrad_pending", "nesie_k4_moments_bytes", "nesie_pw_wgrad_bn_backward_k4_slots", "nesie_pw_wgrad_bn_supported", "nesie_blend_conv_runs", "nesie_blend_conv_backward_workspace_bytes",
                            "nesie_conv_wgrad_workspace_bytes", "nesie_mlp_stream_partials",
                            "nesie_blend_conv_bn_workspace_bytes",
                            "nesie_flat_adamw_workspace_bytes")]
    assert sorted(compute) == sorted(_lib.SIGNATURES)
    lib = _lib.load()
    assert lib.nesie_abi_version() >= 1
    assert isinstance(lib.nesie_last_error(), bytes)


def test_invalid_arguments_return_a_status_not_a_crash():
    lib = _lib.load()
    # negative size -> NESIE_ERR_INVALID_ARG, no launch attempted (safe without a GPU)
    st = lib.nesie_ball_query_wrapper(-1, 4, 4, 0.0, 1.0, 2, None, None, None, None)
    assert st == 1
    assert b"ball_query_wrapper" in lib.nesie_last_error()
    # empty problems succeed without touching the device
    assert lib.nesie_group_points_forward(0, 3, 5, 2, 2, None, None, None, None) == 0
    assert lib.nesie_furthest_point_sampling_wrapper(2, 10, 0, None, None, None, None) == 0


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from nesie_amd.mmdet3d_ops import ball_query, furthest_point_sample
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        furthest_point_sample(torch.rand(1, 32, 3), 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ball_query(0.0, 0.5, 4, torch.rand(1, 32, 3), torch.rand(1, 4, 3))


def test_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "nesie_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), \
                    f"{f} imports the oracle"


def test_cu_budget_is_validated_and_restored():
    """nesie_set_cu_count: host-side sizing state only (no GPU call): multiples of 8 in [8, 256],
    anything else is refused with the usual status + message; HipKernels.cu_budget restores the
    previous value on exit, also when the body raises."""
    import pytest
    from nesie_amd.kernels import HipKernels
    lib = _lib.load()
    assert lib.nesie_get_cu_count() == 256
    for bad in (0, 4, 100, 260, -8):
        with pytest.raises(RuntimeError, match="set_cu_count"):
            _lib.call("nesie_set_cu_count", bad)
    assert lib.nesie_get_cu_count() == 256
    with HipKernels.cu_budget(232):
        assert lib.nesie_get_cu_count() == 232
        with HipKernels.cu_budget(248):
            assert lib.nesie_get_cu_count() == 248
        assert lib.nesie_get_cu_count() == 232
    assert lib.nesie_get_cu_count() == 256
    with pytest.raises(ValueError):
        with HipKernels.cu_budget(224):
            raise ValueError("body failed")
    assert lib.nesie_get_cu_count() == 256
