"""Mirror of ``mmdet3d/ops/ball_query/ball_query.py:7-47``."""
import torch
from torch.autograd import Function

from ..kernels import backend_for


class BallQuery(Function):
    """First ``sample_num`` neighbours in index order inside [min_r, max_r)."""

    @staticmethod
    def forward(ctx, min_radius: float, max_radius: float, sample_num: int,
                xyz: torch.Tensor, center_xyz: torch.Tensor) -> torch.Tensor:
        assert center_xyz.is_contiguous()
        assert xyz.is_contiguous()
        assert min_radius < max_radius
        B, N, _ = xyz.size()
        npoint = center_xyz.size(1)
        idx = xyz.new_zeros((B, npoint, sample_num), dtype=torch.int32)
        backend_for(xyz).ball_query_wrapper(B, N, npoint, min_radius, max_radius,
                                            sample_num, center_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


ball_query = BallQuery.apply
