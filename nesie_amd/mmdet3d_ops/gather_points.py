"""Mirror of ``mmdet3d/ops/gather_points/gather_points.py:7-52``."""
import torch
from torch.autograd import Function

from ..kernels import backend_for


class GatherPoints(Function):
    """features (B,C,N), indices (B,M) int32 -> (B,C,M)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert indices.is_contiguous()
        B, npoint = indices.size()
        _, C, N = features.size()
        output = features.new_empty((B, C, npoint))
        backend_for(features).gather_points_wrapper(B, C, N, npoint, features, indices,
                                                    output)
        ctx.for_backwards = (indices, C, N)
        ctx.mark_non_differentiable(indices)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        B, npoint = idx.size()
        grad_out_data = grad_out.data.contiguous()
        backend = backend_for(grad_out_data)
        build = getattr(backend, 'scatter_index', None)
        csr = None if build is None else build(idx.view(B, npoint, 1), N)
        if csr is not None:     # fixed order whatever the indices repeat (gather_points_cuda.cu:51-70: atomicAdd)
            grad_features = grad_out.new_empty((B, C, N))                    # (written in full)
            backend.group_points_backward_csr(grad_out_data.view(B, C, npoint, 1), csr[0], csr[1],
                                              grad_features.data)
        else:
            grad_features = grad_out.new_zeros((B, C, N))
            backend.gather_points_grad_wrapper(B, C, N, npoint, grad_out_data, idx,
                                               grad_features.data)
        return grad_features, None


gather_points = GatherPoints.apply
