"""Max over the neighbourhood axis of a grouped tensor -- the pooling step of
``BasePointSAModule._pool_features`` (reference point_sa_module.py:136-158, an ATen
``F.max_pool2d(kernel=[1, nsample])``) and of ``MiniPointNet`` (``torch.max(dim=-1)``,
side_pooling_module.py:361,368) on the native kernel."""
import torch
from torch.autograd import Function

from ..kernels import backend_for


class GroupMaxPool(Function):
    """(..., ns) -> (...): max over the last axis, gradient to the first arg-max."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = x.new_empty(x.shape[:-1])
        arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
        backend_for(x).group_max_pool_forward(x, out, arg)
        ctx.save_for_backward(arg)
        ctx.ns = x.shape[-1]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (arg,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        grad_x = grad_out.new_empty(tuple(grad_out.shape) + (ctx.ns,))
        backend_for(grad_out).group_max_pool_backward(grad_out, arg, grad_x)
        return grad_x


class GroupMaxPoolShared(Function):
    """x (..., ns) -> (max over ns, x itself): for a tensor that feeds BOTH the max and another
    consumer (MiniPointNet's f: pooled into the global feature and convolved as the local one,
    side_pooling_module.py:359-365).  The backward adds the pooled gradient into the dense
    gradient of the second output in place -- no one-hot tensor, no full-size add."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out = x.new_empty(x.shape[:-1])
        arg = torch.empty(x.shape[:-1], dtype=torch.uint8, device=x.device)
        backend_for(x).group_max_pool_forward(x, out, arg)
        ctx.save_for_backward(arg)
        ctx.ns = x.shape[-1]
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, grad_out, grad_same):
        (arg,) = ctx.saved_tensors
        backend = backend_for(arg)
        if grad_same is None:
            grad_x = grad_out.new_empty(tuple(grad_out.shape) + (ctx.ns,))
            backend.group_max_pool_backward(grad_out.contiguous(), arg, grad_x)
            return grad_x
        grad_x = grad_same if grad_same.is_contiguous() else grad_same.contiguous()
        if grad_out is not None:
            backend.group_max_pool_backward_add(grad_out.contiguous(), arg, grad_x)
        return grad_x


def group_max_pool_shared(x):
    """-> (max over the last axis, x) with the two gradients merged in place."""
    ns = x.shape[-1]
    if 4 <= ns <= 64 and (ns & (ns - 1)) == 0 and x.dtype == torch.float32:
        return GroupMaxPoolShared.apply(x)
    return torch.max(x, dim=-1).values, x


def group_max_pool(x):
    ns = x.shape[-1]
    if 4 <= ns <= 64 and (ns & (ns - 1)) == 0 and x.dtype == torch.float32:
        return GroupMaxPool.apply(x)
    return torch.max(x, dim=-1).values  # shapes the kernel is not built for (not on the hot path)
