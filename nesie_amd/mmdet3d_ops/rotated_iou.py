"""``mmdet3d/ops/rotated_iou`` on the native kernels: ``sort_v`` (cuda_op/cuda_ext.py:6-17 ->
``nesie_sort_vertices_forward``) and ``cal_iou_3d`` (oriented_iou_loss.py:86-109), which the
reference evaluates as ~100 torch kernels around ``sort_v`` and this library as ONE kernel
(``nesie_iou3d_forward``: value + Jacobian w.r.t. the first box).  GIoU / DIoU / enclosing boxes
are outside the hot path (IoU3DLoss uses plain IoU).

There is no torch chain in the product: an injected test back end supplies its own
``rotated_iou_3d`` (the CPU oracle restates the reference chain in oracle/rotated_iou.py).
"""
import torch
from torch.autograd import Function

from ..kernels import backend_for


class SortVertices(Function):
    """vertices (B,N,24,2) f32, mask (B,N,24) bool, num_valid (B,N) i32 -> (B,N,9) i32."""

    @staticmethod
    def forward(ctx, vertices, mask, num_valid):
        if vertices.dtype != torch.float32 or mask.dtype != torch.bool \
                or num_valid.dtype != torch.int32:
            raise RuntimeError('sort_vertices: vertices f32, mask bool, num_valid int32')
        vertices, mask, num_valid = vertices.contiguous(), mask.contiguous(), num_valid.contiguous()
        B, N = vertices.shape[:2]
        idx = vertices.new_empty((B, N, 9), dtype=torch.int32)
        backend_for(vertices).sort_vertices_forward(vertices, mask, num_valid, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, gradout):
        return None, None, None


sort_v = SortVertices.apply


class RotatedIoU3D(Function):
    """cal_iou_3d as one native kernel: the IoU and its Jacobian w.r.t. the first box."""

    @staticmethod
    def forward(ctx, box3d1, box3d2):
        shape = box3d1.shape[:-1]
        b1 = box3d1.reshape(-1, 7).contiguous().float()
        b2 = box3d2.reshape(-1, 7).contiguous().float()
        iou = b1.new_empty(b1.shape[0])
        jac = b1.new_empty(b1.shape[0], 7) if box3d1.requires_grad else None
        backend_for(b1).iou3d_forward(b1, b2, iou, jac)
        ctx.save_for_backward(jac)
        ctx.in_shape = box3d1.shape
        return iou.view(shape)

    @staticmethod
    def backward(ctx, grad):
        (jac,) = ctx.saved_tensors
        if jac is None:
            return None, None
        return (grad.reshape(-1, 1) * jac).view(ctx.in_shape), None


def cal_iou_3d(box3d1, box3d2):
    """3-D IoU of (B,N,7) boxes rotated about z only; differentiable in ``box3d1`` (the second
    box is a target and never requires grad on this path)."""
    backend = backend_for(box3d1)
    if getattr(backend, 'name', '') == 'hip':
        if box3d2.requires_grad:
            raise RuntimeError('cal_iou_3d: the second box must not require grad')
        return RotatedIoU3D.apply(box3d1, box3d2)
    return backend.rotated_iou_3d(box3d1, box3d2)
