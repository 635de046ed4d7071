"""Mirror of ``mmdet3d/ops/rotated_iou`` for the IoU3D loss / IoU labels:
``cal_iou_3d`` (oriented_iou_loss.py:86-109), ``cal_iou`` (:39-58),
``box2corners_th`` (:6-36), ``oriented_box_intersection_2d``
(box_intersection_2d.py:13-184) and ``sort_v`` (cuda_op/cuda_ext.py:6-17).

The vertex ordering is the native ``sort_vertices`` kernel of libnesie_hip.so;
the differentiable geometry around it is torch glue, as in the reference.
GIoU / DIoU / enclosing boxes are outside the hot path (IoU3DLoss uses plain IoU).
"""
import torch
from torch.autograd import Function

from ..kernels import backend_for

EPSILON = 1e-8


class SortVertices(Function):
    """vertices (B,N,24,2) f32, mask (B,N,24) bool, num_valid (B,N) i32 -> (B,N,9) i32."""

    @staticmethod
    def forward(ctx, vertices, mask, num_valid):
        if vertices.dtype != torch.float32 or mask.dtype != torch.bool \
                or num_valid.dtype != torch.int32:
            raise RuntimeError('sort_vertices: vertices f32, mask bool, num_valid int32')
        vertices = vertices.contiguous()
        mask = mask.contiguous()
        num_valid = num_valid.contiguous()
        B, N = vertices.shape[:2]
        idx = vertices.new_empty((B, N, 9), dtype=torch.int32)
        backend_for(vertices).sort_vertices_forward(vertices, mask, num_valid, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, gradout):
        return None, None, None


sort_v = SortVertices.apply


def box_intersection_th(corners1, corners2):
    """Edge-edge intersections of two rectangles: (B,N,4,4,2) points + (B,N,4,4) mask
    (box_intersection_2d.py:13-54)."""
    # edge i runs from corner i to corner i+1 (roll instead of a [1,2,3,0] index list: no
    # host->device index copy, so the step stays hipGraph-capturable)
    line1 = torch.cat([corners1, torch.roll(corners1, -1, dims=2)], dim=3)
    line2 = torch.cat([corners2, torch.roll(corners2, -1, dims=2)], dim=3)
    line1_ext = line1.unsqueeze(3).repeat([1, 1, 1, 4, 1])
    line2_ext = line2.unsqueeze(2).repeat([1, 1, 4, 1, 1])
    x1, y1, x2, y2 = (line1_ext[..., i] for i in range(4))
    x3, y3, x4, y4 = (line2_ext[..., i] for i in range(4))
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    zero = num == .0
    t = torch.where(zero, torch.full_like(num, -1.), den_t / num)
    mask_t = (t > 0) * (t < 1)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    u = torch.where(zero, torch.full_like(num, -1.), -den_u / num)
    mask_u = (u > 0) * (u < 1)
    mask = mask_t * mask_u
    t = den_t / (num + EPSILON)
    intersections = torch.stack([x1 + t * (x2 - x1), y1 + t * (y2 - y1)], dim=-1)
    intersections = intersections * mask.float().unsqueeze(-1)
    return intersections, mask


def box1_in_box2(corners1, corners2):
    """Which corners of box1 lie in box2, edges included (:57-82)."""
    a = corners2[:, :, 0:1, :]
    b = corners2[:, :, 1:2, :]
    d = corners2[:, :, 3:4, :]
    ab = b - a
    am = corners1 - a
    ad = d - a
    p_ab = torch.sum(ab * am, dim=-1)
    norm_ab = torch.sum(ab * ab, dim=-1)
    p_ad = torch.sum(ad * am, dim=-1)
    norm_ad = torch.sum(ad * ad, dim=-1)
    cond1 = (p_ab / norm_ab > -1e-6) * (p_ab / norm_ab < 1 + 1e-6)
    cond2 = (p_ad / norm_ad > -1e-6) * (p_ad / norm_ad < 1 + 1e-6)
    return cond1 * cond2


def box_in_box_th(corners1, corners2):
    return box1_in_box2(corners1, corners2), box1_in_box2(corners2, corners1)


def build_vertices(corners1, corners2, c1_in_2, c2_in_1, inters, mask_inter):
    """24 candidate vertices + validity mask (:101-124)."""
    B, N = corners1.size()[:2]
    vertices = torch.cat([corners1, corners2, inters.view([B, N, -1, 2])], dim=2)
    mask = torch.cat([c1_in_2, c2_in_1, mask_inter.view([B, N, -1])], dim=2)
    return vertices, mask


def sort_indices(vertices, mask):
    """Mean-centre the valid vertices and order them (:127-147)."""
    num_valid = torch.sum(mask.int(), dim=2).int()
    mean = torch.sum(vertices * mask.float().unsqueeze(-1), dim=2, keepdim=True) \
        / num_valid.unsqueeze(-1).unsqueeze(-1)
    vertices_normalized = vertices - mean
    return sort_v(vertices_normalized.detach(), mask, num_valid).long()


def calculate_area(idx_sorted, vertices):
    """Shoelace area of the 9 gathered vertices (:150-166)."""
    idx_ext = idx_sorted.unsqueeze(-1).repeat([1, 1, 1, 2])
    selected = torch.gather(vertices, 2, idx_ext)
    total = selected[:, :, 0:-1, 0] * selected[:, :, 1:, 1] \
        - selected[:, :, 0:-1, 1] * selected[:, :, 1:, 0]
    total = torch.sum(total, dim=2)
    return torch.abs(total) / 2, selected


def oriented_box_intersection_2d(corners1, corners2):
    """Intersection area of two rotated rectangles given their corners (:169-184)."""
    inters, mask_inter = box_intersection_th(corners1, corners2)
    c12, c21 = box_in_box_th(corners1, corners2)
    vertices, mask = build_vertices(corners1, corners2, c12, c21, inters, mask_inter)
    sorted_indices = sort_indices(vertices, mask)
    return calculate_area(sorted_indices, vertices)


def box2corners_th(box):
    """(B,N,5) x,y,w,h,alpha -> (B,N,4,2) corners (oriented_iou_loss.py:6-36)."""
    B = box.size()[0]
    x, y, w, h, alpha = (box[..., i:i + 1] for i in range(5))
    # corner signs (+,+) (-,+) (-,-) (+,-) without a host->device constant (graph-safe)
    hw, hh = 0.5 * w, 0.5 * h
    x4 = torch.cat([hw, -hw, -hw, hw], dim=-1)
    y4 = torch.cat([hh, hh, -hh, -hh], dim=-1)
    corners = torch.stack([x4, y4], dim=-1)
    sin = torch.sin(alpha)
    cos = torch.cos(alpha)
    row1 = torch.cat([cos, sin], dim=-1)
    row2 = torch.cat([-sin, cos], dim=-1)
    rot_T = torch.stack([row1, row2], dim=-2)
    rotated = torch.bmm(corners.view([-1, 4, 2]), rot_T.view([-1, 2, 2]))
    rotated = rotated.view([B, -1, 4, 2])
    return rotated + torch.cat([x, y], dim=-1).unsqueeze(2)


def cal_iou(box1, box2):
    """BEV IoU of (B,N,5) boxes: iou, corners1, corners2, union (:39-58)."""
    corners1 = box2corners_th(box1)
    corners2 = box2corners_th(box2)
    inter_area, _ = oriented_box_intersection_2d(corners1, corners2)
    area1 = box1[:, :, 2] * box1[:, :, 3]
    area2 = box2[:, :, 2] * box2[:, :, 3]
    u = area1 + area2 - inter_area
    return inter_area / u, corners1, corners2, u


class RotatedIoU3D(Function):
    """cal_iou_3d as ONE native kernel: value + Jacobian w.r.t. the first box."""

    @staticmethod
    def forward(ctx, box3d1, box3d2):
        shape = box3d1.shape[:-1]
        b1 = box3d1.reshape(-1, 7).contiguous().float()
        b2 = box3d2.reshape(-1, 7).contiguous().float()
        iou = b1.new_empty(b1.shape[0])
        need_grad = box3d1.requires_grad
        jac = b1.new_empty(b1.shape[0], 7) if need_grad else None
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


def cal_iou_3d(box3d1, box3d2, verbose=False):
    """3-D IoU of (B,N,7) boxes rotated about z only (:86-109).  On the HIP back end the
    whole chain is one kernel (the second box must not require grad -- it never does on
    this path); other back ends evaluate the torch chain below."""
    if not verbose and getattr(backend_for(box3d1), 'name', '') == 'hip' \
            and not box3d2.requires_grad:
        return RotatedIoU3D.apply(box3d1, box3d2)
    return cal_iou_3d_torch(box3d1, box3d2, verbose)


def cal_iou_3d_torch(box3d1, box3d2, verbose=False):
    """The reference's torch chain (:86-109) with the native sort_vertices inside."""
    box1 = torch.cat([box3d1[..., 0:2], box3d1[..., 3:5], box3d1[..., 6:7]], dim=-1)  # x y w h a
    box2 = torch.cat([box3d2[..., 0:2], box3d2[..., 3:5], box3d2[..., 6:7]], dim=-1)
    zmax1 = box3d1[..., 2] + box3d1[..., 5] * 0.5
    zmin1 = box3d1[..., 2] - box3d1[..., 5] * 0.5
    zmax2 = box3d2[..., 2] + box3d2[..., 5] * 0.5
    zmin2 = box3d2[..., 2] - box3d2[..., 5] * 0.5
    z_overlap = (torch.min(zmax1, zmax2) - torch.max(zmin1, zmin2)).clamp_min(0.)
    iou_2d, corners1, corners2, u = cal_iou(box1, box2)
    intersection_3d = iou_2d * u * z_overlap
    v1 = box3d1[..., 3] * box3d1[..., 4] * box3d1[..., 5]
    v2 = box3d2[..., 3] * box3d2[..., 4] * box3d2[..., 5]
    u3d = v1 + v2 - intersection_3d
    if verbose:
        z_range = (torch.max(zmax1, zmax2) - torch.min(zmin1, zmin2)).clamp_min(0.)
        return intersection_3d / u3d, corners1, corners2, z_range, u3d
    return intersection_3d / u3d
