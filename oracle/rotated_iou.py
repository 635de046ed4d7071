"""CPU restatement (torch, any float dtype, differentiable) of the reference's rotated 3-D IoU
chain -- TEST INFRASTRUCTURE: the checker of ``nesie_iou3d_forward`` (csrc/iou3d.hip) and the IoU
of the CPU-oracle leg; the product computes the whole chain in one HIP kernel.

Follows, step for step:
  corners of a BEV rectangle                 rotated_iou/oriented_iou_loss.py:6-36
  4 x 4 edge crossings (t, u in (0, 1))      rotated_iou/box_intersection_2d.py:13-54
  corners of one box inside the other        box_intersection_2d.py:57-98 (1e-6 slack)
  24 candidate vertices, centred, ordered    box_intersection_2d.py:101-147 + sort_vertices
  shoelace area of the 9 ordered vertices    box_intersection_2d.py:150-166
  BEV IoU, z overlap, volumes                oriented_iou_loss.py:39-58, 86-109
``order`` = callable(vertices (B,N,24,2), mask (B,N,24) bool, num_valid (B,N) int32) -> (B,N,9)
indices: the sort_vertices op under test (oracle C restatement, or the HIP kernel).
"""
import torch

_TINY = 1e-8


def _cross(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def bev_corners(rect):
    """(…, 5) = x, y, w, h, alpha -> (…, 4, 2); corner order (+,+) (-,+) (-,-) (+,-) turned by
    +alpha."""
    cx, cy, w, h, al = rect.unbind(-1)
    hx = torch.stack([w, -w, -w, w], -1) * 0.5
    hy = torch.stack([h, h, -h, -h], -1) * 0.5
    c, s = torch.cos(al).unsqueeze(-1), torch.sin(al).unsqueeze(-1)
    return torch.stack([hx * c - hy * s + cx.unsqueeze(-1), hx * s + hy * c + cy.unsqueeze(-1)], -1)


def edge_crossings(ca, cb):
    """every edge of rectangle a against every edge of b -> points (…, 4, 4, 2) (zero where the
    edges do not cross), mask (…, 4, 4)."""
    p, d1 = ca.unsqueeze(-2), (torch.roll(ca, -1, -2) - ca).unsqueeze(-2)      # (…, 4, 1, 2)
    q, d2 = cb.unsqueeze(-3), (torch.roll(cb, -1, -2) - cb).unsqueeze(-3)      # (…, 1, 4, 2)
    denom = _cross(d1, d2)
    t_num = _cross(q - p, d2)
    u_num = _cross(q - p, d1)
    parallel = denom == 0
    minus = torch.full_like(denom, -1.0)
    t = torch.where(parallel, minus, t_num / denom)
    u = torch.where(parallel, minus, u_num / denom)
    hit = (t > 0) & (t < 1) & (u > 0) & (u < 1)
    t_safe = t_num / (denom + _TINY)
    pts = (p + t_safe.unsqueeze(-1) * d1) * hit.unsqueeze(-1).to(ca.dtype)
    return pts, hit


def corners_inside(ca, cb):
    """which corners of a lie inside b (edges included, 1e-6 slack on the normalised
    projections onto b's first and last edge)."""
    o = cb[..., 0:1, :]
    e1, e2 = cb[..., 1:2, :] - o, cb[..., 3:4, :] - o
    r = ca - o
    s1 = (r * e1).sum(-1) / (e1 * e1).sum(-1)
    s2 = (r * e2).sum(-1) / (e2 * e2).sum(-1)
    return (s1 > -1e-6) & (s1 < 1 + 1e-6) & (s2 > -1e-6) & (s2 < 1 + 1e-6)


def intersection_area(ca, cb, order):
    pts, hit = edge_crossings(ca, cb)
    lead = ca.shape[:-2]
    verts = torch.cat([ca, cb, pts.reshape(*lead, 16, 2)], -2)                  # (…, 24, 2)
    valid = torch.cat([corners_inside(ca, cb), corners_inside(cb, ca), hit.reshape(*lead, 16)], -1)
    count = valid.sum(-1).to(torch.int32)
    centre = (verts * valid.unsqueeze(-1).to(verts.dtype)).sum(-2, keepdim=True) \
        / count.unsqueeze(-1).unsqueeze(-1)
    idx = order((verts - centre).detach(), valid, count)                        # (…, 9)
    ring = torch.gather(verts, -2, idx.long().unsqueeze(-1).expand(*idx.shape, 2))
    twice = (ring[..., :-1, 0] * ring[..., 1:, 1] - ring[..., :-1, 1] * ring[..., 1:, 0]).sum(-1)
    return twice.abs() * 0.5


def rotated_iou_3d(box_a, box_b, order):
    """(B, N, 7) = centre, size, yaw (boxes turned about z only) -> IoU (B, N)."""
    ra = torch.cat([box_a[..., 0:2], box_a[..., 3:5], box_a[..., 6:7]], -1)
    rb = torch.cat([box_b[..., 0:2], box_b[..., 3:5], box_b[..., 6:7]], -1)
    inter2d = intersection_area(bev_corners(ra), bev_corners(rb), order)
    union2d = ra[..., 2] * ra[..., 3] + rb[..., 2] * rb[..., 3] - inter2d
    iou2d = inter2d / union2d
    top = torch.min(box_a[..., 2] + box_a[..., 5] * 0.5, box_b[..., 2] + box_b[..., 5] * 0.5)
    bot = torch.max(box_a[..., 2] - box_a[..., 5] * 0.5, box_b[..., 2] - box_b[..., 5] * 0.5)
    inter3d = iou2d * union2d * (top - bot).clamp_min(0)
    vol = box_a[..., 3] * box_a[..., 4] * box_a[..., 5] + box_b[..., 3] * box_b[..., 4] * box_b[..., 5]
    return inter3d / (vol - inter3d)
