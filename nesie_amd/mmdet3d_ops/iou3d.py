"""``mmdet3d.ops.iou3d`` slice used by evaluation (iou3d_utils.py, src/iou3d.cpp:66-90):
the rotated BEV overlap behind ``BaseInstance3DBoxes.overlaps``."""
import torch

from ..kernels import backend_for


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N,5), (M,5) (x1, y1, x2, y2, ry) -> (N,M) overlap areas
    (``iou3d_cuda.boxes_overlap_bev_gpu``; the reference fills a caller-made tensor)."""
    ans = boxes_a.new_zeros((boxes_a.shape[0], boxes_b.shape[0]))
    backend_for(boxes_a).boxes_overlap_bev(boxes_a.contiguous().float(),
                                           boxes_b.contiguous().float(), ans)
    return ans
