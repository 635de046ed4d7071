"""Mirror of ``mmdet3d/ops/roiaware_pool3d/points_in_boxes.py:85-123``
(only ``points_in_boxes_batch`` is on the hot path)."""
import torch

from ..kernels import backend_for


def points_in_boxes_batch(points, boxes):
    """points (B,M,3) LiDAR frame, boxes (B,T,7) bottom-centre -> (B,M,T) int32."""
    assert boxes.shape[0] == points.shape[0], \
        f'Points and boxes should have the same batch size, got {boxes.shape[0]} and {points.shape[0]}'
    assert boxes.shape[2] == 7, f'boxes dimension should be 7, got unexpected shape {boxes.shape[2]}'
    assert points.shape[2] == 3, f'points dimension should be 3, got unexpected shape {points.shape[2]}'
    batch_size, num_points, _ = points.shape
    num_boxes = boxes.shape[1]
    box_idxs_of_pts = points.new_zeros((batch_size, num_points, num_boxes), dtype=torch.int)
    assert points.device == boxes.device, 'Points and boxes should be put on the same device'
    backend_for(points).points_in_boxes_batch(boxes.contiguous(), points.contiguous(),
                                              box_idxs_of_pts)
    return box_idxs_of_pts


def points_in_boxes_count(points, boxes):
    """points (B,M,3) LiDAR frame, boxes (B,T,7) -> (B,T) int32: how many of the scene's points
    lie in each box -- ``points_in_boxes_batch(points, boxes).sum(1)`` without the table (the
    non-empty test of NesieHead.multiclass_nms_single, nesie_head.py:744-750)."""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 3
    counts = points.new_empty((boxes.shape[0], boxes.shape[1]), dtype=torch.int)
    backend_for(points).points_in_boxes_count(boxes.contiguous(), points.contiguous(), counts)
    return counts
