"""``mmdet3d.core.post_processing`` slice of the VoteNet/Nesie test path:
``aligned_3d_nms`` (box3d_nms.py:129-176).  The reference walks the candidate list in python
(about ten small launches and a host sync per pick); here every scene of the batch is one
workgroup of a single launch."""
import torch

from .kernels import backend_for


def batched_aligned_3d_nms(boxes, scores, classes, thresh, valid=None):
    """boxes (B,K,6) axis-aligned, scores (B,K), classes (B,K) any integer type,
    valid (B,K) bool or None -> picks (B,K) int32 = kept positions in pick order (-1 padded),
    count (B) int32.  No host synchronisation."""
    b, k = scores.shape
    picks = torch.empty((b, k), dtype=torch.int32, device=boxes.device)
    count = torch.empty((b,), dtype=torch.int32, device=boxes.device)
    backend_for(boxes).aligned_3d_nms(
        boxes.contiguous().float(), scores.contiguous().float(),
        classes.to(torch.int32).contiguous(),
        None if valid is None else valid.to(torch.uint8).contiguous(), thresh, picks, count)
    return picks, count


def aligned_3d_nms(boxes, scores, classes, thresh):
    """Same signature and result as the reference function: boxes (n,6), scores (n),
    classes (n) -> LongTensor of the selected indices, best score first."""
    if boxes.shape[0] == 0:
        return boxes.new_zeros((0,), dtype=torch.long)
    picks, count = batched_aligned_3d_nms(boxes.unsqueeze(0), scores.unsqueeze(0),
                                          classes.unsqueeze(0), thresh)
    return picks[0, :int(count[0])].long()
