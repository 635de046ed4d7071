"""Student/teacher step helpers for the parity tests (test infrastructure): one step of
VoteNetNesie / VoteNetSAQE on any device and precision, with the teacher's pseudo labels RECORDED
by the first leg and REPLAYED by the others -- labels, boxes, qualities and validity are detached
inputs of the student's loss, and a thresholded decision that flips between two precisions would
change the loss itself (both outcomes legitimate), which says nothing about the gradient kernels."""
import copy

import torch

from nesie_amd import kernels
from nesie_amd.votenet import semi
from nesie_amd.votenet.nesie_head import GTBatch
from tests import _small


def as_double(model):
    """A float64 copy of a student/teacher detector (parameters, EMA buffers, jitter noise)."""
    m = copy.deepcopy(model).double()
    if getattr(m.bbox_head, 'jitter_noise', None) is not None:
        m.bbox_head.jitter_noise = tuple(t.double() for t in m.bbox_head.jitter_noise)
    return m


def semi_step(model, device, backend=None, full=False, dtype=torch.float32, book=None):
    """One student/teacher step (3 scenes, 1 labeled : 2 unlabeled) -> (losses, grads, pseudo labels).
    ``book``: a dict shared by the legs of one comparison -- the first leg stores its pseudo labels in
    it, later legs use them instead of their own."""
    from contextlib import nullcontext
    if full:
        from nesie_amd.scenes import make_batch
        model.init_label_state(120, 1081, device)
        pts, boxes, labels = make_batch(4242, 3, 40000)
    else:
        model.init_label_state(12, 108, device)
        pts, boxes, labels = _small.small_batch(batch=3, n=2048)
    g = torch.Generator().manual_seed(1)
    meta_t = semi.AugMeta.random(3, device, g, strong=False)
    meta_s = semi.AugMeta.random(3, device, g, strong=True)
    for meta in (meta_t, meta_s):
        meta.rot_mat, meta.scale, meta.trans = (t.to(dtype) for t in (meta.rot_mat, meta.scale, meta.trans))
    pts = pts.to(device=device, dtype=dtype)
    gt = GTBatch.collate(boxes[:1], labels[:1], device)
    gt.boxes, gt.valid = gt.boxes.to(dtype), gt.valid.to(dtype)
    rows = torch.tensor([5, 17], device=device)
    picks = {}
    inner = model.get_pseudo_labels

    def pseudo(preds, name='ScanNet'):
        out = inner(preds, name)
        own = dict(labels=out[0].cpu(), boxes=out[1].double().cpu(), quality=out[2].double().cpu(),
                   valid=out[3].cpu())
        picks.update(own)
        if book is None:
            return out
        if 'pseudo' not in book:
            book['pseudo'] = own
            return out
        rec = book['pseudo']
        return (rec['labels'].to(device), rec['boxes'].to(device=device, dtype=out[1].dtype),
                rec['quality'].to(device=device, dtype=out[2].dtype), rec['valid'].to(device))
    model.get_pseudo_labels = pseudo
    # ... and the proposal <-> ground-truth assignment of both losses (nesie_head.py:656-676: a
    # proposal whose distance to a centre sits on the 0.3 / 0.6 threshold takes another objectness
    # label in another precision, and one term of three losses appears or vanishes)
    head = model.bbox_head
    inner_targets = head.get_targets
    calls = [0]

    def targets(*a, **kw):
        out = inner_targets(*a, **kw)
        if book is None:
            return out
        rec = book.setdefault('targets', [])
        i, calls[0] = calls[0], calls[0] + 1
        if i >= len(rec):
            rec.append(tuple(t.detach().cpu() for t in out))
            return out
        return tuple(r.to(device=o.device, dtype=o.dtype) for r, o in zip(rec[i], out))
    head.get_targets = targets
    # ... and every ball-query result: the backbone's are functions of the input coordinates (equal on
    # every leg anyway), the vote aggregation's groups PREDICTED votes around predicted centres -- a vote
    # at the radius joins or leaves a ball with the last bit of a coordinate
    bq_backend = backend if backend is not None else kernels.backend_for(pts)
    inner_bq = bq_backend.ball_query_wrapper
    bq_calls = [0]

    def ball_query(b_, n_, m_, min_r, max_r, ns_, new_xyz, xyz, idx):
        inner_bq(b_, n_, m_, min_r, max_r, ns_, new_xyz, xyz, idx)
        if book is None:
            return
        rec = book.setdefault('ball_query', [])
        i, bq_calls[0] = bq_calls[0], bq_calls[0] + 1
        if i >= len(rec):
            rec.append(idx.detach().cpu().clone())
        else:
            book.setdefault('ball_query_differed', []).append(int((rec[i] != idx.cpu()).any(-1).sum()))
            idx.copy_(rec[i].to(idx.device))
    bq_backend.ball_query_wrapper = ball_query
    for p in model.parameters():
        p.grad = None
    try:
        with (kernels.use_backend(backend) if backend is not None else nullcontext()):
            losses = model.forward_train(meta_s.apply_points(pts), meta_t.apply_points(pts), gt,
                                         [True, False, False], meta_s, meta_t, rows)
            model.parse_losses(losses).backward()
    finally:
        del model.get_pseudo_labels
        del head.get_targets
        del bq_backend.ball_query_wrapper
    grads = _small.grads_of(model, cpu=True)
    return {k: v.detach().double().cpu() for k, v in losses.items()}, grads, picks
