"""``mmdet3d.core.evaluation.indoor_eval`` (indoor_eval.py:7-328): per-class average precision
and recall of indoor detections at several 3-D IoU thresholds.

Same numbers as the reference, arranged for the device: the reference calls the rotated
overlap op once per (class, scene) on freshly re-boxed tensors (:92-104); here every scene's
detection x ground-truth IoU matrix comes from ONE ``DepthInstance3DBoxes.overlaps`` call and
the per-class blocks are slices of it (the pairs are independent, so the values are the
same).  Greedy matching, the precision/recall curves and the area-under-curve AP follow
:112-160 and :7-52 step for step.
"""
import numpy as np
import torch

from .votenet.boxes import DepthInstance3DBoxes


def average_precision(recalls, precisions, mode='area'):
    """AP of one or several (recall, precision) curves (:7-52); float32 array."""
    recalls, precisions = np.asarray(recalls), np.asarray(precisions)
    if recalls.ndim == 1:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    scales = recalls.shape[0]
    ap = np.zeros(scales, dtype=np.float32)
    if mode == 'area':
        col = lambda v: np.full((scales, 1), v, dtype=recalls.dtype)  # noqa: E731
        mrec = np.hstack((col(0), recalls, col(1)))
        mpre = np.hstack((col(0), precisions, col(0)))
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]   # running max from the right
        for i in range(scales):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(scales):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
            ap /= 11   # inside the loop over scales, as in the reference (:46-47)
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def _curves(conf, scene, ious, n_gt_of, iou_thr):
    """Greedy matching of one class (:112-160).  conf (D,), scene (D,) scene id of every
    detection, ious: list of D rows (IoU against the GT boxes of that class in its scene),
    n_gt_of {scene: count}.  -> [(recall, precision, ap)] per threshold."""
    npos = int(sum(n_gt_of.values()))
    order = np.argsort(-np.array(conf))          # dtype of the scores, as in the reference (:115)
    taken = [{s: np.zeros(n, dtype=bool) for s, n in n_gt_of.items()} for _ in iou_thr]
    nd = len(order)
    tp = [np.zeros(nd) for _ in iou_thr]
    fp = [np.zeros(nd) for _ in iou_thr]
    for d, src in enumerate(order):
        row = ious[src]
        best, jmax = -np.inf, -1
        if n_gt_of[scene[src]] > 0 and len(row):
            clean = np.where(np.isnan(row), -np.inf, row)   # `iou > iou_max` skips NaN
            jmax = int(np.argmax(clean))         # first maximum, like the `>` scan
            best = clean[jmax]
        for t, thr in enumerate(iou_thr):
            if best > thr and not taken[t][scene[src]][jmax]:
                tp[t][d] = 1.
                taken[t][scene[src]][jmax] = True
            else:
                fp[t][d] = 1.
    out = []
    for t in range(len(iou_thr)):
        ctp, cfp = np.cumsum(tp[t]), np.cumsum(fp[t])
        with np.errstate(divide='ignore', invalid='ignore'):   # a predicted class without GT
            recall = ctp / float(npos) if npos else ctp / np.float64(0)
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def eval_map_recall(pred, gt, ovthresh=None, iou_of=None):
    """pred {class: {scene: [(box row index, score)]}}, gt {class: {scene: [gt row index]}},
    iou_of(scene) -> (n_det, n_gt) numpy IoU matrix of that scene.  Returns (recall,
    precision, ap), each a list over thresholds of {class: array} (:163-199)."""
    results = {}
    for label in gt.keys():
        if label not in pred:
            continue
        conf, scene, rows, n_gt_of = [], [], [], {}
        for s, members in gt[label].items():
            n_gt_of[s] = len(members)
        for s, dets in pred[label].items():
            if not dets:
                continue
            cols = gt[label][s]
            block = iou_of(s)[np.asarray([i for i, _ in dets])][:, np.asarray(cols, dtype=np.int64)] \
                if cols else None
            for k, (_, score) in enumerate(dets):
                conf.append(score)
                scene.append(s)
                rows.append(block[k] if block is not None else np.zeros(1))
        results[label] = _curves(conf, scene, rows, n_gt_of, ovthresh)
    recall = [{} for _ in ovthresh]
    precision = [{} for _ in ovthresh]
    ap = [{} for _ in ovthresh]
    for label in gt.keys():
        for t in range(len(ovthresh)):
            if label in pred:
                recall[t][label], precision[t][label], ap[t][label] = results[label][t]
            else:
                recall[t][label] = precision[t][label] = ap[t][label] = np.zeros(1)
    return recall, precision, ap


def indoor_eval(gt_annos, dt_annos, metric, label2cat, logger=None, box_type_3d=None,
                box_mode_3d=None):
    """gt_annos: [{'gt_num', 'gt_boxes_upright_depth' (n,6|7) gravity-centre boxes, 'class'}],
    dt_annos: [{'boxes_3d', 'scores_3d', 'labels_3d'}] (``bbox3d2result``), metric: IoU
    thresholds -> {'<cat>_AP_0.25', '<cat>_rec_0.25', 'mAP_0.25', 'mAR_0.25', ...}
    (:202-328)."""
    assert len(dt_annos) == len(gt_annos)
    box_type_3d = box_type_3d or DepthInstance3DBoxes
    pred, gt, iou_cache, scene_boxes = {}, {}, {}, {}
    for s, (det, gta) in enumerate(zip(dt_annos, gt_annos)):
        labels = np.asarray(det['labels_3d'].cpu() if torch.is_tensor(det['labels_3d'])
                            else det['labels_3d'])
        scores = np.asarray(det['scores_3d'].cpu() if torch.is_tensor(det['scores_3d'])
                            else det['scores_3d'])
        det_boxes = det['boxes_3d'].convert_to(box_mode_3d)
        for i in range(len(labels)):
            label = int(labels[i])
            pred.setdefault(label, {}).setdefault(s, []).append((i, scores[i]))
            gt.setdefault(label, {}).setdefault(s, [])
        if gta['gt_num'] != 0:
            raw = np.asarray(gta['gt_boxes_upright_depth'], dtype=np.float32)
            gt_boxes = box_type_3d(raw, box_dim=raw.shape[-1],
                                   origin=(0.5, 0.5, 0.5)).convert_to(box_mode_3d)
            gt_labels = gta['class']
        else:
            gt_boxes = box_type_3d(np.array([], dtype=np.float32))
            gt_labels = np.array([], dtype=np.int64)
        for i in range(len(gt_labels)):
            gt.setdefault(gt_labels[i], {}).setdefault(s, []).append(i)
        scene_boxes[s] = (det_boxes, gt_boxes)

    def iou_of(s):
        if s not in iou_cache:
            d, g = scene_boxes[s]
            iou_cache[s] = d.overlaps(d, g).cpu().numpy()
        return iou_cache[s]

    rec, prec, ap = eval_map_recall(pred, gt, metric, iou_of)
    ret, header = {}, ['classes']
    columns = [[label2cat[label] for label in ap[0].keys()] + ['Overall']]
    for t, thr in enumerate(metric):
        header += [f'AP_{thr:.2f}', f'AR_{thr:.2f}']
        for label in ap[t].keys():
            ret[f'{label2cat[label]}_AP_{thr:.2f}'] = float(ap[t][label][0])
        ret[f'mAP_{thr:.2f}'] = float(np.mean(list(ap[t].values())))
        columns.append([f'{float(np.ravel(v)[0]):.4f}' for v in ap[t].values()] + [f"{ret[f'mAP_{thr:.2f}']:.4f}"])
        rec_list = []
        for label in rec[t].keys():
            ret[f'{label2cat[label]}_rec_{thr:.2f}'] = float(rec[t][label][-1])
            rec_list.append(rec[t][label][-1])
        ret[f'mAR_{thr:.2f}'] = float(np.mean(rec_list))
        columns.append([f'{float(v):.4f}' for v in rec_list] + [f"{ret[f'mAR_{thr:.2f}']:.4f}"])
    if logger != 'silent':
        rows = [header] + [list(r) for r in zip(*columns)]
        width = [max(len(str(r[c])) for r in rows) for c in range(len(header))]
        text = '\n'.join(' | '.join(str(v).ljust(w) for v, w in zip(r, width)) for r in rows)
        (logger.info if hasattr(logger, 'info') else print)('\n' + text)
    return ret
