"""VoteNet detector step (``mmdet3d/models/detectors/votenet.py:27-60``) and the
Nesie-VoteNet ScanNet hyper-parameters (``configs/Nesie/nesie-votenet-scannet-pretrain-010.py:2-111``
restated as a plain dict; the file itself does not travel)."""
import copy

import torch
from torch import nn

from ..mmdet3d_ops.norm import deferred_bn_counters
from .backbone import PointNet2SASSG
from .nesie_head import NesieHead


def nesie_votenet_scannet_cfg():
    return dict(
        backbone=dict(
            in_channels=4, num_points=(2048, 1024, 512, 256), radius=(0.2, 0.4, 0.8, 1.2),
            num_samples=(64, 32, 16, 16),
            sa_channels=((64, 64, 128), (128, 128, 256), (128, 128, 256), (128, 128, 256)),
            fp_channels=((256, 256), (256, 256)), norm_cfg=dict(type='BN2d'),
            sa_cfg=dict(type='PointSAModule', pool_mod='max', use_xyz=True,
                        normalize_xyz=True)),
        bbox_head=dict(
            num_classes=18, reg_max=32, alpha=1.0,
            vote_module_cfg=dict(
                in_channels=256, vote_per_seed=1, gt_per_seed=3, conv_channels=(256, 256),
                conv_cfg=dict(type='Conv1d'), norm_cfg=dict(type='BN1d'), norm_feats=True,
                vote_loss=dict(type='ChamferDistance', mode='l1', reduction='none',
                               loss_dst_weight=10.0)),
            vote_aggregation_cfg=dict(
                type='PointSAModule', num_point=256, radius=0.3, num_sample=16,
                mlp_channels=[256, 128, 128, 128], use_xyz=True, normalize_xyz=True),
            pred_layer_cfg=dict(in_channels=128, shared_conv_channels=(128, 128), bias=True),
            objectness_loss=dict(type='CrossEntropyLoss', class_weight=[0.2, 0.8],
                                 reduction='sum', loss_weight=5.0),
            center_loss=dict(type='ChamferDistance', mode='l2', reduction='sum',
                             loss_src_weight=10.0, loss_dst_weight=10.0),
            iou_loss=dict(type='IoU3DLoss', reduction='sum', loss_weight=3.0),
            semantic_loss=dict(type='CrossEntropyLoss', reduction='sum', loss_weight=1.0),
            iou_pred_loss=dict(type='GeneralQualityFocalLoss', reduction='sum',
                               use_sigmoid=False, beta=2.0, loss_weight=3.0),
            surface_loss=dict(type='SurfaceLoss', func_type='MSELoss', beta=5.0,
                              reduction='sum', loss_weight=10.0),
            side_loss=dict(type='SidePredLoss', label_func_type='SmoothL1Loss',
                           loss_func_type='MSELoss', beta=5.0, reduction='sum',
                           loss_weight=1.0),
            grid_conv_cfg=dict(num_class=18, num_heading_bin=1, num_size_cluster=18,
                               mean_size_arr_path=None, num_proposal=256,
                               sampling='seed_fps', query_feats='seed')),
        train_cfg=dict(pos_distance_thr=0.3, neg_distance_thr=0.6, sample_mod='vote'),
        test_cfg=dict(sample_mod='seed', nms_thr=0.25, score_thr=0.05,
                      per_class_proposal=True, add_info=True),
        optimizer=dict(type='AdamW', lr=0.008, weight_decay=0.01),
        grad_clip=dict(max_norm=10, norm_type=2),
    )


def saqe_votenet_scannet_cfg():
    """``configs/SAQE/saqe-votenet-scannet-pretrain-010.py``: the Nesie config with the SAQE
    head's two extra losses (``:78-81``)."""
    cfg = nesie_votenet_scannet_cfg()
    cfg['bbox_head'].update(
        angle_loss=dict(type='SmoothL1Loss', reduction='sum', loss_weight=10.0),
        angle_pred_loss=dict(type='MSELoss', reduction='sum', loss_weight=1.0))
    cfg['head_type'] = 'SAQEHead'
    return cfg


def bbox3d2result(bboxes, scores, labels, attrs=None):
    """core/bbox/transforms.py:49-75: detection results as host tensors."""
    result = dict(boxes_3d=bboxes.to('cpu'), scores_3d=scores.cpu(), labels_3d=labels.cpu())
    if attrs is not None:
        result['attrs_3d'] = attrs.cpu()
    return result


class VoteNet(nn.Module):
    """backbone -> bbox head -> losses dict; total loss = sum of every entry whose key
    contains 'loss' (mmdet BaseDetector._parse_losses)."""

    def __init__(self, backbone, bbox_head, train_cfg=None, test_cfg=None, head_type='NesieHead'):
        super().__init__()
        self.backbone = PointNet2SASSG(**backbone)
        if head_type == 'SAQEHead':
            from .saqe_head import SAQEHead as head_cls
        else:
            head_cls = NesieHead
        self.bbox_head = head_cls(**bbox_head, train_cfg=train_cfg, test_cfg=test_cfg)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg

    keep_head_inputs = False
    head_inputs = None

    def take_head_inputs(self):
        """The boundary tensors of the last forward_train (needs ``keep_head_inputs = True``)."""
        out, self.head_inputs = self.head_inputs, None
        return out

    def extract_feat(self, points, precomputed=None):
        return self.backbone(points, precomputed=precomputed)

    def forward_train(self, points, img_metas, gt_bboxes_3d, gt_labels_3d,
                      pts_semantic_mask=None, pts_instance_mask=None, gt_bboxes_ignore=None,
                      precomputed=None):
        points_cat = torch.stack(points) if isinstance(points, (list, tuple)) else points
        votes = None
        if isinstance(precomputed, dict):   # {'indices': [...], 'vote_targets': (t, mask)}
            precomputed, votes = precomputed.get('indices'), precomputed.get('vote_targets')
        with deferred_bn_counters():  # one launch for all num_batches_tracked increments
            x = self.extract_feat(points_cat, precomputed)
            if self.keep_head_inputs:
                # the one tensor that carries gradient from the head back into the backbone
                # (dp.backward_in_two_phases cuts the backward pass here); opt-in, because it
                # keeps the autograd graph alive until take_head_inputs()
                self.head_inputs = [x['fp_features'][-1]]
            bbox_preds = self.bbox_head(x, self.train_cfg['sample_mod'])
        return self.bbox_head.loss(bbox_preds, points_cat, gt_bboxes_3d, gt_labels_3d,
                                   pts_semantic_mask, pts_instance_mask, img_metas,
                                   gt_bboxes_ignore=gt_bboxes_ignore, vote_targets=votes)

    def simple_test(self, points, img_metas, imgs=None, rescale=False):
        """Test-time forward (detectors/votenet.py:62-83, votenet_nesie.py:326-348): head in
        `test_cfg.sample_mod`, box decoding + NMS, results moved to the host."""
        points_cat = torch.stack(points) if isinstance(points, (list, tuple)) else points
        cfg = self.test_cfg if isinstance(self.test_cfg, dict) else vars(self.test_cfg)
        with torch.no_grad():
            x = self.extract_feat(points_cat)
            bbox_preds = self.bbox_head(x, cfg['sample_mod'])
            tensors = self.bbox_head.detect_tensors(points_cat, bbox_preds,
                                                    cfg.get('use_iou_for_nms', True))
            # the results go to the host anyway (bbox3d2result): one transfer of the five small
            # per-proposal tensors, then the per-scene selection there -- instead of a boolean
            # indexing (a device->host synchronisation) per scene
            bbox_list = self.bbox_head.boxes_from_tensors(tuple(t.cpu() for t in tensors),
                                                          img_metas)
        return [bbox3d2result(bboxes, scores, labels) for bboxes, scores, labels in bbox_list]

    def graphed_simple_test(self, batch, num_points, feat_dim=4, device=None):
        """-> callable(points, img_metas=None) with ``simple_test``'s results, the device half
        (backbone, head, score fusion, point counts, NMS) captured once as a hipGraph for a fixed
        (batch, num_points) and replayed per call; only the per-scene boolean selection and the
        copy to the host stay eager."""
        return GraphedSimpleTest(self, batch, num_points, feat_dim, device)

    def stacked_parameter_groups(self):
        """Parameter lists the step uses stacked (``SidePooling.stacked_parameter_groups``), for
        ``dp.FlatTrainState(model.parameters(), stack_groups=...)``."""
        pool = getattr(self.bbox_head, 'grid_conv', None)
        return pool.stacked_parameter_groups() if hasattr(pool, 'stacked_parameter_groups') else []

    @staticmethod
    def parse_losses(losses):
        """Total = sum of every entry whose key contains 'loss' (mmdet BaseDetector._parse_losses);
        0-dim terms are added in one stack + sum instead of a chain of scalar additions."""
        terms = [v for k, v in losses.items() if 'loss' in k]
        if len(terms) > 2 and all(torch.is_tensor(v) and v.dim() == 0 for v in terms):
            return torch.stack(terms).sum()
        return sum(terms)


class GraphedSimpleTest:
    """``simple_test`` for a fixed (batch, num_points) as two hipGraphs: the FPS / ball-query /
    3-NN index chain of the backbone (coordinates only, ~5.5 of the 11.4 ms eval forward at
    8 x 40 000 points) on a side stream, and the network + score fusion + NMS on the main
    stream.  ``__call__`` runs them back to back for one batch; ``stream(batches)`` overlaps the
    index chain of batch t+1 with the network of batch t (throughput mode)."""

    def __init__(self, model, batch, num_points, feat_dim=4, device=None):
        from .backbone import clone_index_tree, index_tree_tensors
        self.model = model.eval()
        device = device or next(model.parameters()).device
        self.device = device
        cfg = model.test_cfg if isinstance(model.test_cfg, dict) else vars(model.test_cfg)
        self.pts = torch.zeros(batch, num_points, feat_dim, device=device)
        # plausible coordinates for the warm-up passes (degenerate all-zero clouds are legal
        # but exercise nothing)
        self.pts[..., :3].uniform_(-3.0, 3.0)
        self.pts_next = self.pts.clone()
        self.side = torch.cuda.Stream(device)
        self.main = torch.cuda.current_stream(device)
        self.ready, self.taken = torch.cuda.Event(), torch.cuda.Event()
        with torch.no_grad():
            self.idx_next = model.backbone.sample_and_group_indices(self.pts_next)
            self.idx_cur = clone_index_tree(self.idx_next)
        self._flat_next, self._flat_cur = index_tree_tensors(self.idx_next), \
            index_tree_tensors(self.idx_cur)

        def network():
            x = model.extract_feat(self.pts, self.idx_cur)
            preds = model.bbox_head(x, cfg['sample_mod'])
            return model.bbox_head.detect_tensors(self.pts, preds,
                                                  cfg.get('use_iou_for_nms', True))

        def index_chain():
            fresh = model.backbone.sample_and_group_indices(self.pts_next)
            torch._foreach_copy_(self._flat_next, index_tree_tensors(fresh))
        self.side.wait_stream(self.main)
        with torch.no_grad(), torch.cuda.stream(self.side):
            for _ in range(2):
                index_chain()
                network()
        self.main.wait_stream(self.side)
        torch.cuda.synchronize(device)
        self.graph, self.g_idx = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.tensors = network()
        with torch.no_grad(), torch.cuda.graph(self.g_idx, stream=self.side):
            index_chain()
        self.taken.record(self.main)

    def _load_next(self, pts):
        """Stage a batch and start its index chain on the side stream."""
        pts = torch.stack(pts) if isinstance(pts, (list, tuple)) else pts
        assert pts.shape == self.pts.shape, (tuple(pts.shape), tuple(self.pts.shape))
        # the side stream must see (a) whatever produced `pts` on the caller's stream and
        # (b) the consumption of the previous staging (`taken`, recorded on the main stream)
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        self.side.wait_event(self.taken)
        pts.record_stream(self.side)   # its memory must outlive the side-stream copy
        with torch.cuda.stream(self.side):
            self.pts_next.copy_(pts, non_blocking=True)
            self.g_idx.replay()
            self.ready.record(self.side)

    def _rotate(self):
        """Make the staged batch current (its indices are complete)."""
        self.main.wait_event(self.ready)
        torch._foreach_copy_(self._flat_cur + [self.pts], self._flat_next + [self.pts_next])
        self.taken.record(self.main)

    def _results(self, img_metas):
        with torch.no_grad():   # one device->host transfer, selection on the host
            out = self.model.bbox_head.boxes_from_tensors(tuple(t.cpu() for t in self.tensors),
                                                          img_metas)
        return [bbox3d2result(b, s, l) for b, s, l in out]

    def __call__(self, points, img_metas=None):
        self._load_next(points)
        self._rotate()
        self.graph.replay()
        return self._results(img_metas)

    def _launch(self):
        """Replay the network graph and start the device->host copy of its five small output
        tensors into pinned buffers (two sets, alternating); -> (host tensors, event)."""
        if getattr(self, '_host', None) is None:
            self._host = [[torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in self.tensors]
                          for _ in range(2)]
            self._done = [torch.cuda.Event(), torch.cuda.Event()]
            self._flip = 0
        self.graph.replay()
        host, ev = self._host[self._flip], self._done[self._flip]
        for h, t in zip(host, self.tensors):
            h.copy_(t, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        self._flip ^= 1
        return host, ev

    def _host_results(self, launched, img_metas):
        host, ev = launched
        ev.synchronize()            # waits for THIS batch's outputs only, not for the next graph
        with torch.no_grad():
            out = self.model.bbox_head.boxes_from_tensors(tuple(host), img_metas)
        return [bbox3d2result(b, s, l) for b, s, l in out]

    def stream(self, batches, img_metas=None):
        """Generator over the results of consecutive batches.  Three things overlap: the index
        chain of batch t+1 (side-stream graph), the network of batch t (main graph) and the
        host-side selection of batch t-1 (its outputs were copied to pinned memory before the
        next graph was queued)."""
        it = iter(batches)
        cur = next(it, None)
        if cur is None:
            return
        self._load_next(cur)
        pending = None
        while cur is not None:
            nxt = next(it, None)
            self._rotate()
            if nxt is not None:
                self._load_next(nxt)
            launched = self._launch()
            if pending is not None:
                yield self._host_results(pending, img_metas)
            pending = launched
            cur = nxt
        yield self._host_results(pending, img_metas)


def build_nesie_votenet(cfg=None):
    cfg = copy.deepcopy(cfg or nesie_votenet_scannet_cfg())
    return VoteNet(cfg['backbone'], cfg['bbox_head'], cfg['train_cfg'], cfg['test_cfg'],
                   head_type=cfg.get('head_type', 'NesieHead'))


def build_saqe_votenet(cfg=None):
    return build_nesie_votenet(cfg or saqe_votenet_scannet_cfg())
