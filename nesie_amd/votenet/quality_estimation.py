"""SAQE quality head (``mmdet3d/models/dense_heads/quelity_estimation_module.py``): the
SidePooling variant with a 3x3x3 grid, every face sampled on three planes (the face and
the face shifted by -/+10 % of its offset: 27 points), 128-wide MiniPointNets, and ONE
global head over the six concatenated face features that emits per-class IoU scores,
per-class rotation scores and a 2-way objectness (``:54-76, 142-167, 323-344``)."""
import torch
from torch import nn

from ..mmdet3d_ops.norm import FusedBNReLU1d
from ..mmdet3d_ops.pointnet_modules import PointwiseConv1d
from ..kernels import backend_for
from .side_pooling import (MiniPointNet, SidePooling, batched_heads, grouped_mini_pointnets,
                           heads_batchable, mini_pointnets_groupable)


class QualityEstimation(SidePooling):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr_path,
                 num_proposal, sampling, seed_feat_dim=256, query_feats='seed',
                 iou_class_depend=True):
        nn.Module.__init__(self)
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = None
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.query_feats = query_feats
        self.iou_class_depend = iou_class_depend
        self.reg_topk = 4
        self.grid_size = g = 3
        self.left_mask = [i // g * g * g + i % g for i in range(g * g)]
        self.right_mask = [i // g * g * g + i % g + g * (g - 1) for i in range(g * g)]
        self.iou_size = num_class if iou_class_depend else 1
        before, head = [], []
        for _ in range(6):
            before.append(MiniPointNet(seed_feat_dim + 3, 128, hide_dim=128))
            head.append(nn.Sequential(PointwiseConv1d(128 + 33 + 4 + 1, 128, 1),
                                      FusedBNReLU1d(128), nn.Identity(),
                                      PointwiseConv1d(128, self.iou_size, 1)))
        head.append(nn.Sequential(
            PointwiseConv1d((128 + 33 + 4 + 1) * 6, 512, 1), FusedBNReLU1d(512), nn.Identity(),
            PointwiseConv1d(512, 256, 1), FusedBNReLU1d(256), nn.Identity(),
            PointwiseConv1d(256, self.iou_size * 2 + 2, 1)))
        self.mlps_before = nn.ModuleList(before)
        self.mlps_head = nn.ModuleList(head)
        # face selections (front, back, top, down, left, right) and, per face, which box-frame
        # axis its +-10 % plane offset acts on (x for front/back, z for top/down, y for sides)
        face_idx = (list(range(0, g * g)) + list(range(g ** 3 - g * g, g ** 3))
                    + list(range(g - 1, g ** 3, g)) + list(range(0, g ** 3, g))
                    + self.left_mask + self.right_mask)
        self.register_buffer('_face_idx', torch.tensor(face_idx, dtype=torch.long),
                             persistent=False)
        axis = torch.zeros(6, 3)
        for f, a in enumerate([0, 0, 2, 2, 1, 1]):
            axis[f, a] = 0.1
        self.register_buffer('_plane_axis', axis, persistent=False)
        self._register_grid_tables(face_idx, plane=axis)

    def grid_for_side(self, whole_grid, center, heading):
        B, K = center.shape[:2]
        g2 = self.grid_size * self.grid_size
        faces = torch.index_select(whole_grid, 2, self._face_idx).view(B, K, 6, g2, 3)
        zero = faces * self._plane_axis.view(1, 1, 6, 1, 3)   # face * 0.1 on its own axis only
        planes = torch.cat([faces - zero, faces, faces + zero], dim=3)  # (B,K,6,3*g2,3)
        return self._to_scene(planes.reshape(B, K, -1, 3), center, heading)

    def forward(self, center, size, heading, end_points, prefix=''):
        B, K = size.shape[:2]
        origin_xyz, origin_features = self.extract_features(end_points)
        fused = backend_for(origin_xyz).name == 'hip'
        side_nets = list(self.mlps_before[:6])
        if fused:
            side_c0, side_normed, side_stats = self.first_conv_through_blend(
                side_nets, origin_xyz, origin_features, None, center,
                taps=self.fused_taps(origin_xyz, center, size, heading, 'side'), with_norm=self.fuse_first_norm)
        else:
            whole_grid = self.generate_grid(size)
            side_grid = self.grid_for_side(whole_grid, center, heading).view(B, -1, 3).contiguous()
            side_feats = self.grid_features(origin_xyz, origin_features, side_grid, center, segs=6)
        dist_feature = self.dist_feature(end_points, prefix,
                                         copies=K // end_points[f'{prefix}bbox_probs'].shape[-1])
        if fused and mini_pointnets_groupable(side_nets, side_c0):
            pooled = grouped_mini_pointnets(side_nets, side_c0, normed=side_normed,
                                            c0_stats=side_stats)
        elif fused:
            key = 'a0' if side_normed else 'conv0_out'
            pooled = torch.stack([side_nets[i](**{key: side_c0[:, i]}) for i in range(6)], 1)
        else:
            pooled = torch.stack([side_nets[i](side_feats[:, i]) for i in range(6)], 1)
        heads = list(self.mlps_head[:6])
        x = torch.cat([pooled, dist_feature.transpose(0, 1)], dim=2)      # (B,6,166,2K)
        if heads_batchable(heads, x[:, 0]):
            side_scores = batched_heads(heads, x).transpose(0, 1).contiguous()
        else:
            side_scores = torch.stack([self.mlps_head[i](x[:, i]) for i in range(6)], 0)
        all_features = x.flatten(1, 2)                   # == cat of the six (B,166,2K) inputs
        end_points[f'{prefix}side_scores'] = side_scores
        global_scores = self.mlps_head[6](all_features).transpose(2, 1)
        n = self.iou_size
        end_points[f'{prefix}iou_scores'] = global_scores[..., :n]
        end_points[f'{prefix}rotate_scores'] = global_scores[..., n:n * 2]
        end_points[f'{prefix}R_obj_scores'] = global_scores[..., n * 2:]
        return end_points
