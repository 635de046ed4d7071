"""Hough voting (``mmdet3d/models/model_utils/vote_module.py:9-180``)."""
import torch
from torch import nn

from ..mmdet3d_ops import ConvModule, PointwiseConv1d
from .losses import build_loss


class VoteModule(nn.Module):
    """Seeds -> votes: Conv1d stack -> (offset, residual feature) per seed."""

    def __init__(self, in_channels, vote_per_seed=1, gt_per_seed=3, num_points=-1,
                 conv_channels=(16, 16), conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='BN1d'), act_cfg=dict(type='ReLU'), norm_feats=True,
                 with_res_feat=True, vote_xyz_range=None, vote_loss=None):
        super().__init__()
        self.in_channels = in_channels
        self.vote_per_seed = vote_per_seed
        self.gt_per_seed = gt_per_seed
        self.num_points = num_points
        self.norm_feats = norm_feats
        self.with_res_feat = with_res_feat
        self.vote_xyz_range = vote_xyz_range
        if vote_loss is not None:
            self.vote_loss = build_loss(vote_loss)
        prev_channels = in_channels
        vote_conv_list = []
        for k in range(len(conv_channels)):
            vote_conv_list.append(
                ConvModule(prev_channels, conv_channels[k], 1, padding=0, conv_cfg=conv_cfg,
                           norm_cfg=norm_cfg, act_cfg=act_cfg, bias=True, inplace=True))
            prev_channels = conv_channels[k]
        self.vote_conv = nn.Sequential(*vote_conv_list)
        out_channel = ((3 + in_channels) if with_res_feat else 3) * self.vote_per_seed
        self.conv_out = PointwiseConv1d(prev_channels, out_channel, 1)

    def forward(self, seed_points, seed_feats):
        """(B,N,3),(B,C,N) -> vote_points (B,M,3), vote_feats (B,C,M), offset (B,3,M)."""
        if self.num_points != -1:
            assert self.num_points < seed_points.shape[1]
            seed_points = seed_points[:, :self.num_points]
            seed_feats = seed_feats[..., :self.num_points]
        batch_size, feat_channels, num_seed = seed_feats.shape
        num_vote = num_seed * self.vote_per_seed
        x = self.vote_conv(seed_feats)
        votes = self.conv_out(x)
        votes = votes.transpose(2, 1).view(batch_size, num_seed, self.vote_per_seed, -1)
        offset = votes[:, :, :, 0:3]
        if self.vote_xyz_range is not None:
            limited = [offset[..., a].clamp(min=-self.vote_xyz_range[a],
                                            max=self.vote_xyz_range[a])
                       for a in range(len(self.vote_xyz_range))]
            vote_points = (seed_points.unsqueeze(2) + torch.stack(limited, -1)).contiguous()
        else:
            vote_points = (seed_points.unsqueeze(2) + offset).contiguous()
        vote_points = vote_points.view(batch_size, num_vote, 3)
        offset = offset.reshape(batch_size, num_vote, 3).transpose(2, 1)
        if self.with_res_feat:
            res_feats = votes[:, :, :, 3:]
            vote_feats = (seed_feats.transpose(2, 1).unsqueeze(2) + res_feats).contiguous()
            vote_feats = vote_feats.view(batch_size, num_vote,
                                         feat_channels).transpose(2, 1).contiguous()
            if self.norm_feats:
                features_norm = torch.norm(vote_feats, p=2, dim=1)
                vote_feats = vote_feats.div(features_norm.unsqueeze(1))
        else:
            vote_feats = seed_feats
        return vote_points, vote_feats, offset

    def get_loss(self, seed_points, vote_points, seed_indices, vote_targets_mask,
                 vote_targets):
        """Chamfer-L1 of each vote to its seed's (up to 3) GT votes, min over the 3,
        mask-weighted sum (:149-180)."""
        batch_size, num_seed = seed_points.shape[:2]
        seed_gt_votes_mask = torch.gather(vote_targets_mask, 1, seed_indices).float()
        seed_indices_expand = seed_indices.unsqueeze(-1).repeat(1, 1, 3 * self.gt_per_seed)
        seed_gt_votes = torch.gather(vote_targets, 1, seed_indices_expand)
        seed_gt_votes = seed_gt_votes + seed_points.repeat(1, 1, self.gt_per_seed)
        weight = seed_gt_votes_mask / (torch.sum(seed_gt_votes_mask) + 1e-6)
        distance = self.vote_loss(
            vote_points.view(batch_size * num_seed, -1, 3),
            seed_gt_votes.view(batch_size * num_seed, -1, 3),
            dst_weight=weight.view(batch_size * num_seed, 1))[1]
        return torch.sum(torch.min(distance, dim=1)[0])
