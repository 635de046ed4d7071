"""Mirror of ``mmdet3d/ops/furthest_point_sample`` (reference
furthest_point_sample.py:8-77, points_sampler.py:11-161, utils.py:4-34).

Same names, argument meaning and return types; the native call goes to
libnesie_hip.so through :mod:`nesie_amd.kernels` instead of the reference's
``furthest_point_sample_ext``.
"""
from typing import List

import torch
from torch import nn
from torch.autograd import Function

from ..kernels import backend_for


class FurthestPointSampling(Function):
    """D-FPS: (B,N,3) points -> (B,num_points) int32 indices (reference :15-35)."""

    @staticmethod
    def forward(ctx, points_xyz: torch.Tensor, num_points: int) -> torch.Tensor:
        assert points_xyz.is_contiguous()
        B, N = points_xyz.size()[:2]
        output = points_xyz.new_empty((B, num_points), dtype=torch.int32)
        temp = points_xyz.new_full((B, N), 1e10, dtype=torch.float32)
        backend_for(points_xyz).furthest_point_sampling_wrapper(
            B, N, num_points, points_xyz, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


class FurthestPointSamplingWithDist(Function):
    """F-FPS on a precomputed (B,N,N) distance matrix (reference :46-73)."""

    @staticmethod
    def forward(ctx, points_dist: torch.Tensor, num_points: int) -> torch.Tensor:
        assert points_dist.is_contiguous()
        B, N, _ = points_dist.size()
        output = points_dist.new_zeros([B, num_points], dtype=torch.int32)
        temp = points_dist.new_zeros([B, N]).fill_(1e10)
        backend_for(points_dist).furthest_point_sampling_with_dist_wrapper(
            B, N, num_points, points_dist, temp, output)
        ctx.mark_non_differentiable(output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply
furthest_point_sample_with_dist = FurthestPointSamplingWithDist.apply


def calc_square_dist(point_feat_a, point_feat_b, norm=True):
    """(B,N,C),(B,M,C) -> (B,N,M) squared distances (reference utils.py:4-34)."""
    num_channel = point_feat_a.shape[-1]
    a_square = torch.sum(point_feat_a.unsqueeze(dim=2).pow(2), dim=-1)
    b_square = torch.sum(point_feat_b.unsqueeze(dim=1).pow(2), dim=-1)
    coor = torch.matmul(point_feat_a, point_feat_b.transpose(1, 2))
    dist = a_square + b_square - 2 * coor
    if norm:
        dist = torch.sqrt(dist) / num_channel
    return dist


class DFPS_Sampler(nn.Module):
    """Euclidean FPS (reference points_sampler.py:104-116)."""

    def forward(self, points, features, npoint):
        return furthest_point_sample(points.contiguous(), npoint)


class FFPS_Sampler(nn.Module):
    """Feature-distance FPS (reference points_sampler.py:119-137)."""

    def forward(self, points, features, npoint):
        assert features is not None, 'feature input to FFPS_Sampler should not be None'
        features_for_fps = torch.cat([points, features.transpose(1, 2)], dim=2)
        features_dist = calc_square_dist(features_for_fps, features_for_fps, norm=False)
        return furthest_point_sample_with_dist(features_dist.contiguous(), npoint)


class FS_Sampler(nn.Module):
    """F-FPS and D-FPS concatenated (reference points_sampler.py:140-161)."""

    def forward(self, points, features, npoint):
        assert features is not None, 'feature input to FS_Sampler should not be None'
        features_for_fps = torch.cat([points, features.transpose(1, 2)], dim=2)
        features_dist = calc_square_dist(features_for_fps, features_for_fps, norm=False)
        fps_idx_ffps = furthest_point_sample_with_dist(features_dist.contiguous(), npoint)
        fps_idx_dfps = furthest_point_sample(points.contiguous(), npoint)
        return torch.cat([fps_idx_ffps, fps_idx_dfps], dim=1)


def get_sampler_type(sampler_type):
    table = {'D-FPS': DFPS_Sampler, 'F-FPS': FFPS_Sampler, 'FS': FS_Sampler}
    if sampler_type not in table:
        raise ValueError('Only "sampler_type" of "D-FPS", "F-FPS", or "FS"'
                         f' are supported, got {sampler_type}')
    return table[sampler_type]


class Points_Sampler(nn.Module):
    """Range-sliced list of samplers (reference points_sampler.py:34-101)."""

    def __init__(self, num_point: List[int], fps_mod_list: List[str] = ['D-FPS'],
                 fps_sample_range_list: List[int] = [-1]):
        super().__init__()
        assert len(num_point) == len(fps_mod_list) == len(fps_sample_range_list)
        self.num_point = num_point
        self.fps_sample_range_list = fps_sample_range_list
        self.samplers = nn.ModuleList([get_sampler_type(m)() for m in fps_mod_list])
        self.fp16_enabled = False

    def forward(self, points_xyz, features):
        points_xyz = points_xyz.float()  # @force_fp32 in the reference (:65)
        if features is not None:
            features = features.float()
        picked, start = [], 0
        for stop, sampler, npoint in zip(self.fps_sample_range_list, self.samplers, self.num_point):
            assert stop < points_xyz.shape[1]
            end = None if stop == -1 else stop
            part_xyz = points_xyz[:, start:end]
            part_feats = features[:, :, start:end] if features is not None else None
            idx = sampler(part_xyz.contiguous(), part_feats, npoint)
            picked.append(idx + start if start else idx)      # (no launch for the usual offset 0)
            start += stop
        return picked[0] if len(picked) == 1 else torch.cat(picked, dim=1)
