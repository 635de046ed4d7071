"""Mirror of ``mmdet3d/ops/group_points/group_points.py`` (QueryAndGroup :11-128,
GroupAll :131-169, GroupingOperation :172-227)."""
from typing import Tuple

import torch
from torch import nn
from torch.autograd import Function

from ..kernels import backend_for
from .ball_query import ball_query


class GroupingOperation(Function):
    """features (B,C,N), indices (B,npoint,nsample) int32 -> (B,C,npoint,nsample)."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous()
        assert indices.is_contiguous()
        B, nfeatures, nsample = indices.size()
        _, C, N = features.size()
        output = features.new_empty((B, C, nfeatures, nsample))
        backend_for(features).group_points_forward(B, C, N, nfeatures, nsample, features,
                                                   indices, output)
        ctx.for_backwards = (indices, N)
        return output

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        idx, N = ctx.for_backwards
        B, C, npoint, nsample = grad_out.size()
        grad_features = grad_out.new_zeros((B, C, N))
        grad_out_data = grad_out.data.contiguous()
        backend_for(grad_out_data).group_points_backward(
            B, C, N, npoint, nsample, grad_out_data, idx, grad_features.data)
        return grad_features, None


grouping_operation = GroupingOperation.apply
group_points = grouping_operation  # the name mmdet3d.ops re-exports


def inverted_index(idx, n):
    """idx (B, M, ns) int32 source-point indices -> (order, sources), both (B, M*ns) int32 = the
    grouped columns sorted by source point and that point for each, or None when the back end
    has no index builder for this size.  Pure index work: depends on coordinates only, so it is built ahead
    of the step with the ball query."""
    backend = backend_for(idx)
    if not hasattr(backend, 'inverted_index') or n > 8192:
        return None
    return backend.inverted_index(idx.contiguous(), n)


class QueryGroupCat(Function):
    """cat[(xyz[idx] - centre) (/ radius), features[idx]] -> (B, 3+C, M, ns) in one pass
    (reference :100-128 transposes, groups twice, subtracts, divides, concatenates); the
    backward scatters channels 3.. of the gradient in place (no slice copy).  Only the features
    carry a gradient (the coordinates are inputs of the step)."""

    @staticmethod
    def forward(ctx, points_xyz, center_xyz, features, idx, radius, csr=None):
        points_xyz, center_xyz = points_xyz.contiguous(), center_xyz.contiguous()
        features = features.contiguous()
        b, n = points_xyz.shape[:2]
        out = points_xyz.new_empty(b, 3 + features.shape[1], idx.shape[1], idx.shape[2])
        backend_for(points_xyz).query_and_group_forward(points_xyz, center_xyz, features, idx,
                                                        radius, out)
        ctx.has_csr = csr is not None
        ctx.save_for_backward(idx, *(csr if ctx.has_csr else ()))
        ctx.cn = (features.shape[1], n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx = ctx.saved_tensors[0]
        c, n = ctx.cn
        grad_out = grad_out.contiguous()
        backend = backend_for(grad_out)
        if ctx.has_csr and hasattr(backend, 'query_and_group_backward_csr'):
            order, offsets = ctx.saved_tensors[1:]
            grad_features = grad_out.new_zeros(grad_out.shape[0], c, n)
            backend.query_and_group_backward_csr(grad_out, idx.shape, order, offsets,
                                                 grad_features)
        else:
            grad_features = grad_out.new_zeros(grad_out.shape[0], c, n)
            backend.query_and_group_backward(grad_out, idx, grad_features)
        return None, None, grad_features, None, None, None


class QueryAndGroup(nn.Module):
    """ball query -> group xyz -> minus centre (-> / radius) -> group features ->
    concat [xyz(3), features(C)]  (reference :64-128).  kNN grouping
    (``max_radius is None``) and ``uniform_sample`` are outside the hot path."""

    def __init__(self, max_radius, sample_num, min_radius=0, use_xyz=True,
                 return_grouped_xyz=False, normalize_xyz=False, uniform_sample=False,
                 return_unique_cnt=False, return_grouped_idx=False):
        super().__init__()
        self.max_radius = max_radius
        self.min_radius = min_radius
        self.sample_num = sample_num
        self.use_xyz = use_xyz
        self.return_grouped_xyz = return_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.uniform_sample = uniform_sample
        self.return_unique_cnt = return_unique_cnt
        self.return_grouped_idx = return_grouped_idx
        if self.return_unique_cnt:
            assert self.uniform_sample, \
                'uniform_sample should be True when returning the count of unique samples'
        if self.max_radius is None:
            raise NotImplementedError(
                'kNN grouping (max_radius=None) is outside the VoteNet/Nesie hot path '
                '(SURVEY.md 2a #30): every shipped config gives a radius')
        if self.uniform_sample:
            raise NotImplementedError('uniform_sample is not used by any shipped config')

    def ball_indices(self, points_xyz, center_xyz):
        return ball_query(self.min_radius, self.max_radius, self.sample_num, points_xyz,
                          center_xyz)

    def forward(self, points_xyz, center_xyz, features=None, idx=None, csr=None):
        """``idx`` (optional, not in the reference): ball-query indices computed ahead of time
        for exactly these points/centres (they depend on coordinates only, never on weights);
        ``csr`` (optional) = ``inverted_index(idx, N)`` for the backward gather-sum."""
        if idx is None:
            idx = ball_query(self.min_radius, self.max_radius, self.sample_num, points_xyz,
                             center_xyz)
        if (features is not None and self.use_xyz and not self.return_grouped_xyz
                and not self.return_grouped_idx and not points_xyz.requires_grad
                and not center_xyz.requires_grad):
            return QueryGroupCat.apply(points_xyz, center_xyz, features, idx,
                                       float(self.max_radius) if self.normalize_xyz else 0.0,
                                       csr)
        xyz_trans = points_xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)  # (B, 3, npoint, sample_num)
        grouped_xyz = grouped_xyz - center_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.max_radius

        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, 'Cannot have not features and not use xyz as a feature!'
            new_features = grouped_xyz

        ret = [new_features]
        if self.return_grouped_xyz:
            ret.append(grouped_xyz)
        if self.return_grouped_idx:
            ret.append(idx)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Group every point into one set (reference :131-169)."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        return grouped_xyz
