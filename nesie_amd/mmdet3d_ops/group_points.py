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
        grad_out_data = grad_out.data.contiguous()
        backend = backend_for(grad_out_data)
        csr = _scatter_index(backend, idx, N)
        if csr is not None:     # fixed-order scatter (group_points_cuda.cu:10-31 adds with atomicAdd);
            #                     it WRITES every point: no zero fill
            grad_features = grad_out.new_empty((B, C, N))
            backend.group_points_backward_csr(grad_out_data, csr[0], csr[1], grad_features.data)
        else:
            grad_features = grad_out.new_zeros((B, C, N))
            backend.group_points_backward(B, C, N, npoint, nsample, grad_out_data, idx,
                                          grad_features.data)
        return grad_features, None


grouping_operation = GroupingOperation.apply
group_points = grouping_operation  # the name mmdet3d.ops re-exports


def _scatter_index(backend, idx, n):
    """The inverted index a backward builds for itself when nobody handed one over and the back
    end runs its scatters in a fixed order (``HipKernels.DETERMINISTIC``, the default)."""
    build = getattr(backend, 'scatter_index', None)
    return None if build is None else build(idx, n)


def inverted_index(idx, n):
    """idx (B, M, ns) int32 source-point indices -> (order, sources), both (B, M*ns) int32 = the
    grouped columns sorted by source point and that point for each, or None when the back end
    has no index builder.  Pure index work: depends on coordinates only, so it is built ahead
    of the step with the ball query."""
    backend = backend_for(idx)
    if not hasattr(backend, 'inverted_index'):
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
        csr = ctx.saved_tensors[1:] if ctx.has_csr else _scatter_index(backend, idx, n)
        if csr and hasattr(backend, 'query_and_group_backward_csr'):
            grad_features = grad_out.new_empty(grad_out.shape[0], c, n)      # (written in full)
            backend.query_and_group_backward_csr(grad_out, idx.shape, csr[0], csr[1], grad_features)
        else:
            grad_features = grad_out.new_zeros(grad_out.shape[0], c, n)
            backend.query_and_group_backward(grad_out, idx, grad_features)
        return None, None, grad_features, None, None, None


class SampleQueryGroupCat(Function):
    """Sample + group over NETWORK-COMPUTED coordinates as one autograd node: centres =
    xyz[sample] and cat[(xyz[idx] - centre) (/ radius), features[idx]] for a ball query around
    those centres -- the grouping of the vote aggregation module, whose points are the predicted
    votes (nesie_head.py:243 -> point_sa_module.py:122-131, 160-211; group_points.py:98-128).
    The reference reaches it through transpose / gather_points / transpose / ball_query /
    group_points x 2 / sub / div / cat, and its backward through two atomicAdd scatters; here the
    forward is four launches (centre gather, ball query, inverted index, one-pass grouping) and the
    backward two: the feature scatter through the inverted index and ONE kernel for the coordinate
    gradient (direct term, centre term, gradient of the returned centres), both in a fixed order.
    -> (centres (B, M, 3), grouped (B, 3+C, M, ns), ball indices)."""

    @staticmethod
    def forward(ctx, points_xyz, features, sample, min_radius, max_radius, nsample, normalize):
        backend = backend_for(points_xyz)
        xyz = points_xyz.contiguous()
        features = features.contiguous()
        b, n = xyz.shape[:2]
        centres = backend.gather_rows3(xyz, sample)
        idx = xyz.new_zeros((b, sample.shape[1], nsample), dtype=torch.int32)
        backend.ball_query_wrapper(b, n, sample.shape[1], min_radius, max_radius, nsample, centres, xyz, idx)
        order, sources = backend.inverted_index(idx, n)
        radius = float(max_radius) if normalize else 0.0
        out = xyz.new_empty(b, 3 + features.shape[1], sample.shape[1], nsample)
        backend.query_and_group_forward(xyz, centres, features, idx, radius, out)
        ctx.save_for_backward(idx, order, sources, sample)
        ctx.dims = (features.shape[1], n, radius)
        ctx.mark_non_differentiable(idx)
        return centres, out, idx

    @staticmethod
    def backward(ctx, d_centres, grad_out, _):
        idx, order, sources, sample = ctx.saved_tensors
        c, n, radius = ctx.dims
        backend = backend_for(idx)
        if grad_out is None:
            grad_out = idx.new_zeros((idx.shape[0], 3 + c, idx.shape[1], idx.shape[2]), dtype=torch.float32)
        grad_out = grad_out.contiguous()
        d_feat = d_xyz = None
        if ctx.needs_input_grad[1]:
            d_feat = grad_out.new_empty(grad_out.shape[0], c, n)             # (written in full)
            backend.query_and_group_backward_csr(grad_out, idx.shape, order, sources, d_feat)
        if ctx.needs_input_grad[0]:
            d_xyz = backend.query_and_group_backward_xyz(
                grad_out, radius, order, sources, sample,
                None if d_centres is None else d_centres.contiguous(), n)
        return d_xyz, d_feat, None, None, None, None, None


# Tests flip this to obtain the literal op-by-op grouping over differentiable coordinates.
SAMPLE_GROUP_FUSED = True


def sample_query_group_supported(points_xyz, features, grouper):
    """True when ``SampleQueryGroupCat`` serves a grouper over differentiable coordinates."""
    backend = backend_for(points_xyz)
    return (SAMPLE_GROUP_FUSED and getattr(backend, 'name', '') == 'hip' and features is not None
            and grouper.use_xyz
            and not grouper.return_grouped_xyz and not grouper.return_grouped_idx
            and points_xyz.dtype == torch.float32 and features.dtype == torch.float32
            and points_xyz.shape[1] <= 8192 and torch.is_grad_enabled()      # (qg_xyz_bwd: one thread per point)
            and (points_xyz.requires_grad or features.requires_grad))


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
