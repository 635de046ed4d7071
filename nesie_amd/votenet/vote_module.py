"""Hough voting: every seed point predicts where its object's centre is.

Behaviour follows ``mmdet3d/models/model_utils/vote_module.py:9-180`` (constructor keywords,
output shapes and orderings, loss definition); the data flow is channel-major throughout, so
the (B, C, N) activations are never transposed to (B, N, C) and back.
"""
import torch
from torch import nn

from ..mmdet3d_ops import ConvModule, PointwiseConv1d
from .losses import build_loss


# Tests flip this to obtain the op-by-op tail of VoteModule.forward on the device.
FUSED_FINISH = True


class VoteFinishFn(torch.autograd.Function):
    """(raw (B, 3+C, N), seed_points (B, N, 3), seed_feats (B, C, N)) -> (vote_points, vote_feats):
    seed + offset, and the L2-normalised sum of seed and residual features (vote_module.py:106-147 for
    one vote per seed).  Gradients: d_raw from one kernel, d_seed_feats = its feature rows (a view),
    d_seed_points = the incoming gradient of the vote points."""

    @staticmethod
    def forward(ctx, raw, seed_points, seed_feats, normalise):
        from ..kernels import backend_for
        vp, vf, inv = backend_for(raw).vote_finish_forward(raw.contiguous(), seed_points.contiguous(),
                                                           seed_feats.contiguous(), normalise)
        ctx.normalise = normalise
        ctx.save_for_backward(vf, inv)
        return vp, vf

    @staticmethod
    def backward(ctx, g_points, g_feats):
        from ..kernels import backend_for
        vf, inv = ctx.saved_tensors
        d_raw = backend_for(vf).vote_finish_backward(
            None if g_feats is None else g_feats.contiguous(),
            None if g_points is None else g_points.contiguous(), vf, inv, ctx.normalise)
        return d_raw, g_points, d_raw[:, 3:], None


class VoteModule(nn.Module):
    """``vote_per_seed`` votes per seed: a 1x1-conv stack, then one linear map whose output
    channels are laid out vote-major, ``[dx, dy, dz, residual feature (C)]`` per vote."""

    def __init__(self, in_channels, vote_per_seed=1, gt_per_seed=3, num_points=-1,
                 conv_channels=(16, 16), conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='BN1d'), act_cfg=dict(type='ReLU'), norm_feats=True,
                 with_res_feat=True, vote_xyz_range=None, vote_loss=None):
        super().__init__()
        self.in_channels, self.vote_per_seed, self.gt_per_seed = in_channels, vote_per_seed, gt_per_seed
        self.num_points, self.norm_feats, self.with_res_feat = num_points, norm_feats, with_res_feat
        self.vote_xyz_range = vote_xyz_range
        if vote_loss is not None:
            self.vote_loss = build_loss(vote_loss)
        widths = [in_channels, *conv_channels]
        self.vote_conv = nn.Sequential(*[
            ConvModule(cin, cout, 1, padding=0, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                       act_cfg=act_cfg, bias=True, inplace=True)
            for cin, cout in zip(widths, widths[1:])])
        self.per_vote = 3 + in_channels if with_res_feat else 3
        self.conv_out = PointwiseConv1d(widths[-1], self.per_vote * vote_per_seed, 1)
        from .head_loss import new_ticket
        self.register_buffer('_loss_ticket', new_ticket(), persistent=False)

    def _vote_stack(self, seed_feats):
        """conv_out(vote_conv(x)): the Hough-voting MLP (vote_module.py:65-74) -- on the HIP back end
        in training one fused chain on the layer kernel (``fused_mlp.Stack1dFn``), else layer by layer."""
        from ..kernels import backend_for
        from ..mmdet3d_ops import fused_mlp
        convs = [m.conv for m in self.vote_conv] + [self.conv_out]
        norms = [m.norm for m in self.vote_conv] + [None]
        shapes = [(c.in_channels, c.out_channels) for c in convs]
        if all(m.act_fused for m in self.vote_conv) and \
                fused_mlp.stack1d_supported(backend_for(seed_feats), seed_feats, shapes, norms, which=fused_mlp.VOTE):
            return fused_mlp.stack1d(seed_feats, convs, norms)
        return self.conv_out(self.vote_conv(seed_feats))

    def forward(self, seed_points, seed_feats):
        """seed_points (B, N, 3), seed_feats (B, C, N) -> vote_points (B, N*V, 3), vote_feats
        (B, C, N*V), offset (B, 3, N*V); vote v of seed n sits at column n*V + v."""
        if self.num_points != -1:
            if not self.num_points < seed_points.shape[1]:
                raise AssertionError('num_points must be smaller than the number of seeds')
            seed_points, seed_feats = seed_points[:, :self.num_points], seed_feats[..., :self.num_points]
        B, C, N = seed_feats.shape
        V = self.vote_per_seed
        raw3 = self._vote_stack(seed_feats)
        from ..kernels import backend_for
        if (V == 1 and self.with_res_feat and self.vote_xyz_range is None and FUSED_FINISH
                and getattr(backend_for(seed_feats), 'name', '') == 'hip' and seed_feats.dtype == torch.float32
                and torch.is_grad_enabled()):
            # everything behind the last convolution as one launch per direction (csrc/vote.hip)
            vote_points, vote_feats = VoteFinishFn.apply(raw3, seed_points, seed_feats, bool(self.norm_feats))
            return vote_points, vote_feats, raw3[:, :3]
        raw = raw3.view(B, V, self.per_vote, N)
        # one split: its backward is one concatenation (two slices cost two zero fills + an add)
        shift, residual = raw.split([3, self.per_vote - 3], dim=2) if self.with_res_feat \
            else (raw, None)                                              # (B, V, 3, N), (B, V, C, N)
        moved = shift
        if self.vote_xyz_range is not None:
            # each axis limited to its own +-range; the returned offset stays unlimited (:117-133)
            cap = shift.new_tensor(list(self.vote_xyz_range)).view(1, 1, -1, 1)
            moved = torch.maximum(torch.minimum(shift, cap), -cap)
        vote_points = (seed_points.unsqueeze(2) + moved.permute(0, 3, 1, 2)).reshape(B, N * V, 3)
        offset = shift.permute(0, 2, 3, 1).reshape(B, 3, N * V)
        if not self.with_res_feat:
            return vote_points, seed_feats, offset
        vote_feats = (seed_feats.unsqueeze(-1) + residual.permute(0, 2, 3, 1)).reshape(B, C, N * V)
        if self.norm_feats:
            vote_feats = vote_feats / vote_feats.norm(p=2, dim=1, keepdim=True)
        return vote_points, vote_feats, offset

    def get_loss(self, seed_points, vote_points, seed_indices, vote_targets_mask, vote_targets):
        """Every seed that lies inside an object is pulled to the nearest of its (up to
        ``gt_per_seed``) ground-truth centres: configured Chamfer term, vote -> target direction,
        closest target per vote, weights = in-object mask / number of in-object seeds
        (:149-180).  vote_targets holds OFFSETS from the input point, 3 per ground truth."""
        from . import head_loss
        if head_loss.vote_loss_usable(self, vote_points, seed_indices, vote_targets_mask, vote_targets):
            return head_loss.VoteLossFn.apply(vote_points, seed_points, seed_indices,
                                              vote_targets_mask, vote_targets,
                                              self.vote_loss.loss_dst_weight, self._loss_ticket)
        B, N = seed_points.shape[:2]
        G = self.gt_per_seed
        inside = vote_targets_mask.gather(1, seed_indices).float()                     # (B, N)
        wanted = vote_targets.gather(1, seed_indices.unsqueeze(-1).expand(-1, -1, 3 * G))
        wanted = (wanted.view(B, N, G, 3) + seed_points.unsqueeze(2)).view(B * N, G, 3)
        share = (inside / (inside.sum() + 1e-6)).view(B * N, 1)
        _, to_target = self.vote_loss(vote_points.view(B * N, -1, 3), wanted, dst_weight=share)[:2]
        return to_target.min(dim=1).values.sum()
