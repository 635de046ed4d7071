"""PointNet++ SSG backbone (``mmdet3d/models/backbones/pointnet2_sa_ssg.py:11-142``,
``base_pointnet.py:20-37``): 4 set-abstraction + 2 feature-propagation layers."""
import torch
from torch import nn

from ..mmdet3d_ops import PointFPModule, build_sa_module


class PointNet2SASSG(nn.Module):
    def __init__(self, in_channels, num_points=(2048, 1024, 512, 256),
                 radius=(0.2, 0.4, 0.8, 1.2), num_samples=(64, 32, 16, 16),
                 sa_channels=((64, 64, 128), (128, 128, 256), (128, 128, 256),
                              (128, 128, 256)),
                 fp_channels=((256, 256), (256, 256)), norm_cfg=dict(type='BN2d'),
                 sa_cfg=dict(type='PointSAModule', pool_mod='max', use_xyz=True,
                             normalize_xyz=True)):
        super().__init__()
        self.num_sa = len(sa_channels)
        self.num_fp = len(fp_channels)
        assert len(num_points) == len(radius) == len(num_samples) == len(sa_channels)
        assert len(sa_channels) >= len(fp_channels)
        self.SA_modules = nn.ModuleList()
        sa_in_channel = in_channels - 3
        skip_channel_list = [sa_in_channel]
        for sa_index in range(self.num_sa):
            cur_sa_mlps = [sa_in_channel] + list(sa_channels[sa_index])
            sa_out_channel = cur_sa_mlps[-1]
            self.SA_modules.append(
                build_sa_module(num_point=num_points[sa_index], radius=radius[sa_index],
                                num_sample=num_samples[sa_index], mlp_channels=cur_sa_mlps,
                                norm_cfg=norm_cfg, cfg=sa_cfg))
            skip_channel_list.append(sa_out_channel)
            sa_in_channel = sa_out_channel
        self.FP_modules = nn.ModuleList()
        fp_source_channel = skip_channel_list.pop()
        fp_target_channel = skip_channel_list.pop()
        for fp_index in range(len(fp_channels)):
            cur_fp_mlps = [fp_source_channel + fp_target_channel] + list(fp_channels[fp_index])
            self.FP_modules.append(PointFPModule(mlp_channels=cur_fp_mlps))
            if fp_index != len(fp_channels) - 1:
                fp_source_channel = cur_fp_mlps[-1]
                fp_target_channel = skip_channel_list.pop()

    @staticmethod
    def _split_point_feats(points):
        xyz = points[..., 0:3].contiguous()
        features = points[..., 3:].transpose(1, 2).contiguous() if points.size(-1) > 3 else None
        return xyz, features

    def sample_and_group_indices(self, points):
        """FPS + ball-query indices of all SA layers: they depend on the input coordinates
        only, so a loop may compute them for the next batch while this one trains."""
        xyz = points[..., 0:3].contiguous()
        out, sa_xyz = [], [xyz]
        for sa in self.SA_modules:
            pre = sa.sample_and_group_indices(xyz)
            out.append(pre)
            xyz = pre['new_xyz']
            sa_xyz.append(xyz)
        # feature-propagation taps (3-NN between consecutive levels) ride on the first entry
        out[0]['fp_taps'] = [self.FP_modules[i].interpolation_taps(
            sa_xyz[self.num_sa - i - 1], sa_xyz[self.num_sa - i]) for i in range(self.num_fp)]
        return out

    def forward(self, points, precomputed=None):
        """(B,N,3+C) -> dict of fp_xyz / fp_features / fp_indices (+ the sa_* lists).
        ``precomputed`` = sample_and_group_indices(points) evaluated earlier (optional)."""
        xyz, features = self._split_point_feats(points)
        batch, num_points = xyz.shape[:2]
        indices = torch.arange(num_points, device=xyz.device).unsqueeze(0).repeat(batch, 1).long()
        sa_xyz, sa_features, sa_indices = [xyz], [features], [indices]
        for i in range(self.num_sa):
            cur_xyz, cur_features, cur_indices = self.SA_modules[i](
                sa_xyz[i], sa_features[i],
                precomputed=None if precomputed is None else precomputed[i])
            sa_xyz.append(cur_xyz)
            sa_features.append(cur_features)
            sa_indices.append(torch.gather(sa_indices[-1], 1, cur_indices.long()))
        fp_xyz, fp_features, fp_indices = [sa_xyz[-1]], [sa_features[-1]], [sa_indices[-1]]
        for i in range(self.num_fp):
            taps = None if precomputed is None else precomputed[0].get('fp_taps')
            fp_features.append(self.FP_modules[i](
                sa_xyz[self.num_sa - i - 1], sa_xyz[self.num_sa - i],
                sa_features[self.num_sa - i - 1], fp_features[-1],
                taps=None if taps is None else taps[i]))
            fp_xyz.append(sa_xyz[self.num_sa - i - 1])
            fp_indices.append(sa_indices[self.num_sa - i - 1])
        return dict(fp_xyz=fp_xyz, fp_features=fp_features, fp_indices=fp_indices,
                    sa_xyz=sa_xyz, sa_features=sa_features, sa_indices=sa_indices)


def index_tree_tensors(tree):
    """Every tensor of a ``sample_and_group_indices`` result, in a fixed order (for copying one
    such result into the static buffers of another: the pipelined loops of bench.py and of
    ``GraphedSimpleTest``)."""
    out = []
    for d in tree:
        out += [d['indices'], d['new_xyz']] + list(d['group_idx'])
        for csr in d.get('group_csr', ()):
            out += list(csr or ())
        for idx_, w_, csr in d.get('fp_taps', ()):
            out += [idx_, w_] + list(csr or ())
    return out


def clone_index_tree(tree):
    return [dict(indices=d['indices'].clone(), new_xyz=d['new_xyz'].clone(),
                 group_idx=[t.clone() for t in d['group_idx']],
                 group_csr=[None if csr is None else tuple(t.clone() for t in csr)
                            for csr in d.get('group_csr', ())],
                 **({'fp_taps': [(i_.clone(), w_.clone(),
                                  None if c_ is None else tuple(t.clone() for t in c_))
                                 for i_, w_, c_ in d['fp_taps']]} if 'fp_taps' in d else {}))
            for d in tree]
