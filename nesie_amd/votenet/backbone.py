"""PointNet++ single-scale-grouping backbone: 4 set-abstraction + 2 feature-propagation levels
(behaviour of ``mmdet3d/models/backbones/pointnet2_sa_ssg.py:11-142``, ``base_pointnet.py:20-37``)."""
import torch
from torch import nn

from ..mmdet3d_ops import PointFPModule, build_sa_module


class PointNet2SASSG(nn.Module):
    """Encoder-decoder over point sets.  Level 0 is the input cloud, level i the centres the
    i-th set-abstraction module sampled; ``widths[i]`` is the feature width at level i.  The
    decoder walks back from the coarsest level: step j interpolates the carried features of
    level ``depth - j`` onto level ``depth - j - 1`` and joins that level's own features.
    Constructor keywords, module names (``SA_modules`` / ``FP_modules``: state-dict keys) and the
    returned dict follow pointnet2_sa_ssg.py:11-142."""

    def __init__(self, in_channels, num_points=(2048, 1024, 512, 256),
                 radius=(0.2, 0.4, 0.8, 1.2), num_samples=(64, 32, 16, 16),
                 sa_channels=((64, 64, 128), (128, 128, 256), (128, 128, 256),
                              (128, 128, 256)),
                 fp_channels=((256, 256), (256, 256)), norm_cfg=dict(type='BN2d'),
                 sa_cfg=dict(type='PointSAModule', pool_mod='max', use_xyz=True,
                             normalize_xyz=True)):
        super().__init__()
        depth = len(sa_channels)
        if not (len(num_points) == len(radius) == len(num_samples) == depth):
            raise AssertionError('one entry per set-abstraction level is required')
        if len(fp_channels) > depth:
            raise AssertionError('more feature-propagation steps than levels')
        self.num_sa, self.num_fp = depth, len(fp_channels)
        widths = [in_channels - 3]
        self.SA_modules = nn.ModuleList()
        for count, reach, group, mlp in zip(num_points, radius, num_samples, sa_channels):
            self.SA_modules.append(build_sa_module(
                num_point=count, radius=reach, num_sample=group,
                mlp_channels=[widths[-1], *mlp], norm_cfg=norm_cfg, cfg=sa_cfg))
            widths.append(mlp[-1])
        self.FP_modules = nn.ModuleList()
        carried = widths[depth]
        for j, mlp in enumerate(fp_channels):
            self.FP_modules.append(PointFPModule(mlp_channels=[carried + widths[depth - 1 - j], *mlp]))
            carried = mlp[-1]

    @staticmethod
    def _split_point_feats(points):
        """(B, N, 3 + C) -> coordinates (B, N, 3), channel-major features (B, C, N) or None."""
        extra = points.shape[-1] - 3
        return (points[..., :3].contiguous(),
                points[..., 3:].transpose(1, 2).contiguous() if extra > 0 else None)

    def sample_and_group_indices(self, points):
        """FPS + ball-query indices of all SA layers: they depend on the input coordinates
        only, so a loop may compute them for the next batch while this one trains."""
        level_xyz = [points[..., :3].contiguous()]
        plan, cum = [], None
        for sa in self.SA_modules:
            plan.append(sa.sample_and_group_indices(level_xyz[-1]))
            level_xyz.append(plan[-1]['new_xyz'])
            # positions of this level's centres in the INPUT cloud (index bookkeeping only)
            picked = plan[-1]['indices'].long()
            cum = picked if cum is None else cum.gather(1, picked)
            plan[-1]['input_indices'] = cum
        # feature-propagation taps (3-NN from each finer level into the coarser one below it)
        # ride on the first entry
        plan[0]['fp_taps'] = [fp.interpolation_taps(level_xyz[self.num_sa - 1 - j],
                                                    level_xyz[self.num_sa - j])
                              for j, fp in enumerate(self.FP_modules)]
        return plan

    def forward(self, points, precomputed=None):
        """(B, N, 3 + C) -> dict of fp_xyz / fp_features / fp_indices (+ the sa_* lists); index
        lists hold positions in the INPUT cloud.  ``precomputed`` = sample_and_group_indices(
        points) evaluated earlier (optional)."""
        xyz, feats = self._split_point_feats(points)
        B, N = xyz.shape[:2]
        sa_xyz, sa_features = [xyz], [feats]
        sa_indices = [torch.arange(N, device=xyz.device).expand(B, N)]
        for i, sa in enumerate(self.SA_modules):
            centres, pooled, picked = sa(sa_xyz[-1], sa_features[-1],
                                         precomputed=precomputed[i] if precomputed is not None else None)
            sa_xyz.append(centres)
            sa_features.append(pooled)
            pre_idx = precomputed[i].get('input_indices') if precomputed is not None else None
            sa_indices.append(pre_idx if pre_idx is not None
                              else sa_indices[-1].gather(1, picked.long()))
        taps = precomputed[0].get('fp_taps') if precomputed is not None else None
        fp_xyz, fp_features, fp_indices = [sa_xyz[-1]], [sa_features[-1]], [sa_indices[-1]]
        for j, fp in enumerate(self.FP_modules):
            fine = self.num_sa - 1 - j
            fp_features.append(fp(sa_xyz[fine], sa_xyz[fine + 1], sa_features[fine], fp_features[-1],
                                  taps=taps[j] if taps is not None else None))
            fp_xyz.append(sa_xyz[fine])
            fp_indices.append(sa_indices[fine])
        return dict(fp_xyz=fp_xyz, fp_features=fp_features, fp_indices=fp_indices,
                    sa_xyz=sa_xyz, sa_features=sa_features, sa_indices=sa_indices)


def index_tree_tensors(tree):
    """Every tensor of a ``sample_and_group_indices`` result, in a fixed order (for copying one
    such result into the static buffers of another: the pipelined loops of bench.py and of
    ``GraphedSimpleTest``)."""
    out = []
    for d in tree:
        out += [d['indices'], d['new_xyz']] + list(d['group_idx'])
        if 'input_indices' in d:
            out.append(d['input_indices'])
        for csr in d.get('group_csr', ()):
            out += list(csr or ())
        for idx_, w_, csr in d.get('fp_taps', ()):
            out += [idx_, w_] + list(csr or ())
    return out


def clone_index_tree(tree):
    return [dict(indices=d['indices'].clone(), new_xyz=d['new_xyz'].clone(),
                 group_idx=[t.clone() for t in d['group_idx']],
                 **({'input_indices': d['input_indices'].clone()} if 'input_indices' in d else {}),
                 group_csr=[None if csr is None else tuple(t.clone() for t in csr)
                            for csr in d.get('group_csr', ())],
                 **({'fp_taps': [(i_.clone(), w_.clone(),
                                  None if c_ is None else tuple(t.clone() for t in c_))
                                 for i_, w_, c_ in d['fp_taps']]} if 'fp_taps' in d else {}))
            for d in tree]


def index_tree_like(tree, tensors):
    """``tree`` with its tensors replaced, in ``index_tree_tensors`` order, by ``tensors``."""
    it = iter(tensors)
    out = []
    for d in tree:
        e = dict(indices=next(it), new_xyz=next(it), group_idx=[next(it) for _ in d['group_idx']])
        if 'input_indices' in d:
            e['input_indices'] = next(it)
        e.update(group_csr=[None if csr is None else tuple(next(it) for _ in csr)
                            for csr in d.get('group_csr', ())])
        if 'fp_taps' in d:
            e['fp_taps'] = [(next(it), next(it), None if c_ is None else tuple(next(it) for _ in c_))
                            for _, _, c_ in d['fp_taps']]
        out.append(e)
    return out


def pack_tensors(tensors):
    """Copies of ``tensors`` as views into ONE byte buffer (16-byte aligned slots): handing a
    whole set of index tensors from one static buffer set to another is then a single
    device-to-device copy instead of one launch per tensor.  -> (views, buffer)."""
    sizes = [t.numel() * t.element_size() for t in tensors]
    slots = [(n + 15) // 16 * 16 for n in sizes]
    arena = torch.empty(sum(slots), dtype=torch.uint8, device=tensors[0].device)
    views, off = [], 0
    for t, n, slot in zip(tensors, sizes, slots):
        v = arena[off:off + n].view(t.dtype).view(t.shape)
        v.copy_(t)
        views.append(v)
        off += slot
    return views, arena


def copy_tensors(dst, src):
    """``dst[i].copy_(src[i])`` for lists of mixed dtypes as one multi-tensor launch per dtype:
    ``torch._foreach_copy_`` over a mixed list falls back to one device-to-device copy per tensor
    (32 blit launches for one index set)."""
    groups = {}
    for d, s in zip(dst, src):
        groups.setdefault((d.dtype, s.dtype), ([], []))
        groups[(d.dtype, s.dtype)][0].append(d)
        groups[(d.dtype, s.dtype)][1].append(s)
    for ds, ss in groups.values():
        torch._foreach_copy_(ds, ss)

