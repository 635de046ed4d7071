"""Prediction layers (``mmdet3d/models/dense_heads/reliable_conv_bbox_module.py:9-177``):
shared Conv1d stack, then class / side-distribution / heading 1x1 convs."""
import torch
from torch import nn

from ..mmdet3d_ops import ConvModule, PointwiseConv1d


class ReliableConvBboxHead(nn.Module):
    def __init__(self, in_channels=0, shared_conv_channels=(), cls_conv_channels=(),
                 num_cls_out_channels=0, bbox_conv_channels=(), num_bbox_out_channels=0,
                 heading_conv_channels=(), num_heading_out_channels=0, reg_max=16,
                 conv_cfg=dict(type='Conv1d'), norm_cfg=dict(type='BN1d'),
                 act_cfg=dict(type='ReLU'), bias='auto'):
        super().__init__()
        assert in_channels > 0 and num_cls_out_channels > 0
        assert num_bbox_out_channels > 0 and num_heading_out_channels > 0
        self.shared_conv_channels = shared_conv_channels
        self.cls_conv_channels = cls_conv_channels
        self.bbox_conv_channels = bbox_conv_channels
        self.heading_conv_channels = heading_conv_channels
        self.conv_cfg, self.norm_cfg, self.act_cfg, self.bias = conv_cfg, norm_cfg, act_cfg, bias
        out_channels = in_channels
        if len(shared_conv_channels) > 0:
            self.shared_convs = self._add_conv_branch(in_channels, shared_conv_channels)
            out_channels = shared_conv_channels[-1]
        prev = out_channels
        if len(cls_conv_channels) > 0:
            self.cls_convs = self._add_conv_branch(prev, cls_conv_channels)
            prev = cls_conv_channels[-1]
        self.conv_cls = PointwiseConv1d(prev, num_cls_out_channels, 1)
        prev = out_channels
        if len(bbox_conv_channels) > 0:
            self.bbox_convs = self._add_conv_branch(prev, bbox_conv_channels)
            prev = bbox_conv_channels[-1]
        self.conv_bbox = PointwiseConv1d(prev, num_bbox_out_channels, 1)
        prev = out_channels
        if len(heading_conv_channels) > 0:
            self.heading_convs = self._add_conv_branch(
                prev, heading_conv_channels, dict(type='GN', num_groups=reg_max))
            prev = heading_conv_channels[-1]
        self.conv_heading = PointwiseConv1d(prev, num_heading_out_channels, 1)

    def _add_conv_branch(self, in_channels, conv_channels, norm_cfg=None):
        spec = [in_channels] + list(conv_channels)
        layers = nn.Sequential()
        for i in range(len(spec) - 1):
            layers.add_module(
                f'layer{i}',
                ConvModule(spec[i], spec[i + 1], kernel_size=1, padding=0,
                           conv_cfg=self.conv_cfg, norm_cfg=norm_cfg or self.norm_cfg,
                           act_cfg=self.act_cfg, bias=self.bias, inplace=True))
        return layers

    def _fused(self, feats):
        """The shared trunk as ONE fused chain on the layer kernel (``fused_mlp.Stack1dFn``); the class,
        side-distribution and heading convolutions are three single-layer launches of the same kernel
        on its output (reliable_conv_bbox_module.py:144-177).  None when the chain is not served (evaluation,
        CPU checker, configured branch convs).
        (Round 3 also ran the three output convolutions as one stacked 220-row layer inside the
        chain.  Values and gradients matched, but a step captured in a hipGraph with that layer
        aborted on replay while every launch of it replays fine on its own
        (tools/debug/graph_ops.py, graph_bisect.py); not understood, so not shipped.  Round 4: the
        220-row layer -- forward, input gradient, weight gradient -- between canaries, eager and in a
        replayed graph of its own, touches nothing outside its buffers and equals float64
        (tests/test_pwconv_gpu.py::test_ragged_220_row_output_layer_stays_inside_its_buffers): the
        kernels' ragged-row guards are not the cause.)"""
        from ..kernels import backend_for
        from ..mmdet3d_ops import fused_mlp
        if (len(self.cls_conv_channels) or len(self.bbox_conv_channels) or len(self.heading_conv_channels)
                or len(self.shared_conv_channels) == 0):
            return None
        trunk = list(self.shared_convs)
        shapes = [(m.conv.in_channels, m.conv.out_channels) for m in trunk]
        norms = [m.norm for m in trunk]
        if not all(m.act_fused for m in trunk) or \
                not fused_mlp.stack1d_supported(backend_for(feats), feats, shapes, norms, which=fused_mlp.PRED):
            return None
        x = fused_mlp.stack1d(feats, [m.conv for m in trunk], norms)

        def out_conv(conv):     # one bias-carrying layer of the same kernel each (OUT bit of NESIE_STACK1D)
            if fused_mlp.stack1d_supported(backend_for(x), x, [(conv.in_channels, conv.out_channels)], [None],
                                           which=fused_mlp.PRED_OUT):
                return fused_mlp.stack1d(x, [conv], [None])
            return conv(x)
        return out_conv(self.conv_cls), torch.cat((out_conv(self.conv_bbox), out_conv(self.conv_heading)), dim=1)

    def forward(self, feats):
        fused = self._fused(feats)
        if fused is not None:
            return fused
        x = self.shared_convs(feats) if len(self.shared_conv_channels) > 0 else feats
        x_cls = self.cls_convs(x) if len(self.cls_conv_channels) > 0 else x
        cls_score = self.conv_cls(x_cls)
        x_bbox = self.bbox_convs(x) if len(self.bbox_conv_channels) > 0 else x
        bbox_pred_bbox = self.conv_bbox(x_bbox)
        x_heading = self.heading_convs(x) if len(self.heading_conv_channels) > 0 else x
        bbox_pred_heading = self.conv_heading(x_heading)
        return cls_score, torch.cat((bbox_pred_bbox, bbox_pred_heading), dim=1)
