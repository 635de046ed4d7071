"""The VoteNet / Nesie detector path on top of ``nesie_amd.mmdet3d_ops``:
PointNet++ backbone, Hough voting + vote aggregation, the side-aware quality
head and the per-side localisation-uncertainty losses."""
from .backbone import PointNet2SASSG
from .boxes import DepthInstance3DBoxes
from .detector import (VoteNet, build_nesie_votenet, build_saqe_votenet,
                       nesie_votenet_scannet_cfg, saqe_votenet_scannet_cfg)
from .nesie_head import NesieHead
from .semi import (AugMeta, EMATeacher, VoteNetNesie, VoteNetSAQE, build_nesie_votenet_semi,
                   build_saqe_votenet_semi)
from .side_pooling import MiniPointNet, SidePooling
from .vote_module import VoteModule

__all__ = ['PointNet2SASSG', 'DepthInstance3DBoxes', 'VoteNet', 'build_nesie_votenet',
           'nesie_votenet_scannet_cfg', 'NesieHead', 'MiniPointNet', 'SidePooling',
           'VoteModule', 'AugMeta', 'EMATeacher', 'VoteNetNesie',
           'build_nesie_votenet_semi', 'build_saqe_votenet', 'VoteNetSAQE',
           'build_saqe_votenet_semi',
           'saqe_votenet_scannet_cfg']
