"""A reduced Nesie-VoteNet (same structure, fewer points) the CPU oracle path can run
forward+backward in seconds."""
import copy

import torch

from nesie_amd.votenet import build_nesie_votenet, nesie_votenet_scannet_cfg
from nesie_amd.scenes import make_batch


def small_cfg():
    cfg = copy.deepcopy(nesie_votenet_scannet_cfg())
    cfg['backbone'].update(num_points=(512, 256, 128, 64), num_samples=(16, 16, 8, 8))
    cfg['bbox_head']['vote_aggregation_cfg'].update(num_point=32, num_sample=8)
    cfg['bbox_head']['grid_conv_cfg'].update(num_proposal=32)
    return cfg


def small_model(seed=0):
    torch.manual_seed(seed)
    return build_nesie_votenet(small_cfg())


def small_batch(seed=7, batch=2, n=4096):
    pts, boxes, labels = make_batch(seed, batch, num_points=n)
    return pts, boxes, labels


def fixed_noise(batch, k, seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, k, 3, generator=g), torch.randn(batch, k, 3, generator=g)


def train_step_losses(model, pts, boxes, labels):
    """One forward+backward; returns (losses dict of floats, {name: grad})."""
    from nesie_amd.votenet.nesie_head import GTBatch
    for p in model.parameters():
        p.grad = None
    gt = GTBatch.collate(boxes, labels, pts.device)
    losses = model.forward_train(pts, None, gt, None)
    total = model.parse_losses(losses)
    total.backward()
    return ({k: v.detach().cpu() for k, v in losses.items()},
            {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()
             if p.grad is not None})


from oracle.forcing import ForcedSampler, force_vote_sampling  # noqa: E402,F401
