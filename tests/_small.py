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


class ForcedSampler(torch.nn.Module):
    """Stands in for ``bbox_head.vote_aggregation.points_sampler`` in parity tests.  The
    furthest-point sampling of the VOTES is a chain of arg-max decisions over coordinates the
    network computed: a 1e-6 difference between two arithmetic paths can flip one pick and, from
    there on, the whole proposal set (both outcomes are legitimate).  The reference leg records
    its picks, the other legs replay them, so the comparison is between the same proposals.
    (The sampling kernel itself is compared bit-for-bit on identical inputs elsewhere.)"""

    book = {}   # key -> [picks of call 0, call 1, ...]; class-level: deep copies of the model share it

    def __init__(self, inner, key):
        super().__init__()
        self.inner, self.key, self.calls, self.agreed = inner, key, 0, []
        ForcedSampler.book[key] = []

    def forward(self, xyz, features):
        own = self.inner(xyz, features)
        rec = ForcedSampler.book[self.key]
        call, self.calls = self.calls, self.calls + 1
        if call >= len(rec):
            rec.append(own.detach().cpu())
            return own
        self.agreed.append(bool(torch.equal(own.detach().cpu(), rec[call])))
        return rec[call].to(own.device)


def force_vote_sampling(model, key='default'):
    """Every copy of ``model`` made after this call shares the picks of the first leg that runs."""
    agg = model.bbox_head.vote_aggregation
    inner = agg.points_sampler.inner if isinstance(agg.points_sampler, ForcedSampler) \
        else agg.points_sampler
    agg.points_sampler = ForcedSampler(inner, key)
    return agg.points_sampler
