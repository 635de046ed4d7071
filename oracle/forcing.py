"""Test infrastructure (like the rest of oracle/): replaying one leg's discrete sampling decisions
in the other legs of a parity comparison.  Never imported by the product path."""
import torch


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


class ForcedTaps:
    """The same for the 3-NN taps of the quality head's grid points (``SidePooling.fused_taps`` /
    ``_blend_taps``: idx, inverse-distance weights, centre-relative xyz).  The grid points are
    functions of PREDICTED boxes; a 1e-7 difference in a coordinate flips the third neighbour
    of a grid point between two seeds at (to fp32) the same distance, that grid point then
    blends another seed's features, and one MiniPointNet's activations and gradients move by
    percents (both outcomes legitimate).  The first leg records its neighbour indices; the other
    legs use them and rebuild the weights of the flipped taps from their own coordinates.
    (The 3-NN kernel itself is compared bit for bit on identical inputs in test_kernels_gpu.py.)

    Installed on the CLASS (a closure stored on an instance would keep pointing at the original
    module after ``copy.deepcopy``); a module takes part only while it carries ``_taps_key``."""

    book = {}        # key -> [idx of call 0, call 1, ...]
    stats = {}       # key -> [flipped grid points, compared grid points] of the replaying legs
    installed = False

    @classmethod
    def install(cls):
        if cls.installed:
            return
        from nesie_amd.votenet.side_pooling import SidePooling
        for name, pick in (('fused_taps', lambda a: (a[0], a[1])),      # (origin_xyz, center, size, heading, which)
                           ('_blend_taps', lambda a: (a[0], a[2]))):    # (origin_xyz, whole_grid, center)
            setattr(SidePooling, name, cls._wrap(getattr(SidePooling, name), pick))
        cls.installed = True

    @classmethod
    def _wrap(cls, inner, pick):
        def forced(self, *args):
            idx, weight, rel = inner(self, *args)
            key = getattr(self, '_taps_key', None)
            if key is None:
                return idx, weight, rel
            rec = cls.book[key]
            call = self._taps_calls
            self._taps_calls = call + 1
            if call >= len(rec):
                rec.append(idx.detach().cpu())
                return idx, weight, rel
            want = rec[call].to(idx.device)
            diff = (want != idx).any(-1)
            st = cls.stats[key]
            st[1] += diff.numel()
            nflip = int(diff.sum())
            if nflip == 0:
                return idx, weight, rel
            st[0] += nflip
            origin_xyz, center = pick(args)
            K = center.shape[1]
            G = rel.shape[1] // K
            world = rel + center.repeat_interleave(G, dim=1)                       # (B, n, 3)
            nb = torch.gather(origin_xyz.unsqueeze(1).expand(-1, want.shape[1], -1, -1), 2,
                              want.long().unsqueeze(-1).expand(-1, -1, -1, 3))     # (B, n, 3, 3)
            d = ((nb - world.unsqueeze(2)) ** 2).sum(-1).sqrt()
            w = 1.0 / (d + 1e-8)
            w = w / w.sum(-1, keepdim=True)
            weight = torch.where(diff.unsqueeze(-1), w.to(weight.dtype), weight).contiguous()
            return want.contiguous(), weight, rel
        return forced


def force_grid_taps(model, key='default'):
    """Every copy of ``model`` made after this call replays the grid taps of the first leg that
    runs; -> [flipped, compared] counters of the replaying legs."""
    ForcedTaps.install()
    ForcedTaps.book[key] = []
    ForcedTaps.stats[key] = [0, 0]
    pooling = model.bbox_head.grid_conv
    pooling._taps_key, pooling._taps_calls = key, 0
    return ForcedTaps.stats[key]
