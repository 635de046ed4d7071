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
