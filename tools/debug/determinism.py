"""Debug aid: is one training step reproducible bit for bit?  Runs forward + backward of the same
batch twice from the same weights (eager, B scenes x 40 000 points) and lists the parameters whose
gradients differ between the two runs, grouped by module, plus the loss terms.
usage: python tools/debug/determinism.py [workload] [batch]"""
import collections
import copy
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from nesie_amd.votenet.nesie_head import GTBatch  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'pretrain'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    noise = (torch.randn(batch, 256, 3, generator=g), torch.randn(batch, 256, 3, generator=g))
    model, step, bucket = bench.build_step(dev, batch, 77, 1e-3, 0.01, graph=False, workload=workload, noise=noise)
    inp = step.inputs
    runs = []
    for r in range(3):
        for p in model.parameters():
            p.grad = None
        if workload == 'pretrain':
            # (the index chain -- and with it the inverted indices of the scatter-adds -- ahead of the
            # step, as bench.py's captured step has it)
            pre = dict(indices=model.backbone.sample_and_group_indices(inp['points']),
                       vote_targets=tuple(model.bbox_head.vote_targets_of(inp['points'], inp['gt'])))
            losses = model.forward_train(inp['points'], None, inp['gt'], None, precomputed=pre)
        else:
            model.init_label_state(120, 1081, dev)
            losses = model.forward_train(inp['points_s'], inp['points_t'], inp['gt'], inp['use_label'],
                                         inp['meta_s'], inp['meta_t'], inp['rows'])
        model.parse_losses(losses).backward()
        torch.cuda.synchronize()
        runs.append(({k: v.detach().clone() for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
        # (running statistics move between runs; they do not enter a training-mode step's arithmetic)
    for r in (1, 2):
        l0, g0 = runs[0]
        l1, g1 = runs[r]
        print(f'---- run {r} vs run 0')
        for k in l0:
            if not torch.equal(l0[k], l1[k]):
                print('loss term differs:', k, float(l0[k].sum()), float(l1[k].sum()))
        bad = collections.OrderedDict()
        for n in g0:
            if not torch.equal(g0[n], g1[n]):
                d = float((g0[n] - g1[n]).abs().max() / g0[n].abs().max().clamp_min(1e-30))
                bad[n] = d
        print(f'{len(bad)} of {len(g0)} gradients differ')
        mods = collections.Counter('.'.join(n.split('.')[:3]) for n in bad)
        for m, c in mods.items():
            worst = max(v for n, v in bad.items() if n.startswith(m))
            print(f'  {m}: {c} tensors, worst rel {worst:.2e}')


if __name__ == '__main__':
    main()
