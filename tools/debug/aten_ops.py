"""Debug aid: which ATen operators does one eager training step still call, and from where?
Runs bench.build_step(graph=False) for two steps and records, under a TorchDispatchMode, every
ATen op of the second step with its first argument's shape and the innermost nesie_amd / bench
frame that called it.  usage: python tools/debug/aten_ops.py [workload]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, '.')
import bench  # noqa: E402

SKIP = ('aten.view', 'aten.detach', 'aten.alias', 'aten._unsafe_view', 'aten.t.', 'aten.transpose', 'aten.permute',
        'aten.reshape', 'aten.expand', 'aten.slice', 'aten.select', 'aten.unsqueeze', 'aten.squeeze', 'aten.as_strided',
        'aten.empty', 'aten.new_empty', 'aten.unbind', 'aten.split', 'aten.narrow', 'aten.is_', 'aten.sym_', 'aten.stride',
        'aten.size', 'aten.unflatten', 'aten.flatten', 'aten.chunk', 'aten._local_scalar', 'aten.lift_fresh', 'aten.set_',
        'aten.record_stream', 'aten.resize_')


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            shape = next((tuple(a.shape) for a in args if torch.is_tensor(a)), None)
            where = '?'
            for fr in reversed(traceback.extract_stack(limit=40)):
                f = fr.filename
                if ('nesie_amd' in f or f.endswith('bench.py')) and 'debug' not in f:
                    where = f'{os.path.basename(f)}:{fr.lineno} {fr.name}'
                    break
            self.seen[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else 'pretrain'
    dev = torch.device('cuda:0')
    model, step, bucket = bench.build_step(dev, 8, 0, 1e-3, 0.01, graph=False, workload=workload)
    step()
    torch.cuda.synchronize()
    # the weight-independent index chain runs on the side stream in graph mode: keep it out of the list
    pre = None
    if workload == 'pretrain' and '--with-index-chain' not in sys.argv:
        inp = step.inputs
        pre = dict(indices=model.backbone.sample_and_group_indices(inp['points']),
                   vote_targets=tuple(model.bbox_head.vote_targets_of(inp['points'], inp['gt'])))
    spy = Spy()
    with spy:
        step(pre) if pre is not None else step()
    torch.cuda.synchronize()
    by_where = collections.Counter()
    for (name, where, shape), n in spy.seen.items():
        by_where[where] += n
    print('total recorded ops', sum(spy.seen.values()))
    print('---- by caller')
    for w, n in by_where.most_common(60):
        print(f'{n:5d}  {w}')
    print('---- by (op, caller, shape)')
    for (name, where, shape), n in spy.seen.most_common(150):
        print(f'{n:4d}  {name:38s} {where:50s} {shape}')


if __name__ == '__main__':
    main()
