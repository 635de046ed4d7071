"""Debug aid: the kernels of the weight-independent index chain (what bench.py replays on its side
stream under every step), counted with the torch profiler: name, launches per pass, device time."""
import collections
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    model, step, bucket = bench.build_step(dev, 8, 0, 1e-3, 0.01, graph=False, workload='pretrain')
    inp = step.inputs

    def chain():
        idx = model.backbone.sample_and_group_indices(inp['points'])
        vt = model.bbox_head.vote_targets_of(inp['points'], inp['gt'])
        return idx, vt
    chain()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            chain()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            rows[e.name][0] += 1
            rows[e.name][1] += e.device_time
    total_n = sum(v[0] for v in rows.values()) / 5
    total_t = sum(v[1] for v in rows.values()) / 5
    print(f'index chain: {total_n:.0f} launches, {total_t / 1e3:.3f} ms of kernel time per pass')
    for name, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f'{n / 5:6.1f} {t / 5:9.1f} us  {name[:110]}')


if __name__ == '__main__':
    main()
