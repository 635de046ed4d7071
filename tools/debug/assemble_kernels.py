"""Debug aid: the kernels one call of ResidentScenes.assemble_batch (+ refresh_noise) launches --
what `bench.py --resident-input` adds to the side stream per step -- with device time per kernel."""
import collections
import sys

import torch

sys.path.insert(0, '.')
from nesie_amd import input_pipeline  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    B, N, R = 8, 40000, 64
    from nesie_amd.scenes import make_scene
    scenes = input_pipeline.ResidentScenes(dev)
    for i in range(R):
        p_, b_, l_ = make_scene(7 * i, 50000)
        centre = torch.cat([b_[:, :2], b_[:, 2:3] + b_[:, 5:6] * 0.5, b_[:, 3:6]], 1)
        scenes.add_scene(p_[:, :3].numpy(), None, centre.numpy(), l_.numpy())
    scenes.finalize()
    ids = torch.arange(B, device=dev)
    noise = scenes.new_noise(B, N)

    def once():
        scenes.refresh_noise(noise)
        return scenes.assemble_batch(ids, num_points=N, noise=noise)
    once()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            once()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            rows[e.name][0] += 1
            rows[e.name][1] += e.device_time
    print(f'assembly: {sum(v[0] for v in rows.values()) / 5:.0f} launches, '
          f'{sum(v[1] for v in rows.values()) / 5 / 1e3:.3f} ms of kernel time per call')
    for name, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f'{n / 5:6.1f} {t / 5:9.1f} us  {name[:110]}')


if __name__ == '__main__':
    main()
