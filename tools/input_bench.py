"""Input side: batches/s of the resident-scene assembly on the GPU beside a numpy restatement
of the reference's per-sample CPU pipeline (load -> height -> align -> sample -> flip -> rot /
scale / trans) on one host core.  usage: python tools/input_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nesie_amd.input_pipeline import ResidentScenes  # noqa: E402
from tests.golden import golden_inputs  # noqa: E402


def cpu_sample(raw6, align, rng, n=40000):
    xyz = raw6[:, :3]
    floor = np.percentile(xyz[:, 2], 0.99)
    pts = np.concatenate([xyz, (xyz[:, 2] - floor)[:, None]], 1).astype(np.float32)
    pts[:, :3] = pts[:, :3] @ align[:3, :3].T.astype(np.float32) + align[:3, 3].astype(np.float32)
    pts = pts[rng.choice(pts.shape[0], n, replace=pts.shape[0] < n)]
    if rng.rand() < 0.5:
        pts[:, 0] = -pts[:, 0]
    if rng.rand() < 0.5:
        pts[:, 1] = -pts[:, 1]
    a = rng.uniform(-0.087266, 0.087266)
    c, s = np.float32(np.cos(a)), np.float32(np.sin(a))
    pts[:, :3] = pts[:, :3] @ np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], np.float32)
    return pts


def main():
    dev = torch.device('cuda:0')
    scenes = ResidentScenes(dev)
    raws = []
    for seed in range(64):
        raw6, align, gt, labels = golden_inputs.raw_scene(100 + seed, 50000, False)
        scenes.add_scene(raw6[:, :3], align, gt, labels)
        raws.append((raw6, align))
    scenes.finalize()
    g = torch.Generator(device=dev).manual_seed(0)
    ids = list(range(8))
    for _ in range(3):
        scenes.assemble(ids, generator=g)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 50
    for r in range(reps):
        scenes.assemble([(r * 8 + i) % 64 for i in range(8)], generator=g)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t) / reps
    choices, xform, _ = scenes.draw_on_device(ids, generator=g)
    out = torch.empty(8, 40000, 4, device=dev)
    from nesie_amd.kernels import backend_for
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        backend_for(out).scene_assemble(scenes.pool, scenes.height, choices, xform, out)
    e.record()
    torch.cuda.synchronize()
    kern = s.elapsed_time(e) / 100
    rng = np.random.RandomState(0)
    t = time.perf_counter()
    for i in range(32):
        cpu_sample(*raws[i], rng)
    cpu = (time.perf_counter() - t) / 32
    print(f'resident set: 64 scenes, {scenes.nbytes() / 1e6:.1f} MB in HBM')
    print(f'assemble(8 scenes x 40000), device draws + kernel + boxes: {gpu * 1e3:.3f} ms = {8 / gpu:.0f} scenes/s (eager)')
    print(f'nesie_scene_assemble alone: {kern * 1e3:.1f} us = {8 * 40000 * 32 / kern / 1e6:.1f} GB/s of algorithmic bytes')
    print(f'numpy restatement of the reference pipeline, one core: {cpu * 1e3:.2f} ms/scene = {1 / cpu:.0f} scenes/s per worker')


if __name__ == '__main__':
    main()
