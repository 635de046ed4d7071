"""Debug aid: run bench.other_workload for a sequence of workloads in ONE process (does an earlier
workload slow a later one down?).  usage: python tools/debug/other_workloads_seq.py semi:8:0,saqe:16:1
(workload:batch:gate); a leading `head` item runs the headline step first (graph replays), `head+eager`
also its un-captured steps."""
import gc
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402

dev = torch.device('cuda:0')
for item in sys.argv[1].split(','):
    if item.startswith('head'):
        cfg = bench.nesie_votenet_scannet_cfg()
        model, step, bucket = bench.build_step(dev, 8, 1000, 1e-3, 0.01, graph=True)
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        if 'eager' in item:
            for _ in range(5):
                step.eager()
            torch.cuda.synchronize()
        if 'keep' not in item:
            del model, step, bucket
            gc.collect()
            torch.cuda.empty_cache()
        print(item, 'done', torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
        continue
    w, b, g = item.split(':')
    r = bench.other_workload(dev, w, int(b), 5, 2, gate=bool(int(g)))
    print(item, r.get('ms_per_step'), torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
