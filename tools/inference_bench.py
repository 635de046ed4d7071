"""Test-path timing at the full config (8 scenes x 40 000 points): eval-mode forward,
get_bboxes (point count + NMS + selection), and the reference-style per-scene python NMS
(the literal loop of box3d_nms.py:129-176 written with torch ops on the device) beside it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nesie_amd.scenes import make_batch
from nesie_amd.votenet import build_nesie_votenet
from nesie_amd import post_processing


def loop_nms(boxes, scores, classes, thresh):
    area = (boxes[:, 3] - boxes[:, 0]) * (boxes[:, 4] - boxes[:, 1]) * (boxes[:, 5] - boxes[:, 2])
    order = torch.argsort(scores)
    zero = boxes.new_zeros(1)
    pick = []
    while order.shape[0] != 0:
        i = order[-1]
        pick.append(i)
        rest = order[:-1]
        lo = torch.max(boxes[i, :3], boxes[rest, :3])
        hi = torch.min(boxes[i, 3:], boxes[rest, 3:])
        d = torch.max(zero, hi - lo)
        inter = d[:, 0] * d[:, 1] * d[:, 2]
        iou = inter / (area[i] + area[rest] - inter) * (classes[i] == classes[rest]).float()
        order = rest[torch.nonzero(iou <= thresh, as_tuple=False).flatten()]
    return torch.stack(pick)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = build_nesie_votenet().to(dev)
    pts, _, _ = make_batch(1, 8)
    pts = pts.to(dev)
    model.train()
    with torch.no_grad():
        model.bbox_head(model.extract_feat(pts), 'vote')   # move the running statistics
    model.eval()
    with torch.no_grad():
        preds = model.bbox_head(model.extract_feat(pts), 'seed')
        fwd = timed(lambda: model.bbox_head(model.extract_feat(pts), 'seed'))
        post = timed(lambda: model.bbox_head.get_bboxes(pts, preds, None))
        whole = timed(lambda: model.simple_test(pts, None))
    graphed = model.graphed_simple_test(8, 40000)
    ref = model.simple_test(pts, None)
    got = graphed(pts)
    same = all(torch.equal(a['labels_3d'], b['labels_3d']) and
               torch.allclose(a['scores_3d'], b['scores_3d'], rtol=1e-4, atol=1e-6)
               for a, b in zip(ref, got))
    gwhole = timed(lambda: graphed(pts))
    batches = [pts.roll(i, 0) for i in range(12)]
    for _ in graphed.stream(batches[:3]):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_out = sum(1 for _ in graphed.stream(batches))
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) / n_out * 1e3
    first = next(iter(graphed.stream([pts])))
    same_stream = all(torch.equal(a['labels_3d'], b['labels_3d']) for a, b in zip(ref, first))
    g = torch.Generator().manual_seed(1)
    c = torch.rand(8, 256, 3, generator=g) * 4
    h = 0.2 + torch.rand(8, 256, 3, generator=g) * 0.5
    boxes = torch.cat([c - h, c + h], -1).to(dev)
    scores, classes = torch.rand(8, 256, generator=g).to(dev), torch.randint(0, 18, (8, 256), generator=g).to(dev)
    ours = timed(lambda: post_processing.batched_aligned_3d_nms(boxes, scores, classes, 0.25), 50)
    loop = timed(lambda: [loop_nms(boxes[b], scores[b], classes[b], 0.25) for b in range(8)], 3)
    with torch.no_grad():
        one = pts[:1].contiguous()
        f1 = timed(lambda: model.bbox_head(model.extract_feat(one), 'seed'))
        s1 = timed(lambda: model.simple_test(one, None))
    print(f'BASELINE configs[1] (batch 1, forward only): backbone + head {f1:6.2f} ms, simple_test {s1:6.2f} ms')
    print(f'eval forward (8 x 40k)      {fwd:8.2f} ms')
    print(f'get_bboxes                   {post:8.2f} ms')
    print(f'simple_test                  {whole:8.2f} ms  = {8e3 / whole:.0f} scenes/s')
    print(f'simple_test, hipGraph        {gwhole:8.2f} ms  = {8e3 / gwhole:.0f} scenes/s  (same detections as eager: {same})')
    print(f'simple_test, pipelined graphs{piped:8.2f} ms  = {8e3 / piped:.0f} scenes/s  (index chain of batch t+1 under the network of batch t; same detections: {same_stream})')
    model.test_cfg['skip_jitter'] = True
    lean = model.graphed_simple_test(8, 40000)
    first = next(iter(lean.stream([pts])))
    same_lean = all(torch.equal(a['labels_3d'], b['labels_3d']) and
                    torch.allclose(a['scores_3d'], b['scores_3d'], rtol=1e-4, atol=1e-6)
                    for a, b in zip(ref, first))
    for _ in lean.stream(batches[:3]):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_out = sum(1 for _ in lean.stream(batches))
    torch.cuda.synchronize()
    lean_ms = (time.perf_counter() - t0) / n_out * 1e3
    model.test_cfg['skip_jitter'] = False
    print(f'  + test_cfg.skip_jitter     {lean_ms:8.2f} ms  = {8e3 / lean_ms:.0f} scenes/s  (quality head on the original proposals only; same detections: {same_lean})')
    print(f'aligned_3d_nms, 8 x 256      {ours:8.3f} ms (one launch)  vs  {loop:8.1f} ms python loop of torch ops')


if __name__ == '__main__':
    main()
