"""Debug aid: checksums of the backbone's per-level outputs and head predictions of one training-mode
forward at B scenes (compare two runs with different NESIE_PW_REV_* settings)."""
import sys
sys.path.insert(0, '.')
import torch
from nesie_amd.scenes import make_batch
from nesie_amd.votenet import build_nesie_votenet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
model = build_nesie_votenet().cuda().train()
pts, boxes, labels = make_batch(4242, B, 40000)
with torch.no_grad():
    out = model.backbone(pts.cuda())
    for i, f in enumerate(out['sa_features']):
        if f is not None:
            print('sa', i, tuple(f.shape), '%.6f %.6f' % (float(f.double().sum()), float(f.double().abs().sum())))
    for i, f in enumerate(out['fp_features']):
        print('fp', i, tuple(f.shape), '%.6f %.6f' % (float(f.double().sum()), float(f.double().abs().sum())))
