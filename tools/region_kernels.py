"""Which region of one eager training step launches which kernels (count, time, tiny tail).

Runs on the GPU box: wraps modules / methods in record_function labels, exports a chrome
trace, then maps every kernel -> launching runtime call -> enclosing label.  Backward kernels
are attributed to the region of the forward op that created their autograd node (sequence
number), so the table reads "region, fwd|bwd".
usage: python tools/region_kernels.py [out_prefix]
"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import bench
from nesie_amd.votenet import nesie_votenet_scannet_cfg

out_prefix = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/region'
dev = torch.device('cuda:0')
cfg = nesie_votenet_scannet_cfg()['optimizer']
model, step, bucket = bench.build_step(dev, 8, 1000, cfg['lr'], cfg['weight_decay'], graph=False)


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        with record_function(label):
            return f(*a, **k)
    setattr(obj, name, g)


head = model.bbox_head
for i, m in enumerate(model.backbone.SA_modules):
    wrap(m, 'forward', f'R:sa{i + 1}')
for i, m in enumerate(model.backbone.FP_modules):
    wrap(m, 'forward', f'R:fp{i + 1}')
wrap(head.vote_module, 'forward', 'R:vote_module')
wrap(head.vote_aggregation, 'forward', 'R:vote_agg')
wrap(head.conv_pred, 'forward', 'R:conv_pred')
wrap(head, 'side2box', 'R:decode')
wrap(head, 'jitter_bbox_preds', 'R:jitter')
wrap(head, 'get_targets', 'R:targets')
gc = head.grid_conv
wrap(gc, 'generate_grid', 'R:qh_grid')
wrap(gc, 'grid_for_side', 'R:qh_grid')
wrap(gc, 'grid_for_bbox', 'R:qh_grid')
wrap(gc, 'grid_features', 'R:qh_features')
wrap(gc, 'dist_feature', 'R:qh_distfeat')
for i, m in enumerate(gc.mlps_before):
    wrap(m, 'forward', 'R:qh_minipointnet')
for i, m in enumerate(gc.mlps_head):
    wrap(m, 'forward', 'R:qh_scorehead')
import nesie_amd.votenet.side_pooling as _sp
wrap(_sp, 'grouped_mini_pointnets', 'R:qh_minipointnet')
wrap(_sp, 'batched_heads', 'R:qh_scorehead')
wrap(gc, 'first_conv_through_blend', 'R:qh_features')
for nm in ['objectness_loss', 'center_loss', 'surface_loss', 'semantic_loss', 'iou_loss',
           'iou_pred_loss', 'side_loss']:
    if hasattr(head, nm):
        wrap(getattr(head, nm), 'forward', f'R:L_{nm}')
wrap(head.vote_module, 'get_loss', 'R:L_vote')
wrap(head, 'loss', 'R:loss_glue')

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function('R:step_other'):
        step()
    torch.cuda.synchronize()
trace = out_prefix + '_trace.json'
prof.export_chrome_trace(trace)

ev = json.load(open(trace))['traceEvents']
ev = [e for e in ev if e.get('ph') == 'X']
kernels = [e for e in ev if e.get('cat') == 'kernel']
runtime = {e['args'].get('correlation'): e for e in ev
           if e.get('cat') in ('cuda_runtime', 'cuda_driver') and 'args' in e}
labels = [e for e in ev if e.get('cat') == 'user_annotation' and e['name'].startswith('R:')]
cpu_ops = [e for e in ev if e.get('cat') == 'cpu_op']
by_tid = collections.defaultdict(list)
for e in labels:
    by_tid[e['tid']].append(e)


def innermost_label(tid, ts):
    best = None
    for e in by_tid.get(tid, ()):
        if e['ts'] <= ts <= e['ts'] + e['dur']:
            if best is None or e['dur'] < best['dur']:
                best = e
    return best['name'] if best else None


# forward ops by sequence number -> region
seq_region = {}
for e in cpu_ops:
    a = e.get('args', {})
    sn = a.get('Sequence number')
    if sn is None or sn < 0 or e['name'].startswith('autograd::engine'):
        continue
    if 'Fwd thread id' in a and a.get('Fwd thread id', 0) != 0:
        continue  # this is a backward-side op
    lab = innermost_label(e['tid'], e['ts'])
    if lab and sn not in seq_region:
        seq_region[sn] = lab
bwd_nodes = collections.defaultdict(list)
for e in cpu_ops:
    if e['name'].startswith('autograd::engine::evaluate_function'):
        bwd_nodes[e['tid']].append(e)

stat = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
names = collections.defaultdict(lambda: collections.Counter())
times = collections.defaultdict(lambda: collections.Counter())
for k in kernels:
    r = runtime.get(k['args'].get('correlation'))
    region = 'unattributed'
    if r is not None:
        lab = innermost_label(r['tid'], r['ts'])
        if lab and lab != 'R:step_other':
            region = lab[2:] + ' fwd'
        else:
            node = None
            for e in bwd_nodes.get(r['tid'], ()):
                if e['ts'] <= r['ts'] <= e['ts'] + e['dur']:
                    node = e
                    break
            if node is not None:
                sn = node.get('args', {}).get('Sequence number')
                region = (seq_region.get(sn, 'R:?')[2:] + ' bwd')
            else:
                region = 'step_other (optimizer / zero / clip)'
    s = stat[region]
    s[0] += 1
    s[1] += k['dur']
    if k['dur'] < 6:
        s[2] += 1
        s[3] += k['dur']
    key = k['name'][:150] + ' grid=' + str(k['args'].get('grid'))
    names[region][key] += 1
    times[region][key] += k['dur']

with open(out_prefix + '_table.txt', 'w') as f:
    tot = [0, 0.0, 0, 0.0]
    f.write('%-40s %6s %9s %6s %9s\n' % ('region', 'n', 'ms', 'n<6us', 'ms<6us'))
    for region, s in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        f.write('%-40s %6d %9.3f %6d %9.3f\n' % (region, s[0], s[1] / 1e3, s[2], s[3] / 1e3))
        for i in range(4):
            tot[i] += s[i]
    f.write('%-40s %6d %9.3f %6d %9.3f\n' % ('TOTAL', tot[0], tot[1] / 1e3, tot[2], tot[3] / 1e3))
    f.write('\n')
    for region, s in sorted(stat.items(), key=lambda kv: -kv[1][2]):
        f.write(f'--- {region}: top kernel names\n')
        for n, t in times[region].most_common(14):
            f.write(f'   {names[region][n]:5d} {t / 1e3:8.3f} ms  {n}\n')
os.remove(trace)
print('ok')
