#!/usr/bin/env python
"""Headline benchmark: ScanNet-shaped scenes/sec of one Nesie-VoteNet training step
(forward + backward + grad all-reduce + clip + AdamW; 40 000 points/scene, batch 8 per
GPU, fp32), data-parallel over N MI355X -- BASELINE.json `metric`, workload configs[2].

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` = scenes all ranks processed / max-over-ranks
time of exactly K steps (inputs resident in HBM before the timed region).  `roofline`
prices the dominant kernel family of the step -- the fp32-MFMA layer kernels of the grouped
per-seed MLPs (nesie::pw_fwd_kernel / pw_wgrad_kernel) -- from HIP-event timings of every such
launch; `roofline_step` divides the same FLOPs by the whole step; `parity_gate` is the SURVEY
section 8(d) check run before anything is timed (CPU-oracle and HIP legs on identical inputs
and weights, every loss term within 1e-4, non-zero exit otherwise); `cpu_baseline` times the
same step on the host through the CPU oracle (bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cores():
    """Host threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU
    quota and by the GPU box's per-GPU CPU share (16), overridable with NESIE_CPU_CORES."""
    if os.environ.get('NESIE_CPU_CORES'):
        return int(os.environ['NESIE_CPU_CORES'])
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


# both OpenMP runtimes in the process (torch's and the oracle's) read this at start-up
os.environ.setdefault('OMP_NUM_THREADS', str(host_cores()))

import numpy as np
import torch
import torch.distributed as dist

from nesie_amd import dp, kernels
from nesie_amd.scenes import make_batch
from nesie_amd.votenet import build_nesie_votenet, nesie_votenet_scannet_cfg
from nesie_amd.votenet.nesie_head import GTBatch
from nesie_amd.votenet import semi

NUM_POINTS = 40000
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak


class KernelTimer:
    """HIP-event timing of one back-end method (same stream the kernel launches on:
    libnesie_hip.so launches on torch's current stream, so torch.cuda.Event brackets it)."""

    def __init__(self, backend, method, select):
        self.events, self.enabled = [], False
        inner = getattr(backend, method)

        def wrapped(*args, **kw):
            if self.enabled and select(*args):
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                inner(*args, **kw)
                e.record()
                self.events.append((s, e))
            else:
                inner(*args, **kw)
        setattr(backend, method, wrapped)

    def mean_ms(self):
        ts = [s.elapsed_time(e) for s, e in self.events]
        return sum(ts) / len(ts) if ts else None


class GemmTimer:
    """HIP-event timing + FLOP and algorithmic-byte count of every launch of one matrix-core
    entry point.  ``flops(*args, **kw)`` -> (flop, bytes, positions): launches over >= BIG
    positions (nb * p) are the grouped per-seed MLPs (set-abstraction stacks, MiniPointNets);
    the rest are the 1-D per-seed / per-proposal chains (P = 256 .. 1024 per scene)."""

    BIG = 32768

    def __init__(self, backend, method, flops, also=()):
        """``also`` = [(method, flops)]: more entry points counted in the same class (the forms of
        this one for SA1's rebuilt first activation)."""
        self.events, self.work, self.enabled = [], [], False
        self.method = method
        for name, fl in ((method, flops),) + tuple(also):
            self._wrap(backend, name, fl)

    def _wrap(self, backend, method, flops):
        inner = getattr(backend, method)

        def wrapped(*args, **kw):
            if not self.enabled:
                return inner(*args, **kw)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            r = inner(*args, **kw)
            e.record()
            self.events.append((s, e))
            self.work.append(tuple(flops(*args, **kw)) + (method,))
            return r
        setattr(backend, method, wrapped)

    def totals(self, big=True):
        """(launches, ms, flop, bytes) summed over the recorded launches of one size class."""
        n = ms = fl = by = 0.0
        for (s, e), w in zip(self.events, self.work):
            f, b, pos = w[:3]
            if (pos >= self.BIG) == big:
                n += 1; ms += s.elapsed_time(e); fl += f; by += b
        return n, ms, fl, by

    def launches(self):
        """-> [(shape signature, ms, flop, algorithmic bytes)] of every recorded launch."""
        out = []
        for (s, e), w in zip(self.events, self.work):
            out.append((f'{w[4]}{w[3]}', s.elapsed_time(e), w[0], w[1]))
        return out


def binding_roofline(timers, steps, mfma_peak_tflops, hbm_peak_gbs, list_below=0.5):
    """price every launch of the family against the roofline that BINDS it -- HBM-bound
    launches against HBM, GEMMs against the matrix cores: min_ms = max(flop / MFMA peak, algorithmic
    bytes / HBM peak) per launch, frac_binding = sum(min_ms) / sum(ms).  Launches of one shape are
    pooled; the shapes below ``list_below`` are the work queue (worst lost time first)."""
    pool = {}
    for t in timers:
        for sig, ms, fl, by in t.launches():
            e = pool.setdefault(sig, [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += ms; e[2] += fl; e[3] += by
    tot_ms = tot_min = 0.0
    rows = []
    for sig, (n, ms, fl, by) in pool.items():
        t_mfma = fl / (mfma_peak_tflops * 1e12) * 1e3
        t_hbm = by / (hbm_peak_gbs * 1e9) * 1e3
        mn = max(t_mfma, t_hbm)
        tot_ms += ms; tot_min += mn
        rows.append(dict(launch=sig, per_step=n / steps, ms_per_step=ms / steps,
                         bound='hbm' if t_hbm > t_mfma else 'mfma',
                         frac_binding=mn / ms if ms else None,
                         tflops=fl / (ms * 1e-3) / 1e12 if ms else None,
                         algorithmic_gbs=by / (ms * 1e-3) / 1e9 if ms else None,
                         lost_ms_per_step=(ms - mn) / steps))
    rows.sort(key=lambda r: -r['lost_ms_per_step'])
    return dict(frac_binding=tot_min / tot_ms if tot_ms else None,
                min_ms_per_step=tot_min / steps, ms_per_step=tot_ms / steps,
                hbm_bound_ms_per_step=sum(r['ms_per_step'] for r in rows if r['bound'] == 'hbm'),
                mfma_bound_ms_per_step=sum(r['ms_per_step'] for r in rows if r['bound'] == 'mfma'),
                peaks=dict(mfma_tflops=mfma_peak_tflops, hbm_gbs=hbm_peak_gbs),
                definition='per launch min_ms = max(flop / MFMA peak, algorithmic bytes / HBM peak); '
                           'frac_binding = sum(min_ms) / sum(HIP-event ms) over every launch of the family',
                # the work queue: every shape of the family, most lost time first; `below_half` names the ones
                # under half of their binding roofline
                by_lost_time=rows[:24],
                below_half=[r['launch'] for r in rows if r['frac_binding'] is not None and r['frac_binding'] < list_below])


GATE_LOSS_TOL = 1e-4      # north_star: fp32 features / losses within 1e-4 of the CPU path
GATE_GRAD_TOL = 5e-3      # flat gradient, HIP leg vs CPU-oracle leg (both fp32), relative L2 (measured 1.6e-3
#                           supervised at 2 scenes, 3.1e-3 Nesie student/teacher, 0.8e-3 SAQE at 3).  Round 5 tried
#                           4e-3 and went back: the distance between two fp32 legs is heavy-tailed -- a max-pool
#                           arg-max or a ReLU mask that sits within rounding of a tie routes one gradient term
#                           elsewhere, and the same code gives 5e-4 .. 1.8e-2 at 3 .. 8 scenes depending on where the
#                           rounding falls (tools/debug/gate_params.py) -- so a tighter bound here tests luck.  The
#                           fp64-referenced bound (HIP no farther from float64 than the CPU path is)
#                           lives in tests/test_parity_gpu.py; two fp32 legs sit 2e-3 .. 3.5e-3 from
#                           float64 each at this size, on either side of it


def _index_ops_equal(cpu_tree, hip_tree):
    """-> (tensors compared, names that differ): FPS picks, sampled centres, ball-query rows,
    composed input indices and the feature-propagation 3-NN taps of the whole backbone index
    chain: integer tensors and gathered coordinates bit for bit, the
    taps' inverse-distance weights to 1e-6."""
    bad, n = [], 0
    for lvl, (c, h) in enumerate(zip(cpu_tree, hip_tree)):
        pairs = [('fps_idx', c['indices'], h['indices']), ('new_xyz', c['new_xyz'], h['new_xyz']),
                 ('input_indices', c['input_indices'], h['input_indices'])]
        pairs += [(f'ball_query_idx[{i}]', a, b) for i, (a, b) in enumerate(zip(c['group_idx'], h['group_idx']))]
        for j, (ct, ht) in enumerate(zip(c.get('fp_taps', ()), h.get('fp_taps', ()))):
            pairs.append((f'fp_taps[{j}].idx', ct[0], ht[0]))
            n += 1
            if not torch.allclose(ht[1].cpu(), ct[1], rtol=1e-6, atol=1e-7):
                bad.append(f'level {lvl} fp_taps[{j}].weight')
        for name, a, b in pairs:
            n += 1
            if not torch.equal(b.cpu(), a):
                bad.append(f'level {lvl} {name}')
    return n, bad


def _gate_model(workload):
    """The gate's own model (never the timed one): supervised detector, or the student/teacher
    detector with a teacher biased so that its pseudo-label filters pass SOME proposals at random
    init (the filters must decide identically on both legs; tests/test_parity_gpu.py::_semi_pair)."""
    if workload == 'pretrain':
        model = build_nesie_votenet()
    else:
        model = semi.build_saqe_votenet_semi() if workload == 'saqe' else semi.build_nesie_votenet_semi()
        model.train_cfg.update(pos_distance_thr=1.0, neg_distance_thr=1.5)
        obj_bias = 1.3 if workload == 'saqe' else 1.7
        with torch.no_grad():
            model.bbox_head.conv_pred.conv_cls.bias[1] += obj_bias
            model.bbox_head.conv_pred.conv_cls.bias[2] += 0.75
            if workload == 'saqe':   # VoteNetSAQE filters on the quality head's objectness (last 2 of 38)
                model.bbox_head.grid_conv.mlps_head[6][6].bias[37] += obj_bias
        model.teacher.resync()
    return model.train()


def parity_gate(device, workload='pretrain', scenes=None, backward=True):
    """SURVEY section 8(d) / BASELINE.md section 2, before any timing: the CPU-oracle leg and the
    HIP leg run ONE training step (forward + backward) of the same full-size batch -- 2 scenes x
    40 000 points (supervised) or 3 scenes, 1 labeled : 2 unlabeled, student + teacher (semi /
    saqe) -- with identical weights, inputs and jitter:
      * index ops of the backbone chain (FPS, gathered centres, ball query, composed indices, FP
        3-NN taps) and the per-point vote targets: bit-exact;
      * semi / saqe: the teacher's pseudo-label decisions (validity, classes, class histogram): exact;
      * every loss term within 1e-4 (relative to max(1, |value|));
      * the flat parameter gradient within GATE_GRAD_TOL (relative L2, fp32 leg vs fp32 leg).
    ``scenes`` / ``backward``: the tests run the same check at the reference's own batch sizes
    (8 and 16 scenes per GPU, pretrain-010.py:248) forward-only.
    Two chains of discrete decisions over PREDICTED coordinates are replayed from the CPU leg
    (oracle/forcing.py: the vote FPS picks and the quality head's 3-NN grid taps; both kernels
    are compared bit for bit on identical inputs in tests/); how often the HIP leg's own decision
    agreed is reported.  -> dict for the JSON line; ``passed`` False ends the run with status 3."""
    import copy

    import oracle
    from oracle.forcing import force_grid_taps, force_vote_sampling
    torch.manual_seed(0)
    semi_like = workload != 'pretrain'
    nscene = scenes or (3 if semi_like else 2)
    pts, boxes, labels = make_batch(4242, nscene, NUM_POINTS)
    use_label = [i % 3 == 0 for i in range(nscene)]           # 1 labeled : 2 unlabeled (train-010.py:330)
    lab = [i for i, f in enumerate(use_label) if f]
    cpu_model = _gate_model(workload)
    # the head jitters its proposals with host-side Gaussian noise (nesie_head.py:178-209): both
    # legs get the same draw
    g = torch.Generator().manual_seed(3)
    k = cpu_model.bbox_head.num_proposal
    noise = (torch.randn(nscene, k, 3, generator=g), torch.randn(nscene, k, 3, generator=g))
    cpu_model.bbox_head.jitter_noise = noise
    sampler = force_vote_sampling(cpu_model, 'bench-gate-' + workload)
    tap_stats = force_grid_taps(cpu_model, 'bench-gate-' + workload)
    gpu_model = copy.deepcopy(cpu_model).to(device)     # (shares the recorded picks / taps)
    gpu_model.bbox_head.jitter_noise = tuple(t.to(device) for t in noise)
    gpu_sampler = gpu_model.bbox_head.vote_aggregation.points_sampler

    if semi_like:
        # the two augmented views are INPUTS of the step: built once (on the host) and handed to both
        # legs -- a bmm on either device would already differ in the last bit of a coordinate
        gen = torch.Generator().manual_seed(1)
        cpu_dev = torch.device('cpu')
        host_t = semi.AugMeta.random(nscene, cpu_dev, gen, strong=False)
        host_s = semi.AugMeta.random(nscene, cpu_dev, gen, strong=True)
        views = (host_s.apply_points(pts), host_t.apply_points(pts))

    def leg(model, dev):
        p = pts.to(dev)
        picks = {}
        if semi_like:
            model.init_label_state(120, 1081, dev)
            gen = torch.Generator().manual_seed(1)
            meta_t = semi.AugMeta.random(nscene, dev, gen, strong=False)
            meta_s = semi.AugMeta.random(nscene, dev, gen, strong=True)
            pts_s, pts_t = views[0].to(dev), views[1].to(dev)
            gt = GTBatch.collate([transform_gt(boxes[i], meta_s, i) for i in lab], [labels[i] for i in lab], dev)
            rows = (5 + 12 * torch.arange(nscene - len(lab), device=dev)) % 1081
            inner = model.get_pseudo_labels

            def recording(preds, name='ScanNet'):
                out = inner(preds, name)
                picks.update(labels=out[0].cpu(), valid=out[3].cpu())
                return out
            model.get_pseudo_labels = recording
            tree = model.backbone.sample_and_group_indices(pts_s)
            losses = model.forward_train(pts_s, pts_t, gt, use_label, meta_s, meta_t, rows)
            del model.get_pseudo_labels
            picks.update(ulb_list=model.state.ulb_list.cpu(), ulb_flag=model.state.ulb_flag.cpu())
            votes = ()
        else:
            gt = GTBatch.collate(boxes, labels, dev)
            tree = model.backbone.sample_and_group_indices(p)
            votes = tuple(t.cpu() for t in model.bbox_head.vote_targets_of(p, gt))
            losses = model.forward_train(p, None, gt, None)
        grads = {}
        if backward:
            model.parse_losses(losses).backward()
            grads = {n: q.grad.detach().double().cpu() for n, q in model.named_parameters() if q.grad is not None}
        return {kk: float(v.detach().sum()) for kk, v in losses.items()}, grads, tree, votes, picks

    from contextlib import nullcontext
    with (nullcontext() if backward else torch.no_grad()):
        with kernels.use_backend(oracle.OracleKernels()):
            want, want_g, want_tree, want_votes, want_picks = leg(cpu_model, torch.device('cpu'))
        got, got_g, got_tree, got_votes, got_picks = leg(gpu_model, device)
    if device.type == 'cuda':
        torch.cuda.synchronize(device)
    diffs = {kk: abs(got[kk] - want[kk]) / max(1.0, abs(want[kk])) for kk in want}
    worst = max(diffs, key=diffs.get)
    n_idx, bad_idx = _index_ops_equal(want_tree, got_tree)
    for i, (a, b) in enumerate(zip(want_votes, got_votes)):
        n_idx += 1
        same = torch.equal(a, b) if not a.dtype.is_floating_point else torch.allclose(a, b, rtol=0, atol=1e-6)
        if not same:
            bad_idx.append(f'vote_targets[{i}]')
    pseudo_ok = all(torch.equal(got_picks[kk][want_picks['valid']] if kk == 'labels' else got_picks[kk],
                                want_picks[kk][want_picks['valid']] if kk == 'labels' else want_picks[kk])
                    for kk in want_picks)
    names = sorted(n for n in want_g if n in got_g)
    grad_rel, per_param = 0.0, (0.0, None)
    if backward:
        w = torch.cat([want_g[n].flatten() for n in names])
        h = torch.cat([got_g[n].flatten() for n in names])
        grad_rel = float((h - w).norm() / w.norm())
        gmax = float(w.abs().max())
        ranked = sorted(((float((got_g[n] - want_g[n]).abs().max()) / max(float(want_g[n].abs().max()), 1e-3 * gmax), n)
                         for n in names), reverse=True)
        per_param = ranked[0]
    passed = (diffs[worst] <= GATE_LOSS_TOL and not bad_idx and pseudo_ok and grad_rel <= GATE_GRAD_TOL
              and set(want_g) == set(got_g))
    del gpu_model, cpu_model
    if device.type == 'cuda':
        torch.cuda.empty_cache()
    out = dict(passed=bool(passed), workload=workload, max_rel_diff=diffs[worst], worst_term=worst,
               terms=len(diffs), tolerance=GATE_LOSS_TOL,
               index_ops=dict(tensors_compared=n_idx, bit_exact=not bad_idx, differing=bad_idx[:8]),
               gradient=dict(flat_rel_l2_hip_vs_cpu=grad_rel, tolerance=GATE_GRAD_TOL, parameters=len(names),
                             worst_parameter=dict(name=per_param[1], max_err_over_max_grad=per_param[0]),
                             worst_parameters=[dict(name=n, max_err_over_max_grad=e) for e, n in ranked[:6]])
               if backward else None,
               own_vote_picks_agreed=bool(all(gpu_sampler.agreed)),
               grid_taps=dict(replayed_from_cpu_leg=True, grid_points_compared=tap_stats[1],
                              own_taps_differed=tap_stats[0]),
               sample=f'one {"student/teacher" if semi_like else "supervised"} step (forward{" + backward" if backward else " only"}), '
                      f'{nscene} scenes x {NUM_POINTS} pts, same weights and inputs on the CPU-oracle '
                      'leg and the HIP leg')
    if semi_like:
        out['pseudo_labels'] = dict(exact=bool(pseudo_ok), boxes_kept=int(want_picks['valid'].sum()),
                                    of=int(want_picks['valid'].numel()))
    return out


def transform_gt(boxes, meta, i):
    """GT boxes (K,7) of scene i carried into the student view."""
    one = semi.AugMeta(meta.flip_h[i:i + 1], meta.flip_v[i:i + 1], meta.rot_mat[i:i + 1],
                       meta.scale[i:i + 1], meta.trans[i:i + 1], meta.flow)
    return semi.transform_boxes(boxes.unsqueeze(0).to(meta.trans.device), one)[0].cpu()


_SIDE_STREAMS = {}


def build_step(device, batch, seed, lr, wd, graph=False, workload='pretrain', resident=0,
               noise=None):
    """-> (model, step, bucket).  step() = zero grads, forward, backward, gradient
    all-reduce (world > 1), clip, AdamW.  With graph=True the device work of a step is
    captured once into hipGraphs and replayed (the step has no host synchronisation): g1a =
    forward + the head's backward, g1b = the backbone's backward, g2 = clip + AdamW (+ EMA);
    the RCCL all-reduce of the head's gradient segment runs on a communication stream during
    g1b, the backbone's segment after it.
    ``noise`` = a fixed (centre, size) proposal-jitter pair for the tests (default: drawn on
    the device every step, as the reference does); ``step.inputs`` holds the batch tensors."""
    torch.manual_seed(0)
    on_gpu = device.type == 'cuda'
    pts, boxes, labels = make_batch(seed, batch, NUM_POINTS)
    pts = pts.to(device)
    loss_out = torch.zeros((), device=device)
    if workload in ('semi', 'saqe'):
        # BASELINE configs[3] (Nesie) / configs[4] (SAQE head, same step): 1 labeled + 2 unlabeled scenes per item, student and teacher
        # views of the same scenes (train-010.py:319-336); value counts STUDENT scenes
        model = (semi.build_saqe_votenet_semi() if workload == 'saqe'
                 else semi.build_nesie_votenet_semi()).to(device)
        model.init_label_state(120, 1081, device)
        use_label = [i % 3 == 0 for i in range(batch)]
        g = torch.Generator().manual_seed(seed)
        meta_t = semi.AugMeta.random(batch, device, g, strong=False)
        meta_s = semi.AugMeta.random(batch, device, g, strong=True)
        pts_s, pts_t = meta_s.apply_points(pts), meta_t.apply_points(pts)
        lab = [i for i, f in enumerate(use_label) if f]
        gt = GTBatch.collate([transform_gt(boxes[i], meta_s, i) for i in lab],
                             [labels[i] for i in lab], device)
        rows = torch.arange(sum(1 for f in use_label if not f), device=device)
    else:
        model = build_nesie_votenet().to(device)
        gt = GTBatch.collate(boxes, labels, device)
    scenes = None
    if resident and graph and on_gpu and workload == 'pretrain':
        # --resident-input: a data set of `resident` synthetic raw scans (50 000 points each)
        # lives in HBM; every step trains on a freshly sampled + augmented batch assembled on
        # the side stream (nesie_amd/input_pipeline.py) instead of one fixed batch
        from nesie_amd.input_pipeline import ResidentScenes
        from nesie_amd.scenes import make_scene
        scenes = ResidentScenes(device)
        for i in range(resident):
            p_, b_, l_ = make_scene(seed + 7 * i, 50000)
            centre = torch.cat([b_[:, :2], b_[:, 2:3] + b_[:, 5:6] * 0.5, b_[:, 3:6]], 1)
            scenes.add_scene(p_[:, :3].numpy(), None, centre.numpy(), l_.numpy())
        scenes.finalize()
        ids_next = torch.arange(batch, device=device) % resident
        pts, gt = scenes.assemble_batch(ids_next, num_points=NUM_POINTS)
        pts_next = pts.clone()
        gt_next = GTBatch(gt.boxes.clone(), gt.labels.clone(), gt.count.clone(), gt.valid.clone())
    model.train()
    if noise is not None:
        model.bbox_head.jitter_noise = tuple(t.to(device) for t in noise)
    if workload in ('semi', 'saqe'):
        inputs = dict(points_s=pts_s, points_t=pts_t, gt=gt, use_label=use_label, meta_s=meta_s,
                      meta_t=meta_t, rows=rows)
    else:
        inputs = dict(points=pts, gt=gt)
    # parameters and gradients as two flat vectors (dp.FlatTrainState): the gradient
    # all-reduce, the clip and AdamW each see ONE tensor
    groups = model.stacked_parameter_groups() if hasattr(model, 'stacked_parameter_groups') else None
    bucket = dp.FlatTrainState(model.parameters(), stack_groups=groups)
    if workload in ('semi', 'saqe') and os.environ.get('NESIE_FLAT_EMA', '1') != '0':
        model.teacher.use_flat(bucket)      # the teacher's weights as one vector too (EMATeacher.use_flat)
    if on_gpu:
        # clip (max_norm 10) + AdamW as two launches over the flat vectors (dp.FlatAdamW)
        opt = dp.FlatAdamW(bucket.flat_param, lr=lr, weight_decay=wd, max_norm=10)
    else:
        opt = torch.optim.AdamW([bucket.flat_param], lr=lr, weight_decay=wd, foreach=True)

    # The backward pass is cut at the backbone's output (dp.backward_head / backward_rest):
    # the head's gradients (the tail of the flat vector) are complete after phase 1 and their
    # all-reduce runs on a communication stream while phase 2 (the backbone) still computes.
    n_bb, e_bb = bucket.split_after(model.backbone.parameters())
    e_all = bucket.flat.numel()
    head_params, backbone_params = bucket.params[n_bb:], bucket.params[:n_bb]
    comm = dp.SegmentedAllReduce(bucket.flat)
    model.keep_head_inputs = True
    cut = {}

    def forward_losses(pre=None):
        if workload in ('semi', 'saqe'):
            return model.forward_train(pts_s, pts_t, gt, use_label, meta_s, meta_t, rows,
                                       precomputed=pre)
        return model.forward_train(pts, None, gt, None, precomputed=pre)

    # While the forward pass runs, the side stream's sampling kernels for the NEXT batch hold one CU per
    # scene and XCD (1024 threads, 160 KB of LDS: nothing else fits beside them), and a one-round
    # persistent grid sized for all 32 CUs of an XCD then has two workgroups that find no CU until the
    # round ends: the forward's persistent grids are sized for the CUs
    # that are left.  NESIE_FWD_CUS: A/B switch (256 = the whole chip)
    # (the un-captured form of the step sizes them the same way: tests compare the two forms bit for bit)
    # (one sampling workgroup per scene, dealt round-robin over the 8 XCDs: ceil(batch / 8) CUs per XCD,
    # plus two more per XCD: the 2-per-CU layer launches still stall beside the sampling with one CU
    # spare -- 131 us against 104 us with three, tools/debug/fps_interference.py -- and the sweep of the
    # whole step agrees: 232 at 8 scenes (256: 13.42, 248: 13.26, 232: 13.15, 224: 13.19, 216: 13.26 ms),
    # 224 at 16 (240: 36.17, 224: 35.64, 208: 36.07 ms))
    left = max(256 - 16 - 8 * ((batch + 7) // 8), 192)
    budget = {'cus': int(os.environ.get('NESIE_FWD_CUS', str(left))) if on_gpu else 256}
    # ... and the LATE part of the forward pass -- from the detection head on -- is sized for 16 CUs more:
    # the next batch's 40 000-point sampling launch is over by then, the shorter levels' sampling and the
    # index kernels behind it still run (the whole chip for the head: 12.10 ms; 248 from the head on:
    # 11.85; 232 throughout: 11.92 -- same box, alternating; from SA4 or the first FP level on: no better).
    # (A budget for the head's BACKWARD -- 248 / 240 -- costs 0.12 / 0.2 ms: the chain is over by then.  A third
    # stage of 240 from SA3 / SA4 / the first FP level on: + 0.15 / + 0.15 / 0.0 ms.)
    # Supervised step only (the student / teacher steps run two forwards under one chain): 0 = off
    late_default = min(left + 16, 248) if (workload == 'pretrain' and left < 248) else 0
    late_cus = int(os.environ.get('NESIE_FWD_CUS_LATE', str(late_default))) if on_gpu else 0
    if late_cus and budget['cus'] != 256:
        late_mod = model
        for part in os.environ.get('NESIE_FWD_LATE_AT', 'bbox_head').split('.'):
            late_mod = getattr(late_mod, part)

        def _late(mod, args):
            if budget['cus'] != 256 and kernels._lib.load().nesie_get_cu_count() == budget['cus']:   # (only inside the budgeted forward)
                kernels._lib.call('nesie_set_cu_count', late_cus)
        late_mod.register_forward_pre_hook(_late)
    budget['late'] = late_cus

    head_dummy = None
    if on_gpu and os.environ.get('NESIE_DIAG_HEAD_DUMMY'):   # diagnostic only: a long kernel in front of the step
        n_ = int(os.environ['NESIE_DIAG_HEAD_DUMMY'])
        head_dummy = (torch.randn(n_, n_, device=device), torch.empty(n_, n_, device=device))

    def phase1(pre=None):       # forward + the head's backward
        if head_dummy is not None:
            torch.mm(head_dummy[0], head_dummy[0], out=head_dummy[1])
        bucket.begin()
        if on_gpu and budget['cus'] != 256:
            with kernels.HipKernels.cu_budget(budget['cus']):
                losses = forward_losses(pre)
        else:
            losses = forward_losses(pre)
        total = model.parse_losses(losses)
        cut['boundary'] = model.take_head_inputs()
        cut['grads'] = dp.backward_head(total, cut['boundary'], head_params)
        bucket.collect(n_bb, None)
        loss_out.copy_(total.detach())

    def phase2():               # the backbone's backward
        dp.backward_rest(cut['boundary'], cut['grads'], backbone_params)
        bucket.collect(0, n_bb)
        cut.clear()

    def update():
        if not on_gpu:
            torch.nn.utils.clip_grad_norm_([bucket.flat_param], max_norm=10, norm_type=2)
        opt.step()
        if workload in ('semi', 'saqe'):
            model.teacher.update(1000)  # past the warm-up: momentum 0.001 (simi_teacher_hook.py:57-58)

    comm.steps = 0

    def eager_step(pre=None):
        comm.steps += 1
        phase1(pre)
        comm.launch(e_bb, e_all)
        phase2()
        comm.launch(0, e_bb)
        comm.wait()
        update()
        return loss_out

    eager_step.inputs, eager_step.optimizer, eager_step.comm = inputs, opt, comm
    if not (graph and on_gpu):
        return model, eager_step, bucket

    # ---- software pipeline: the FPS + ball-query index chain of the backbone depends only on
    # the input coordinates, never on the weights, so the chain for the NEXT step runs on a
    # side stream (it occupies one CU per scene) while this step trains.  Every step still
    # executes one full index chain and one full fwd+bwd+update; nothing is cached.
    pipelined = True
    # few, long, latency-bound workgroups.  ONE side stream per device and process: the runtime maps
    # streams onto a handful of hardware queues round-robin, and the third stream a process creates can
    # land on the main stream's queue -- the chain then runs IN LINE with the step (measured: the SAQE
    # step at 16 scenes 70.5 ms instead of 35.3 as the third workload built in one process)
    side = _SIDE_STREAMS.get(device)
    if side is None:
        side = _SIDE_STREAMS[device] = torch.cuda.Stream(device, priority=-1)
    main = torch.cuda.current_stream(device)

    from nesie_amd.votenet.backbone import (copy_tensors, index_tree_like,
                                            index_tree_tensors as flat, pack_tensors)

    semi_like = workload in ('semi', 'saqe')
    # weight-independent work of a step: the backbone's index chain(s) and, for the supervised
    # step, the per-point vote targets (points-in-boxes over 40 000 x T)
    def gt_parts(g_):
        return [g_.boxes, g_.labels, g_.count, g_.valid]

    # --resident-input: the per-sample decisions (which 40 000 of a scan's points, flips, rotation,
    # scale) are drawn on the HOST with numpy in the reference's order (input_pipeline.
    # draw_like_reference, what the reference's data-loader workers do) while the device trains, and
    # reach the device as ONE 1.3 MB DMA copy from pinned memory ahead of the side-stream graph:
    # no sort and no random fills on the device (NESIE_HOST_DRAWS=0: drawn on the device instead --
    # an argsort over 8 x 50 000 random keys, 36 launches -- the round-3 form).
    host_draws = scenes is not None and os.environ.get('NESIE_HOST_DRAWS', '1') != '0'
    noise = scenes.new_noise(batch, NUM_POINTS) if scenes is not None and not host_draws else None
    staging = scenes.new_staging(batch, NUM_POINTS) if host_draws else None
    draw_state = dict(rng=np.random.RandomState(seed + 12345), ids=[i % resident for i in range(batch)],
                      parity=0, done=[torch.cuda.Event(), torch.cuda.Event()]) if host_draws else None

    def stage_next_draws():
        """Host: decisions for the batch the next chain assembles -> pinned buffer -> device (side stream)."""
        st = draw_state
        st['ids'] = [(i + batch) % resident for i in st['ids']]
        k = st['parity']
        st['done'][k].synchronize()           # the copy that last read this pinned buffer is over
        scenes.stage_draws(staging, k, st['ids'], st['rng'])
        staging['dev'].copy_(staging['host'][k], non_blocking=True)
        st['done'][k].record()
        st['parity'] = 1 - k

    def assemble_next():
        ids_next.copy_((ids_next + batch) % resident)
        p_, g_ = scenes.assemble_batch(ids_next, num_points=NUM_POINTS, noise=noise, staging=staging)
        copy_tensors([pts_next] + gt_parts(gt_next), [p_] + gt_parts(g_))

    def input_only_work():
        if scenes is not None:
            assemble_next()
            return (model.backbone.sample_and_group_indices(pts_next),
                    list(model.bbox_head.vote_targets_of(pts_next, gt_next)))
        if semi_like:
            return (model.backbone.sample_and_group_indices(pts_s)
                    + model.backbone.sample_and_group_indices(pts_t), [])
        return (model.backbone.sample_and_group_indices(pts),
                list(model.bbox_head.vote_targets_of(pts, gt)))
    def stage(msg):   # NESIE_DIAG_STAGES=1: synchronise and report where the build is
        if os.environ.get('NESIE_DIAG_STAGES'):
            torch.cuda.synchronize(device)
            print('[stage]', msg, file=sys.stderr, flush=True)
    stage('model + optimiser built')
    if scenes is not None:
        stage_next_draws() if host_draws else scenes.refresh_noise(noise)
    idx_next, votes_next = input_only_work()
    # two static copies of the index set (next: written by the side-stream graph, cur: read by
    # the step graphs), each packed into ONE buffer: the hand-over is a single copy
    n_tree = len(flat(idx_next))
    views, arena_next = pack_tensors(flat(idx_next) + votes_next)
    idx_next, votes_next = index_tree_like(idx_next, views[:n_tree]), views[n_tree:]
    views, arena_cur = pack_tensors(views)
    idx_cur, votes_cur = index_tree_like(idx_next, views[:n_tree]), views[n_tree:]
    stage('first input-only pass (eager)')
    nlev = len(idx_cur) // 2 if semi_like else len(idx_cur)

    def current_indices():
        if semi_like:
            return dict(student=idx_cur[:nlev], teacher=idx_cur[nlev:])
        return dict(indices=idx_cur, vote_targets=tuple(votes_cur))

    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(2):  # warm allocator / library handles before capture
            eager_step(current_indices())
    main.wait_stream(side)
    torch.cuda.synchronize(device)
    stage('eager warm-up steps')
    g1a, g1b, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    # thread_local: a process group's watchdog thread may query events while this thread captures
    mode = dict(capture_error_mode='thread_local')
    # one rank (no exchange between the phases): the whole step is ONE graph -- two replay boundaries
    # less per step (NESIE_ONE_GRAPH=0: keep the three graphs of the multi-rank form, A/B)
    one_graph = not comm.active and os.environ.get('NESIE_ONE_GRAPH', '1') != '0'
    if one_graph:
        with torch.cuda.graph(g1a, **mode):
            phase1(current_indices())
            phase2()
            update()
    else:
        with torch.cuda.graph(g1a, **mode):
            phase1(current_indices())
        with torch.cuda.graph(g1b, pool=g1a.pool(), **mode):
            phase2()
        with torch.cuda.graph(g2, pool=g1a.pool(), **mode):
            update()
    stage('step graphs captured')
    if pipelined:
        g_idx = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_idx, stream=side, **mode):
            if os.environ.get('NESIE_DIAG_SIDE_DUMMY'):   # diagnostic only: N tiny kernels instead of the chain
                dummy = torch.zeros(64, device=device)
                for _ in range(int(os.environ['NESIE_DIAG_SIDE_DUMMY'])):
                    dummy.add_(1.0)
            elif os.environ.get('NESIE_DIAG_SIDE_FPS_ONLY'):   # diagnostic only: the four sampling launches alone
                from nesie_amd.mmdet3d_ops.furthest_point_sample import FurthestPointSampling
                lev = pts[..., :3].contiguous()
                for npoint in (2048, 1024, 512, 256):
                    pick = FurthestPointSampling.apply(lev, npoint)
                    lev = lev.gather(1, pick.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            else:
                fresh, fresh_votes = input_only_work()
                copy_tensors(flat(idx_next) + votes_next, flat(fresh) + fresh_votes)
        ready, copied = torch.cuda.Event(), torch.cuda.Event()
        stage('input graph captured')
        with torch.cuda.stream(side):
            if host_draws:
                stage_next_draws()
            g_idx.replay()
            ready.record(side)
        stage('input graph replayed once')

    def graph_step():
        if pipelined:
            main.wait_event(ready)                       # this step's indices are complete
            arena_cur.copy_(arena_next)
            if scenes is not None:                       # ... and so is this step's batch
                copy_tensors([pts] + gt_parts(gt), [pts_next] + gt_parts(gt_next))
            copied.record(main)
            side.wait_event(copied)
            with torch.cuda.stream(side):                # next step's index chain, overlapped
                if scenes is not None:   # this batch's decisions: one DMA copy (or 4 in-place random fills)
                    stage_next_draws() if host_draws else scenes.refresh_noise(noise)
                if not os.environ.get('NESIE_DIAG_SKIP_CHAIN'):  # diagnostic only
                    g_idx.replay()
                ready.record(side)
        stage('input graph launched for the next step')
        comm.steps += 1
        if one_graph:
            if hasattr(opt, 'sync_hyper'):
                opt.sync_hyper()     # a scheduler's new lr / wd reaches the captured update
            g1a.replay()             # forward + backward + update
            stage('step graph replayed')
            return loss_out
        g1a.replay()                 # forward + head backward
        comm.launch(e_bb, e_all)     # the head's gradients travel ...
        g1b.replay()                 # ... while the backbone's backward computes
        comm.launch(0, e_bb)
        comm.wait()
        stage('forward/backward graphs replayed')
        if hasattr(opt, 'sync_hyper'):
            opt.sync_hyper()         # a scheduler's new lr / wd reaches the captured update
        g2.replay()
        stage('update graph replayed')
        return loss_out
    graph_step.eager = eager_step
    graph_step.forward_cu_budget = budget
    graph_step.graphs_per_step = 1 if one_graph else 3
    graph_step.inputs, graph_step.optimizer, graph_step.comm = inputs, opt, comm
    return model, graph_step, bucket


def cpu_baseline(sample_batch, steps, warmup=2):
    """The same training step on the host cores: index ops through oracle/ (the CPU
    restatement, OpenMP), dense ops through PyTorch-CPU.  kind = "port".  BASELINE.md section 2:
    2 warm-up steps, then the MEDIAN of >= 5 timed steps."""
    import statistics

    import oracle
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = nesie_votenet_scannet_cfg()
    with kernels.use_backend(oracle.OracleKernels()):
        model, step, _ = build_step(torch.device('cpu'), sample_batch, 1000,
                                    cfg['optimizer']['lr'], cfg['optimizer']['weight_decay'])
        for _ in range(warmup):  # allocator, oneDNN primitives, OpenMP teams
            step()
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            step()
            times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    avail = len(os.sched_getaffinity(0))
    return dict(value=sample_batch / dt, unit='scenes/s', cores=cores, kind='port',
                cores_available=avail,
                cores_note=(f'{cores} of the {avail} host cores visible to this process: capped at the GPU box\'s '
                            'per-GPU CPU share (16) so that N ranks on one host do not over-subscribe it; '
                            'NESIE_CPU_CORES overrides'),
                step_seconds=dict(median=dt, min=min(times), max=max(times), mean=sum(times) / len(times)),
                sample=f'median of {steps} timed training step(s) (after {warmup} warm-ups) of '
                       f'{sample_batch} scene(s) x {NUM_POINTS} pts, fwd+bwd+AdamW, oracle index ops + '
                       f'PyTorch-CPU dense ops on {cores} host threads, {dt:.2f} s/step')


def other_workload(device, workload, batch, steps, warmup, gate=True):
    """BASELINE configs[3] / [4] beside the headline: the student/teacher step of ``workload`` ('semi':
    Nesie, 8 scenes per GPU, train-010.py:320-330; 'saqe': SAQE head, 16 scenes per GPU,
    saqe-votenet-scannet-train-010.py) captured and replayed like the headline step -- ``steps`` timed
    replays after ``warmup`` -- with its own parity-gate summary (one full-size 3-scene student/teacher
    step on the CPU-oracle leg and the HIP leg).  -> dict for the JSON line."""
    import gc
    res = dict(workload=workload, scenes_per_gpu=batch, steps=steps, warmup=warmup)
    if gate:
        g = parity_gate(device, workload)
        res['parity_gate'] = dict(passed=g['passed'], max_rel_diff=g['max_rel_diff'], worst_term=g['worst_term'],
                                  terms=g['terms'], index_ops_bit_exact=g['index_ops']['bit_exact'],
                                  pseudo_labels=g.get('pseudo_labels'),
                                  flat_gradient_rel_l2=g['gradient']['flat_rel_l2_hip_vs_cpu'], sample=g['sample'])
        if not g['passed']:
            return res
    cfg = nesie_votenet_scannet_cfg()
    model, step, bucket = build_step(device, batch, 1000, cfg['optimizer']['lr'], cfg['optimizer']['weight_decay'],
                                     graph=True, workload=workload)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res.update(value=batch * steps / dt, unit='student scenes/s', ms_per_step=dt / steps * 1e3,
               loss_is_finite=bool(torch.isfinite(loss).item()),
               graphs_per_step=getattr(step, 'graphs_per_step', None),
               forward_cu_budget=(getattr(step, 'forward_cu_budget', None) or {'cus': 256})['cus'])
    del model, step, bucket, loss
    gc.collect()
    torch.cuda.empty_cache()
    return res


def spawn_ranks(n, argv, script=None):
    """``python bench.py --gpus N`` without a launcher (WORLD_SIZE unset): start N fresh child
    processes of this script -- one rank per GPU, the torchrun environment contract (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT) -- BEFORE this process has made
    any GPU call (it never makes one: it only waits; a process that has touched the GPU must not
    exec or fork workers on this pool).  Rank 0's stdout (the ONE JSON line) and every rank's
    stderr pass through; the exit code is the first non-zero child code, returned as soon as that
    child has exited (the other ranks are terminated, not waited for).  The reference gets its
    ranks from ``train.py --launcher`` -> ``mmdet.apis.train_detector`` (train.py:71-74, 131-136)."""
    import socket
    import subprocess
    with socket.socket() as sock:           # a free rendezvous port
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                LOCAL_WORLD_SIZE=str(n))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # host threads per rank: the box's cores shared between the ranks
    base['OMP_NUM_THREADS'] = str(max(1, host_cores() // n))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv),
                                      env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        # poll ALL ranks: a rank that dies during init or inside a collective leaves the others
        # waiting in RCCL / gloo until the collective time-out, so the first non-zero exit ends the
        # job -- terminate the rest, kill what ignores that after a grace period
        live = list(procs)
        while live and not rc:
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                rc = rc or code
            if live and not rc:
                time.sleep(0.2)
        if rc:
            for pr in live:
                pr.terminate()
            deadline = time.monotonic() + 10.0
            for pr in live:
                try:
                    pr.wait(timeout=max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
    except KeyboardInterrupt:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        rc = 130
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='scenes per GPU')
    ap.add_argument('--graph', type=int, default=1, help='replay the step as hipGraphs')
    ap.add_argument('--workload', default='pretrain', choices=['pretrain', 'semi', 'saqe'],
                    help='pretrain = BASELINE configs[2] (the metric); semi = configs[3]; '
                         'saqe = configs[4] (use --batch 16)')
    ap.add_argument('--resident-input', type=int, default=0,
                    help='N > 0: keep N synthetic raw scans in HBM and assemble a fresh sampled + '
                         'augmented batch per step on the side stream (input_pipeline.py)')
    ap.add_argument('--cpu-baseline', type=int, default=1, help='0 to skip the CPU leg')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--cpu-steps', type=int, default=9)
    ap.add_argument('--loss-trace', type=int, default=0,
                    help='1: keep the total loss of every timed step (a device-side copy per step, no '
                         'synchronisation) and print it as `loss_trace` (tests)')
    ap.add_argument('--other-workloads', type=int, default=1,
                    help='1 (N = 1, pretrain runs only): after the headline, also replay BASELINE configs[3] (semi, 8 '
                         'scenes) and configs[4] (saqe, 16 scenes) for a few steps each, with their parity gates, and '
                         'report them as `other_workloads`; 0 to skip')
    ap.add_argument('--parity-gate', type=int, default=1,
                    help='0 to skip the CPU-vs-HIP check (index ops, loss dict, gradients) that '
                         'precedes the timing (N = 1 only)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # bare `python bench.py --gpus N`: this process becomes the launcher of N ranks
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank, world, local = dp.init_distributed()
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; run '
              f'`python bench.py --gpus {args.gpus}` (it starts its own ranks) or launch '
              f'{args.gpus} ranks', file=sys.stderr)
        sys.exit(2)
    # (several ranks may share a device only in the gloo rehearsal, see dp.init_distributed)
    device = torch.device('cuda', local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    from nesie_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing

    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == args.gpus, \
            f'process group has {dist.get_world_size()} ranks, --gpus {args.gpus}'
    gate = None
    if args.parity_gate and world == 1:
        gate = parity_gate(device, args.workload)
        if not gate['passed']:
            print(json.dumps({'parity_gate': gate}), flush=True)
            print(f"parity gate FAILED: worst loss term {gate['worst_term']} differs by "
                  f"{gate['max_rel_diff']:.3e}; index ops exact: {gate['index_ops']['bit_exact']}; "
                  f"gradient {gate['gradient']}", file=sys.stderr)
            sys.exit(3)
    cfg = nesie_votenet_scannet_cfg()
    model, step, bucket = build_step(device, args.batch, 1000 + 100 * rank,
                                     cfg['optimizer']['lr'], cfg['optimizer']['weight_decay'],
                                     graph=bool(args.graph), workload=args.workload,
                                     resident=args.resident_input)
    hip = kernels.backend_for(torch.empty(1, device=device))
    # longest single launch: D-FPS over the 40 000-point scene (latency-bound; in graph mode it
    # runs on a side stream under the previous step, off the critical path)
    fps_timer = KernelTimer(hip, 'furthest_point_sampling_wrapper',
                            lambda b, n, m, *_: n == NUM_POINTS)
    # largest streaming launches on the critical path: the SA1 MLP tail (B,128,2048,64) through
    # the fused norm+ReLU+max-pool kernels, and the (B,64,2048,64) layers through norm+ReLU
    big = args.batch * 128 * 2048 * 64
    mid = args.batch * 64 * 2048 * 64
    pool_fwd_timer = KernelTimer(hip, 'bn_relu_maxpool_forward', lambda x, *_: x.numel() == big)
    pool_bwd_timer = KernelTimer(hip, 'bn_relu_maxpool_backward',
                                 lambda g, a, x, *_: x.numel() == big)
    bn_fwd_timer = KernelTimer(hip, 'bn_relu_forward', lambda x, *_: x.numel() == mid)
    bn_bwd_timer = KernelTimer(hip, 'bn_relu_backward', lambda dy, *_: dy.numel() == mid)
    # the matrix-core family: every launch of the layer kernel (forward products and, on the
    # transposed weight view, input gradients) and of the weight-gradient kernel
    # -> (flop, algorithmic HBM bytes, positions): operands read once, results written once
    def fwd_flop(x, w, **kw):
        nb, k, p = x.shape
        co = w.shape[1]
        out = nb * co * p * 4 if kw.get('y') is not None else 0
        tag = ('T' if w.stride(2) != 1 else '') + ('+stats' if kw.get('stat_part') is not None else '') + \
              ('+pool' if kw.get('pool_group') else '') + ('+rowbias' if kw.get('row_bias') is not None else '') + \
              ('' if kw.get('y') is not None else '-nostore')
        return 2.0 * nb * k * co * p, nb * k * p * 4 + out, nb * p, f'[{k}->{co}{tag}, {nb}x{p}]'

    def wgrad_flop(dy, x, dw, **kw):
        nb, co, p = dy.shape
        return 2.0 * nb * co * x.shape[1] * p, nb * (co + x.shape[1]) * p * 4, nb * p, f'[{co}x{x.shape[1]}, {nb}x{p}]'

    def dgrad_flop(dy, w, z, z_coef, da, **kw):   # + the raw output Z of the norm-backward reduction
        nb, k, p = dy.shape
        co = w.shape[1]
        return 2.0 * nb * k * co * p, nb * (k + 2 * co) * p * 4, nb * p, f'[{k}->{co}, {nb}x{p}]'
    def wgrad_bn_flop(da, z, z_coef, gamma, part, x, dz, dw, *a, **kw):
        # the weight gradient with the norm backward's apply pass inside: the SAME algorithmic FLOPs
        # (the per-element transform is not counted); reads dA, Z, X once, writes dZ once
        nb, co, p = da.shape
        return 2.0 * nb * co * x.shape[1] * p, nb * (3 * co + x.shape[1]) * p * 4, nb * p, f'[{co}x{x.shape[1]}, {nb}x{p}]'
    def tail_flop(g, pooled, zstar, argmax, coef, gamma, w, z_prev, coef_prev, ns, *a, **kw):
        # backward of a pooled tail without its dense tensors (csrc/pool_tail.hip), ALL its launches as
        # one entry: the layer's input gradient + weight gradient, 2 x 2 nb c k p (what it issues is
        # 2 nb (k + c) k p built-row product + 2 nb k k p Gram product: the same count at c = 2 k);
        # reads the operand's raw form twice, writes dA once
        nb, k, p = z_prev.shape
        c = w.shape[0]
        return 4.0 * nb * c * k * p, 3 * nb * k * p * 4, nb * p, f'[{k}->{c}, {nb}x{p}]'
    # ... and their forms over SA1's rebuilt first activation (Z0 = W0 . X4 is never stored): the same
    # products, the 64-row operand replaced by the 4 rows of X4 in the byte count; the rebuild's own
    # 2 * 4 * 64 flop per position are overhead, not counted
    def fwd_k4_flop(x4, w0, w, in_coef, y, stat_part):
        nb, _, p = x4.shape
        co = w.shape[0]
        return 2.0 * nb * 64 * co * p, nb * (4 + co) * p * 4, nb * p, f'[4=>64->{co}+stats, {nb}x{p}]'

    def dgrad_k4_flop(dy, w, x4, w0, z_coef):
        nb, k, p = dy.shape
        return 2.0 * nb * k * 64 * p, nb * (k + 4) * p * 4, nb * p, f'[{k}->64-nostore, {nb}x{p}]'

    def wgrad_bn_k4_flop(da, z, z_coef, gamma, part, x4, *a, **kw):
        nb, co, p = da.shape
        return 2.0 * nb * co * 64 * p, nb * (3 * co + 4) * p * 4, nb * p, f'[{co}x64<=4, {nb}x{p}]'
    gemm_timers = (GemmTimer(hip, 'pw_layer_forward', fwd_flop, also=[('pw_layer_forward_k4', fwd_k4_flop)]),
                   GemmTimer(hip, 'pw_wgrad', wgrad_flop),
                   GemmTimer(hip, 'pw_dgrad_bn_reduce', dgrad_flop, also=[('pw_dgrad_bn_reduce_k4', dgrad_k4_flop)]),
                   GemmTimer(hip, 'pw_wgrad_bn_backward', wgrad_bn_flop, also=[('pw_wgrad_bn_backward_k4', wgrad_bn_k4_flop)]),
                   GemmTimer(hip, 'pool_tail_backward', tail_flop))
    bn_apply_timer = KernelTimer(hip, 'bn_relu_backward_apply', lambda dy, *_: dy.numel() == mid)
    timers = (fps_timer, pool_fwd_timer, pool_bwd_timer, bn_fwd_timer, bn_bwd_timer, bn_apply_timer) + gemm_timers

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = []
    for _ in range(args.warmup):
        loss = step()
        if args.loss_trace:
            trace.append(loss.detach().clone())
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
        if args.loss_trace:
            trace.append(loss.detach().clone())
    sync()
    elapsed = time.perf_counter() - t0
    # Kernel-level timing for the roofline entry: HIP events cannot bracket a kernel inside
    # a replayed graph, so the same K steps are run once more un-captured with the events
    # on the launch stream (identical kernels, identical inputs).
    for t in timers:
        t.enabled = True
    eager = getattr(step, 'eager', step)
    eager_steps = min(args.steps, 5)
    # (nothing runs beside these launches: they are timed with grids sized for the whole chip, not for
    # the CUs the replayed step's forward leaves to the sampling kernels)
    fwd_budget = getattr(step, 'forward_cu_budget', None) or {'cus': 256}
    budget_in_step, fwd_budget['cus'] = fwd_budget['cus'], 256
    for _ in range(eager_steps):
        eager()
    torch.cuda.synchronize()
    fwd_budget['cus'] = budget_in_step
    for t in timers:
        t.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        # FPS 40000->2048: algorithmic HBM bytes per launch = read xyz + temp once, write
        # temp + idx (DESIGN.md "FPS"): B * (N*12 + N*4 + N*4 + M*4)
        fps_ms = fps_timer.mean_ms()
        alg_bytes = args.batch * (NUM_POINTS * 20 + 2048 * 4)
        achieved = alg_bytes / (fps_ms * 1e-3) / 1e9 if fps_ms else None
        out = {
            'metric': 'ScanNet scenes/sec (fwd+bwd, 40k pts, bs=8)',
            'value': value, 'unit': 'scenes/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('Nesie-VoteNet supervised pretrain step '
                                    '(nesie-votenet-scannet-pretrain-10%): fwd+bwd+AdamW, '
                                    '40000 pts/scene, fp32, random-init weights')
                       if args.workload == 'pretrain' else
                       ('%s student/teacher semi-sup step (%s-votenet-scannet-train-10%%): '
                        % (('SAQE', 'saqe') if args.workload == 'saqe' else ('Nesie', 'nesie')) +
                        'student fwd+bwd on B scenes (1 labeled : 2 unlabeled) + EMA-teacher '
                        'fwd on B + pseudo labels + AdamW + EMA; value = student scenes/s'),
                       'scenes_per_gpu': args.batch, 'global_batch': world * args.batch,
                       'points_per_scene': NUM_POINTS,
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'world_size': dist.get_world_size() if dist.is_initialized() else 1,
                       'collective_backend': dist.get_backend() if dist.is_initialized() else None,
                       # all-reduce launches per step (2 = head segment during the backbone's backward
                       # graph + backbone segment after it; 0 without a process group)
                       'collectives_per_step': step.comm.collectives / max(1, step.comm.steps),
                       'hip_graph': bool(args.graph),
                       # every scatter-add of the backward in an order fixed by the indices (no float atomics)
                       'deterministic_backward': bool(kernels.HipKernels.DETERMINISTIC),
                       'graphs_per_step': getattr(step, 'graphs_per_step', None),
                       'index_chain_pipelined': bool(args.graph),
                       'forward_cu_budget': (getattr(step, 'forward_cu_budget', None) or {'cus': 256})['cus'],
                       'forward_cu_budget_from_the_head_on': (getattr(step, 'forward_cu_budget', None) or {}).get('late') or None,
                       'grad_allreduce_bytes': bucket.nbytes(),
                       'grad_allreduce': 'two segments on a communication stream: head gradients '
                                         'during the backbone backward graph, backbone gradients after it',
                       'resident_input_scenes': args.resident_input},
        }
        # HBM-streaming kernels, priced on their largest launches (537 MB / 268 MB tensors at
        # B = 8, beyond the 256 MB Infinity Cache).  Algorithmic bytes:
        #   SA tail backward = gather sums (pooled-size) + read x + write dx      = 2 passes
        #   SA tail forward  = stats read + pool read                             = 2 passes
        #   norm+ReLU fwd    = stats read + apply read/write                      = 3 passes
        #   norm+ReLU bwd    = reduce (dy, y) + apply (dy, x -> dx)               = 5 passes
        def _stream(name, ms, passes, tensor_bytes, extra=0):
            if not ms:
                return None
            nbytes = passes * tensor_bytes + extra
            a = nbytes / (ms * 1e-3) / 1e9
            return {'kernel': name, 'bound': 'hbm', 'achieved': a, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': a / HBM_PEAK_GBS, 'traffic': None,
                    'avg_launch_ms': ms, 'algorithmic_bytes_per_launch': nbytes}
        pooled_bytes = args.batch * 128 * 2048 * 4
        # ---- the dominant family: the grouped per-seed MLP GEMMs on the fp32 matrix cores.
        # achieved = algorithmic FLOPs (2 * batches * K * Cout * positions per launch) / HIP-event
        # time, summed over every launch of the family in the un-captured steps.
        n_l, ms_l, fl_l, by_l = (a + b for a, b in zip(gemm_timers[0].totals(), gemm_timers[2].totals()))
        n_w, ms_w, fl_w, by_w = gemm_timers[1].totals()
        # weight-gradient launches that also carry the BatchNorm + ReLU backward's apply pass (dZ formed
        # on the operand load from (dA, Z) and written once): 4 tensor passes per launch instead of 2,
        # HBM co-bound -- priced on their own below, and shown inside the family as well
        n_wf, ms_wf, fl_wf, by_wf = gemm_timers[3].totals()
        # the pooled tail's backward (built-row input gradient + Gram / sparse weight gradient + its small
        # launches, timed as ONE entry; of its 3 operand passes the 2 of the layer-kernel launch are in the
        # PMC family rows)
        n_t, ms_t, fl_t, by_t = gemm_timers[4].totals()
        # ... and the same kernels on the 1-D per-seed / per-proposal chains (vote module, prediction
        # trunk, feature propagation, score heads): 8 x 256 .. 1024 positions, launch-bound
        sn, sms, sfl, sby = (sum(v) for v in zip(*(t.totals(big=False) for t in gemm_timers)))
        if n_l + n_w:
            per_step = lambda v: v / eager_steps  # noqa: E731
            # THE entry: every launch of the family -- large plain launches, the weight-gradient launches
            # that carry the norm backward's streaming half, and the small launch-bound 1-D-chain launches
            fl_all, ms_all = fl_l + fl_w + fl_wf + fl_t + sfl, ms_l + ms_w + ms_wf + ms_t + sms
            n_all = n_l + n_w + n_wf + n_t + sn
            tf_all = fl_all / (ms_all * 1e-3) / 1e12
            tf = (fl_l + fl_w) / ((ms_l + ms_w) * 1e-3) / 1e12
            binding = binding_roofline(gemm_timers, eager_steps, MFMA_F32_PEAK_TFLOPS, HBM_PEAK_GBS)
            # HBM bytes of the family as rocprofv3 counted them (separate --pmc FETCH_SIZE / WRITE_SIZE
            # passes of this command, gfx950 fetch correction applied): produced by tools/make_profiles.py
            # together with the sha256 of the library that ran; LOADED here, never measured inside this
            # run, and dropped (null) when the library loaded now is not the one that was profiled
            traffic, traffic_note = None, None
            lib_sha = _lib.library_sha256()
            # a set is for one workload: tools/make_profiles.py records the extra bench arguments it ran with
            want = {'pretrain': '', 'semi': '--workload semi', 'saqe': '--workload saqe'}[args.workload]
            cands = []
            for f in sorted(os.listdir(os.path.join(ROOT, 'profiles'))):
                if f.endswith('_pmc_hbm_traffic.json'):
                    if json.load(open(os.path.join(ROOT, 'profiles', f))).get('bench_args', '') == want:
                        cands.append(f)
            if args.batch != 8:
                traffic_note = 'PMC traffic was collected at B = 8 only'
            elif not cands:
                traffic_note = f'no profiles/*_pmc_hbm_traffic.json for the {args.workload} workload'
            else:
                t = json.load(open(os.path.join(ROOT, 'profiles', cands[-1])))
                if t.get('lib_sha256') != lib_sha:
                    traffic_note = (f'profiles/{cands[-1]} was collected with another build of libnesie_hip.so '
                                    f'(sha256 {str(t.get("lib_sha256"))[:12]}.. vs loaded {lib_sha[:12]}..): not quoted')
                else:
                    traffic = {'fetch_corrected_plus_write_bytes_per_step': t['family_bytes_per_step'],
                               'over_algorithmic': t['family_bytes_per_step'] / per_step(by_l + by_w + by_wf + by_t * 2 / 3),   # (PMC rows include the fused weight-gradient launches and the pooled tail's layer-kernel launch)
                               # the family's largest launch by bytes, counted / algorithmic
                               'largest_launch_over_algorithmic': (
                                   (t['largest_launch']['fetch_corrected_bytes'] + t['largest_launch']['write_bytes'])
                                   / max(b_ for tm in (gemm_timers[0], gemm_timers[2]) for (_, b_, *_r) in tm.work)),
                               'worst_launch': t.get('worst_launch'),
                               'source': 'profiles/' + cands[-1], 'lib_sha256': lib_sha, 'measured_in_run': False}
            out['roofline'] = {
                'kernel': 'nesie::pw_fwd_kernel (forward products + input gradients, with the operand '
                          'normalisation / statistics / pooling / norm-backward reduction epilogues) + '
                          'nesie::pw_wgrad_kernel (plain and with the norm backward on its operand load): EVERY '
                          'launch of the step -- the grouped per-seed MLPs of the SA stacks and the MiniPointNets '
                          'and the 1-D per-seed / per-proposal chains -- fp32 MFMA (v_mfma_f32_16x16x4_f32)',
                'bound': 'mfma', 'achieved': tf_all, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tf_all / MFMA_F32_PEAK_TFLOPS,
                # the same launches, each priced against the roofline that binds IT (HBM or MFMA)
                'frac_binding': binding['frac_binding'], 'binding': binding,
                'traffic': traffic, 'traffic_note': traffic_note,
                'algorithmic_bytes_per_step': per_step(by_l + by_w + by_wf + by_t),   # launches over >= 32768 positions
                'algorithmic_hbm_gbs': (by_l + by_w + by_wf + by_t) / ((ms_l + ms_w + ms_wf + ms_t) * 1e-3) / 1e9,
                'pooled_tail_backward': ({'entries_per_step': per_step(n_t), 'ms_per_step': per_step(ms_t),
                                          'tflops': fl_t / (ms_t * 1e-3) / 1e12} if ms_t else None),
                'launches_per_step': per_step(n_all),
                'family_ms_per_step': per_step(ms_all),
                'grids': ('timed alone, persistent grids sized for all 256 CUs; inside the replayed step the FORWARD '
                          f'launches are sized for {budget_in_step} CUs (the next batch\'s sampling kernels hold one '
                          'CU per XCD meanwhile)' + (f', {fwd_budget.get("late")} from the detection head on' if fwd_budget.get('late') else '')
                          if budget_in_step != 256 else 'sized for all 256 CUs'),
                'avg_launch_ms': ms_all / n_all,
                'algorithmic_flops_per_launch': fl_all / n_all,
                # sub-entry: the launches over >= 32768 positions WITHOUT the fused norm-backward ones
                # (what rounds 1-3 quoted as `frac`)
                'large_plain_launch_subset': {
                    'launches_per_step': per_step(n_l + n_w), 'ms_per_step': per_step(ms_l + ms_w),
                    'tflops': tf, 'frac': tf / MFMA_F32_PEAK_TFLOPS,
                    'layer_kernel': {'launches_per_step': per_step(n_l), 'ms_per_step': per_step(ms_l),
                                     'tflops': fl_l / (ms_l * 1e-3) / 1e12 if ms_l else None},
                    'wgrad_kernel': {'launches_per_step': per_step(n_w), 'ms_per_step': per_step(ms_w),
                                     'tflops': fl_w / (ms_w * 1e-3) / 1e12 if ms_w else None}},
                # ... and with the fused weight-gradient + norm-backward launches counted as GEMMs
                # (their streaming half adds time but no FLOPs)
                'large_launches_with_fused_norm_backward': {
                    'launches_per_step': per_step(n_l + n_w + n_wf),
                    'ms_per_step': per_step(ms_l + ms_w + ms_wf),
                    'tflops': (fl_l + fl_w + fl_wf) / ((ms_l + ms_w + ms_wf) * 1e-3) / 1e12,
                    'frac': (fl_l + fl_w + fl_wf) / ((ms_l + ms_w + ms_wf) * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}}
            if n_wf:
                gbs = by_wf / (ms_wf * 1e-3) / 1e9
                out['roofline_fused_norm_backward'] = {
                    'kernel': 'nesie::pw_wgrad_kernel<..., BNB> (nesie_pw_wgrad_bn_backward): weight gradient '
                              'with the BatchNorm + ReLU backward apply pass on its operand load; reads dA, Z, X, '
                              'writes dZ (replaces bn_bwd_apply_kernel + pw_wgrad_kernel: 5 tensor passes -> 4)',
                    'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': gbs / HBM_PEAK_GBS, 'traffic': None,
                    'launches_per_step': per_step(n_wf), 'ms_per_step': per_step(ms_wf),
                    'avg_launch_ms': ms_wf / n_wf, 'algorithmic_bytes_per_launch': by_wf / n_wf,
                    'tflops': fl_wf / (ms_wf * 1e-3) / 1e12}
            if sn:
                out['roofline_small_layers'] = {
                    'kernel': 'the same kernels on the 1-D chains (fused_mlp.Stack1dFn): P = 256 .. 1024 '
                              'positions per scene, one or two tiles per workgroup',
                    'bound': 'latency', 'launches_per_step': per_step(sn), 'ms_per_step': per_step(sms),
                    'avg_launch_us': 1e3 * sms / sn, 'tflops': sfl / (sms * 1e-3) / 1e12}
            # the same FLOPs against the whole step (everything that is not a GEMM counts as lost)
            step_tf = per_step(fl_l + fl_w + fl_wf + sfl) / (ms_per_step * 1e-3) / 1e12
            out['roofline_step'] = {'bound': 'mfma', 'achieved': step_tf, 'peak': MFMA_F32_PEAK_TFLOPS,
                                    'unit': 'TFLOP/s', 'frac': step_tf / MFMA_F32_PEAK_TFLOPS,
                                    'native_gemm_flop_per_step': per_step(fl_l + fl_w + fl_wf + sfl)}
        else:
            out['roofline'] = None
        out['roofline_streaming'] = [e for e in (
            _stream('nesie::bn_pool_bwd_reduce_kernel + bn_pool_bwd_apply_kernel<16> (SA1 MLP tail, '
                    'x (B,128,2048,64))', pool_bwd_timer.mean_ms(), 2, big * 4,
                    extra=3 * pooled_bytes + pooled_bytes // 4),
            _stream('nesie::bn_bwd_reduce_kernel + bn_bwd_apply_kernel<relu> (B,64,2048,64)',
                    bn_bwd_timer.mean_ms(), 5, mid * 4),
            _stream('nesie::bn_bwd_apply_kernel<relu> from the input-gradient kernel\'s partials '
                    '(B,64,2048,64): read da, z; write dz', bn_apply_timer.mean_ms(), 3, mid * 4)) if e]
        rounds = 2047
        out['roofline_latency_bound'] = {
            'kernel': 'nesie::fps_pruned_kernel<16> (D-FPS 40000->2048: 2047 DEPENDENT rounds, one '
                      '1024-thread workgroup per scene; longest single launch, overlapped with the '
                      'previous step on a side stream in graph mode)',
            'bound': 'latency', 'achieved': fps_ms * 1e3 / rounds if fps_ms else None,
            'unit': 'us/round', 'rounds': rounds, 'avg_launch_ms': fps_ms,
            # what a round re-reads: one 20-byte record per bucket test (625 buckets) + the ~12 active
            # buckets of 64 points x 16 B from L2 + their running minima from LDS
            'on_chip_bytes_per_round': 625 * 20 + 12 * 64 * (16 + 4),
            'compulsory_hbm_bytes_per_launch': alg_bytes, 'peak': None, 'frac': None, 'traffic': None}
        if gate is not None:
            out['parity_gate'] = gate
        if args.loss_trace:
            out['loss_trace'] = [float(t) for t in trace]
        if args.other_workloads and world == 1 and args.workload == 'pretrain' and args.batch == 8 and args.graph:
            # configs[3] / [4] in front of the driver: free the headline's model, graphs and pools first
            import gc
            step = model = bucket = loss = None
            gc.collect()
            torch.cuda.empty_cache()
            out['other_workloads'] = [other_workload(device, 'semi', 8, 5, 2, gate=bool(args.parity_gate)),
                                      other_workload(device, 'saqe', 16, 5, 2, gate=bool(args.parity_gate))]
        if args.cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.cpu_batch, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
