"""Detector path: HIP product path on the GPU vs the same model through the CPU oracle."""
import copy

import pytest
import torch

from nesie_amd import kernels
from tests import _small

pytestmark = pytest.mark.gpu


def test_small_model_losses_and_grads_match_cpu_oracle(oracle_kernels, hip_device):
    model = _small.small_model()
    # wide assignment thresholds so every loss term has positives at random init
    model.train_cfg['pos_distance_thr'] = 1.0
    model.train_cfg['neg_distance_thr'] = 1.5
    pts, boxes, labels = _small.small_batch()
    model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
    # discrete decisions on PREDICTED coordinates (vote sampling, grid taps) are replayed from the
    # CPU leg: a last-bit difference may legitimately flip them (oracle/forcing.py)
    _small.force_vote_sampling(model, 'nesie-small')
    _small.force_grid_taps(model, 'nesie-small')
    gmodel = copy.deepcopy(model).to(hip_device)
    with kernels.use_backend(oracle_kernels):
        want_l, want_g = _small.train_step_losses(model, pts, boxes, labels)
    got_l, got_g = _small.train_step_losses(gmodel, pts.to(hip_device), boxes, labels)
    for k in want_l:  # north_star tolerance: 1e-4 for fp32 losses
        assert want_l[k].item() > 0, k
        torch.testing.assert_close(got_l[k], want_l[k], rtol=1e-4, atol=1e-5, msg=k)
    assert set(got_g) == set(want_g)
    # Gradients: the whole flat gradient vector within 1e-3 relative L2; per parameter
    # within 5e-2 of its largest entry (floored at 1e-3 of the global largest).  The
    # per-parameter bound is loose on purpose: with 2 scenes x 64 proposals the
    # BatchNorm layers of the quality head normalise near-constant channels, and the
    # fp32 CPU path itself sits ~4e-3 away from an fp64 evaluation (tools/debug_sp.py).
    flat_w = torch.cat([want_g[n].flatten() for n in sorted(want_g)]).double()
    flat_g = torch.cat([got_g[n].flatten() for n in sorted(want_g)]).double()
    rel_l2 = ((flat_g - flat_w).norm() / flat_w.norm()).item()
    assert rel_l2 < 1e-3, rel_l2
    gmax = flat_w.abs().max().item()
    for n in want_g:
        denom = max(want_g[n].abs().max().item(), 1e-3 * gmax)
        err = (got_g[n] - want_g[n]).abs().max().item() / denom
        assert err < 5e-2, (n, err)


def test_full_config_step_runs_and_is_finite(hip_device):
    from nesie_amd.scenes import make_batch
    from nesie_amd.votenet import build_nesie_votenet
    from nesie_amd.votenet.nesie_head import GTBatch
    torch.manual_seed(0)
    model = build_nesie_votenet().to(hip_device)
    pts, boxes, labels = make_batch(1000, 2)
    gt = GTBatch.collate(boxes, labels, hip_device)
    losses = model.forward_train(pts.to(hip_device), None, gt, None)
    total = model.parse_losses(losses)
    total.backward()
    assert torch.isfinite(total)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n


def test_hip_graph_replay_matches_eager_forward_backward(hip_device):
    """bench.py replays the step as hipGraphs: a captured forward+backward must give the
    loss and gradients of the un-captured one (same weights, pinned jitter noise)."""
    from nesie_amd import dp
    from nesie_amd.scenes import make_batch
    from nesie_amd.votenet import build_nesie_votenet
    from nesie_amd.votenet.nesie_head import GTBatch
    torch.manual_seed(0)
    model = build_nesie_votenet().to(hip_device)
    g = torch.Generator().manual_seed(11)
    model.bbox_head.jitter_noise = tuple(torch.randn(2, 256, 3, generator=g).to(hip_device)
                                         for _ in range(2))
    pts, boxes, labels = make_batch(1000, 2)
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    bucket = dp.FlatGradBucket(model.parameters())
    loss_out = torch.zeros((), device=hip_device)

    def fwd_bwd():
        bucket.zero_()
        losses = model.forward_train(pts, None, gt, None)
        total = model.parse_losses(losses)
        total.backward()
        loss_out.copy_(total.detach())

    fwd_bwd()
    torch.cuda.synchronize()
    want_loss, want_grad = loss_out.item(), bucket.flat.clone()
    side = torch.cuda.Stream(hip_device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fwd_bwd()
    bucket.flat.fill_(123.0)
    loss_out.fill_(-1.0)
    graph.replay()
    torch.cuda.synchronize()
    assert abs(loss_out.item() - want_loss) <= 1e-5 * abs(want_loss)
    rel = ((bucket.flat - want_grad).norm() / want_grad.norm()).item()
    assert rel < 1e-4, rel


def test_ema_teacher_updates_the_registered_buffers_on_the_device(hip_device):
    """The shipped constructor + .to(device): update / swap act on the ema_* buffers of the state
    dict (simi_teacher_hook.py:54-92), on the device."""
    from nesie_amd.votenet import semi
    torch.manual_seed(0)
    model = semi.build_nesie_votenet_semi().to(hip_device)
    key = 'ema_backbone_SA_modules_0_mlps_0_layer0_conv_weight'
    p0 = model.backbone.SA_modules[0].mlps[0].layer0.conv.weight
    assert all(e.is_cuda for e in model.teacher.emas)
    before = model.state_dict()[key].clone()
    with torch.no_grad():
        p0.add_(1.0)
    model.teacher.update(0)
    after = model.state_dict()[key].clone()
    assert after.is_cuda
    torch.testing.assert_close(after, before * 0.999 + (before + 1) * 0.001)
    student = p0.detach().clone()
    model.teacher.swap()
    assert torch.equal(p0, after) and torch.equal(model.state_dict()[key], student)
    model.teacher.swap()
    assert torch.equal(p0, student)


def test_semi_supervised_step_on_gpu(hip_device):
    """Student/teacher step (BASELINE config 4 shape per GPU, reduced batch) runs on the HIP
    path: teacher pseudo labels, re-augmentation, supervised + unsupervised losses, EMA."""
    from nesie_amd.scenes import make_batch
    from nesie_amd.votenet import semi
    from nesie_amd.votenet.nesie_head import GTBatch
    torch.manual_seed(0)
    model = semi.build_nesie_votenet_semi().to(hip_device)
    model.init_label_state(12, 108, hip_device)
    pts, boxes, labels = make_batch(2000, 3)
    g = torch.Generator().manual_seed(1)
    meta_t = semi.AugMeta.random(3, hip_device, g, strong=False)
    meta_s = semi.AugMeta.random(3, hip_device, g, strong=True)
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes[:1], labels[:1], hip_device)
    rows = torch.tensor([5, 17], device=hip_device)
    losses = model.forward_train(meta_s.apply_points(pts), meta_t.apply_points(pts), gt,
                                 [True, False, False], meta_s, meta_t, rows)
    total = model.parse_losses(losses)
    total.backward()
    model.teacher.update(0)
    assert torch.isfinite(total)
    assert len(losses) == 12
    assert float(model.state.ulb_flag.sum()) == 106


def test_saqe_model_losses_match_cpu_oracle(oracle_kernels, hip_device):
    """SAQE head (BASELINE configs[4]): HIP path vs CPU oracle path on a reduced config."""
    from nesie_amd.votenet import build_saqe_votenet
    from nesie_amd.votenet.detector import saqe_votenet_scannet_cfg
    cfg = _small.small_cfg()
    scfg = saqe_votenet_scannet_cfg()
    cfg['bbox_head'].update(angle_loss=scfg['bbox_head']['angle_loss'],
                            angle_pred_loss=scfg['bbox_head']['angle_pred_loss'])
    cfg['head_type'] = 'SAQEHead'
    cfg['train_cfg'].update(pos_distance_thr=1.0, neg_distance_thr=1.5)
    torch.manual_seed(0)
    model = build_saqe_votenet(cfg)
    pts, boxes, labels = _small.small_batch()
    model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
    # discrete decisions on PREDICTED coordinates (vote sampling, grid taps) are replayed from the
    # CPU leg: a last-bit difference may legitimately flip them (oracle/forcing.py)
    _small.force_vote_sampling(model, 'saqe-small')
    _small.force_grid_taps(model, 'saqe-small')
    gmodel = copy.deepcopy(model).to(hip_device)
    with kernels.use_backend(oracle_kernels):
        want_l, want_g = _small.train_step_losses(model, pts, boxes, labels)
    got_l, got_g = _small.train_step_losses(gmodel, pts.to(hip_device), boxes, labels)
    assert set(want_l) == {'vote_loss', 'objectness_loss', 'semantic_loss', 'center_loss',
                           'surface_loss', 'angle_loss', 'angle_pred_loss', 'iou_loss',
                           'iou_pred_loss', 'side_loss'}
    for k in want_l:
        torch.testing.assert_close(got_l[k], want_l[k], rtol=1e-4, atol=1e-5, msg=k)
    flat_w = torch.cat([want_g[n].flatten() for n in sorted(want_g)]).double()
    flat_g = torch.cat([got_g[n].flatten() for n in sorted(want_g)]).double()
    # 5e-3: the error sits in the backbone convs (BatchNorm backward over 2 tiny scenes
    # cancels in fp32 on either device: the two GPU runs agree to 1e-6, tools/debug_saqe.py)
    assert ((flat_g - flat_w).norm() / flat_w.norm()).item() < 5e-3


def test_precomputed_index_chain_is_equivalent(hip_device):
    """bench.py computes the FPS / ball-query chain of the NEXT batch on a side stream; feeding
    those indices must reproduce the in-line forward bit for bit."""
    from nesie_amd.scenes import make_batch
    from nesie_amd.votenet import build_nesie_votenet
    from nesie_amd.votenet.nesie_head import GTBatch
    torch.manual_seed(0)
    model = build_nesie_votenet().to(hip_device)
    g = torch.Generator().manual_seed(11)
    model.bbox_head.jitter_noise = tuple(torch.randn(2, 256, 3, generator=g).to(hip_device)
                                         for _ in range(2))
    pts, boxes, labels = make_batch(1000, 2)
    pts = pts.to(hip_device)
    gt = GTBatch.collate(boxes, labels, hip_device)
    with torch.no_grad():
        a = model.forward_train(pts, None, gt, None)
        pre = model.backbone.sample_and_group_indices(pts)
        b = model.forward_train(pts, None, gt, None, precomputed=pre)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert pre[0]['indices'].shape == (2, 2048) and pre[0]['group_idx'][0].shape == (2, 2048, 64)


@pytest.mark.gpu
def test_batched_score_heads_equal_the_per_head_loop(hip_device):
    """side_pooling.batched_heads (six heads as one broadcast GEMM + one stacked BatchNorm per
    layer) vs calling the six nn.Sequential heads one by one (side_pooling_module.py:314-321):
    outputs, parameter gradients, running statistics and batch counters."""
    import copy
    from nesie_amd.votenet.side_pooling import _score_head, batched_heads, heads_batchable
    torch.manual_seed(11)
    B, P = 3, 96
    heads = [_score_head(166, 18).to(hip_device) for _ in range(6)]
    for h in heads:
        for m in h.modules():
            if hasattr(m, 'running_mean'):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.3)
    ref = copy.deepcopy(heads)
    x = torch.randn(B, 6, 166, P, device=hip_device)
    go = torch.randn(B, 6, 18, P, device=hip_device)
    assert heads_batchable(heads, x[:, 0])
    x1 = x.clone().requires_grad_(True)
    got = batched_heads(heads, x1)
    got.backward(go)
    x2 = x.clone().requires_grad_(True)
    want = torch.stack([ref[i](x2[:, i]) for i in range(6)], 1)
    want.backward(go)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-3, atol=1e-5)
    for a, b in zip(heads, ref):
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            torch.testing.assert_close(pa.grad, pb.grad, rtol=1e-3, atol=2e-5, msg=n)
        for (n, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
            torch.testing.assert_close(ba.float(), bb.float(), rtol=1e-5, atol=1e-6, msg=n)


def test_grouped_mini_pointnets_equal_the_per_net_loop(hip_device):
    """side_pooling.grouped_mini_pointnets (six MiniPointNets as broadcast GEMMs + stacked
    BatchNorms) vs the six nets called one by one on the same first-conv outputs
    (side_pooling_module.py:343-370): outputs, gradients, running statistics."""
    import copy
    from nesie_amd.votenet.side_pooling import (MiniPointNet, grouped_mini_pointnets,
                                                mini_pointnets_groupable)
    torch.manual_seed(12)
    B, K, G = 2, 40, 16
    nets = [MiniPointNet(19, 32, hide_dim=64).to(hip_device) for _ in range(6)]
    for n in nets:
        for m in n.modules():
            if hasattr(m, 'running_mean'):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.3)
    ref = copy.deepcopy(nets)
    c0 = torch.randn(B, 6, 64, K, G, device=hip_device)
    go = torch.randn(B, 6, 32, K, device=hip_device)
    assert mini_pointnets_groupable(nets, c0)
    x1 = c0.clone().requires_grad_(True)
    got = grouped_mini_pointnets(nets, x1)
    got.backward(go)
    x2 = c0.clone().requires_grad_(True)
    want = torch.stack([ref[i](conv0_out=x2[:, i]) for i in range(6)], 1)
    want.backward(go)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    # a ReLU mask or a max-pool winner within rounding of a tie may flip between the two
    # evaluation orders; one flip moves the BatchNorm sums of its whole channel, so the input
    # gradient is compared in norm (a handful of flips among 5e5 entries)
    rel = ((x1.grad - x2.grad).norm() / x2.grad.norm()).item()
    assert rel < 2e-2, rel
    for a, b in zip(nets, ref):
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            if pb.grad is None:        # first_conv[0] is bypassed by conv0_out
                assert pa.grad is None, n
                continue
            # per-channel sums (norm weight/bias) move by one term per flipped mask entry
            # (a bias in front of a norm layer has an analytically zero gradient: rounding noise
            # on both sides, hence the absolute floor)
            err = (pa.grad - pb.grad).norm().item()
            assert err < 2e-2 * pb.grad.norm().item() + 1e-4 * pb.numel() ** 0.5, (n, err)
        for (n, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
            torch.testing.assert_close(ba.float(), bb.float(), rtol=1e-5, atol=1e-6, msg=n)


@pytest.mark.parametrize("variant", ["nesie", "saqe"])
def test_fused_grid_taps_match_the_literal_chain(hip_device, variant):
    """nesie_grid_taps (grid generation + face selection + rotation + 3-NN + inverse-distance
    weights in one launch) vs the literal torch chain generate_grid -> grid_for_side/bbox ->
    _blend_taps (side_pooling_module.py:87-157, 204-225; quelity_estimation_module.py:142-167)."""
    from nesie_amd.votenet.quality_estimation import QualityEstimation
    from nesie_amd.votenet.side_pooling import SidePooling
    torch.manual_seed(21)
    cls = SidePooling if variant == "nesie" else QualityEstimation
    sp = cls(num_class=3, num_heading_bin=1, num_size_cluster=3, mean_size_arr_path=None,
             num_proposal=40, sampling='vote_fps', seed_feat_dim=16).to(hip_device)
    B, K, N = 2, 40, 300
    xyz = (torch.rand(B, N, 3, device=hip_device) * 4).contiguous()
    center = torch.rand(B, K, 3, device=hip_device) * 4
    size = torch.rand(B, K, 3, device=hip_device) + 0.3
    heading = torch.rand(B, K, device=hip_device) * 6.28 - 3.14
    whole = sp.generate_grid(size)
    sets = [('side', sp.grid_for_side(whole, center, heading))]
    if variant == "nesie":
        sets.append(('box', sp.grid_for_bbox(whole, center, heading)))
    for which, grid in sets:
        grid = grid.reshape(B, -1, 3).contiguous()
        idx0, w0, rel0 = sp._blend_taps(xyz, grid, center)
        idx1, w1, rel1 = sp.fused_taps(xyz, center, size, heading, which)
        assert idx1.shape == idx0.shape and idx1.dtype == torch.int32
        torch.testing.assert_close(rel1, rel0, rtol=1e-5, atol=2e-6)
        same = (idx1 == idx0).all(-1)
        assert same.float().mean().item() > 0.995          # near-ties may order differently
        torch.testing.assert_close(w1[same], w0[same], rtol=1e-4, atol=1e-6)


def test_fused_side_decode_matches_side2box(hip_device):
    """nesie_side_decode_forward/backward vs NesieHead.side2box + Integral + the bbox_probs
    softmax evaluated with ATen ops and autograd (nesie_head.py:19-52, 150-209, 255-257)."""
    import torch.nn.functional as F
    from nesie_amd.votenet.nesie_head import SideDecode
    torch.manual_seed(31)
    from nesie_amd.votenet import build_nesie_votenet
    head = build_nesie_votenet().bbox_head.to(hip_device)
    B, K = 3, 70
    reg = torch.randn(B, head.n_reg_outs + 2, K, device=hip_device) * 2.0
    agg = torch.rand(B, K, 3, device=hip_device) * 4
    g_s = torch.randn(B, K, 6, device=hip_device)
    g_b = torch.randn(B, K, 7, device=hip_device)
    r1, a1 = reg.clone().requires_grad_(True), agg.clone().requires_grad_(True)
    probs1, surf1, box1 = SideDecode.apply(r1, a1, head._side_scale, head._side_sign)
    ((surf1 * g_s).sum() + (box1 * g_b).sum()).backward()
    r2, a2 = reg.clone().requires_grad_(True), agg.clone().requires_grad_(True)
    res = head.side2box(a2, r2.transpose(2, 1), {})
    probs2 = F.softmax(r2[:, :head.n_reg_outs].reshape(B, 6, head.reg_max + 1, -1), dim=2)
    ((res['surface_pred'] * g_s).sum() + (res['bbox_preds'] * g_b).sum()).backward()
    torch.testing.assert_close(probs1, probs2.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(surf1, res['surface_pred'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(box1, res['bbox_preds'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a1.grad, a2.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(r1.grad, r2.grad, rtol=1e-4, atol=1e-5)


def _run_children(cmd, env, timeout=300):
    """subprocess.run(capture_output, text) for a bench.py launch that starts ranks of its own: the
    whole process GROUP is killed on a time-out (a hung rank must not outlive the test holding the
    GPU), and the time-out is short enough for the suite to report it instead of going silent."""
    import os
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = proc.communicate()
        raise AssertionError(f'{" ".join(cmd[-8:])} did not finish within {timeout} s; stderr tail: {err[-1500:]}')
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)



def test_two_rank_step_rehearsal_on_one_gpu():
    """bench.py's N > 1 path end to end (graphs, pipelined index chain, flat gradient
    all-reduce between the two graphs, fused AdamW) with two ranks sharing this GPU; gloo stands
    in for RCCL, which refuses two ranks on one device (NESIE_DIST_BACKEND, dp.init_distributed)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NESIE_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    out = _run_children(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', '29533', os.path.join(root, 'bench.py'),
         '--gpus', '2', '--steps', '2', '--warmup', '1', '--cpu-baseline', '0'], env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 prints ONE line
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 16
    assert res['scaling'] == 'weak' and res['value'] > 0 and 'cpu_baseline' not in res


@pytest.mark.parametrize('workload,batch', [('pretrain', 8), ('semi', 3)])
def test_bare_bench_command_starts_its_own_ranks(workload, batch):
    """The driver's command form, ``python bench.py --gpus N`` with WORLD_SIZE unset: the process
    starts N fresh ranks itself (before it touches the GPU) and rank 0 prints the one line with
    the world size and the collective back end.  Two ranks share this GPU, so gloo stands in for
    RCCL (NESIE_DIST_BACKEND)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['NESIE_DIST_BACKEND'] = 'gloo'
    out = _run_children(
        [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup',
         '1', '--cpu-baseline', '0', '--workload', workload, '--batch', str(batch)], env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['world_size'] == 2
    assert res['config']['collective_backend'] == 'gloo'
    assert res['config']['global_batch'] == 2 * batch and res['value'] > 0


def test_rccl_all_reduce_runs_between_the_graph_replays_on_one_gpu():
    """RCCL itself under the captured step: ``bench.py --gpus 1`` with NESIE_FORCE_PG=1 builds a
    world_size = 1 ``nccl`` (= RCCL) process group, so dp.SegmentedAllReduce really launches both
    gradient all-reduces on the communication stream -- the head's segment between the g1a and g1b
    replays, the backbone's before g2 -- for 50 steps.  A one-rank sum divided by 1 is the identity,
    so the loss trajectory must EQUAL the run without a group, step for step and bit for bit: the
    default backward has no unordered float sum (kernels.HipKernels.DETERMINISTIC), so any
    difference would be the collective's doing -- a gradient segment reduced while
    it was still being written, an update that did not wait for the exchange (the reference's
    collective is NCCL through torch.distributed: train.py:132-139,
    nesie-votenet-scannet-train-010.py:143).  Children are fresh processes; nothing that touched the
    GPU is re-executed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items()
            if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NESIE_DIST_BACKEND', 'NESIE_FORCE_PG')}
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    base.pop('NESIE_DETERMINISTIC', None)              # the default mode, whatever the caller's shell says
    runs = {}
    for name, extra in (('plain', {}), ('rccl', dict(NESIE_FORCE_PG='1', MASTER_ADDR='127.0.0.1',
                                                     MASTER_PORT='29541'))):
        out = _run_children(
            [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '47', '--warmup', '3',
             '--batch', '2', '--cpu-baseline', '0', '--parity-gate', '0', '--loss-trace', '1'],
            dict(base, **extra))
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
        assert len(lines) == 1, out.stdout[-2000:]
        runs[name] = json.loads(lines[0])
    plain, rccl = runs['plain'], runs['rccl']
    assert plain['config']['collective_backend'] is None and plain['config']['collectives_per_step'] == 0
    assert rccl['config']['collective_backend'] == 'nccl' and rccl['config']['world_size'] == 1
    assert rccl['config']['collectives_per_step'] == 2 and rccl['config']['hip_graph']
    a, b = plain['loss_trace'], rccl['loss_trace']
    assert len(a) == len(b) == 50 and all(v == v and v > 0 for v in b)
    assert a == b, [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y][:5]
    assert a[-1] < a[0]                                   # ... and the replayed step trains
    print('loss trajectory with / without the RCCL group: first', a[:3], b[:3], 'last', a[-1], b[-1],
          'largest gap', max(abs(x - y) / max(1.0, abs(x)) for x, y in zip(a, b)))


def test_stream_first_layer_equals_the_gemm_path(hip_device):
    """ConvModule's skinny-first-layer path (streaming MFMA kernel + statistics epilogue feeding
    the norm) against the plain conv -> norm path: outputs, running statistics, gradients."""
    from nesie_amd.mmdet3d_ops import pointnet_modules as pm
    torch.manual_seed(3)
    a = pm.ConvModule(4, 64, 1, conv_cfg=dict(type='Conv2d'), norm_cfg=dict(type='BN2d')).to(hip_device)
    b = copy.deepcopy(a)
    x = torch.randn(2, 4, 1024, 32, device=hip_device)          # P = 32768 per scene
    assert pm.stream_conv_eligible(x, a.conv, a.norm)
    g = torch.randn(2, 64, 1024, 32, device=hip_device)
    ya = a(x)
    (ya * g).sum().backward()
    real = pm.stream_conv_eligible
    pm.stream_conv_eligible = lambda *_: False
    try:
        yb = b(x)
        (yb * g).sum().backward()
    finally:
        pm.stream_conv_eligible = real
    torch.testing.assert_close(ya, yb, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(a.norm.running_mean, b.norm.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.norm.running_var, b.norm.running_var, rtol=1e-5, atol=1e-6)
    for pa, pb in zip(a.parameters(), b.parameters()):
        scale = max(float(pb.grad.abs().max()), 1e-3)
        assert float((pa.grad - pb.grad).abs().max()) / scale < 1e-3


def test_folded_norm_backward_matches_the_separate_apply_pass(hip_device):
    """NESIE_FOLD_NORM_BWD on vs off on the reduced model, one training step from the same state:
    the norm backward applied inside its consumer -- the weight gradient (SA stacks, 1-D chains,
    MiniPointNet norm 1 with the row-bias gradient) or the blend backward (MiniPointNet norm 0,
    inside the one autograd node fused_mlp.BlendMiniHeadFn) -- gives the gradients of the separate
    apply pass."""
    import copy
    from nesie_amd import kernels
    from nesie_amd.mmdet3d_ops import fused_mlp
    model = _small.small_model().to(hip_device).train()
    pts, boxes, labels = _small.small_batch()
    pts = pts.to(hip_device)
    model.bbox_head.jitter_noise = tuple(t.to(hip_device) for t in _small.fixed_noise(2, 32))
    _small.force_vote_sampling(model, 'fold')
    _small.force_grid_taps(model, 'fold')
    seen = []
    backend = type(kernels.backend_for(pts))
    real = backend.blend_conv_backward

    def spy(self, *a, **kw):
        seen.append(kw.get('bnb') is not None)
        return real(self, *a, **kw)
    out = []
    for fold in (False, True):
        fused_mlp.FOLD_NORM_BWD = fold
        backend.blend_conv_backward = spy
        try:
            m = copy.deepcopy(model)
            out.append(_small.train_step_losses(m, pts, boxes, labels))
        finally:
            fused_mlp.FOLD_NORM_BWD = True
            backend.blend_conv_backward = real
    n = len(seen) // 2
    assert n >= 2 and not any(seen[:n]) and all(seen[n:]), seen       # the hand-over really happened
    (l0, g0), (l1, g1) = out
    for k in l0:
        torch.testing.assert_close(l1[k], l0[k], rtol=1e-6, atol=1e-7, msg=k)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        scale = max(float(g0[k].abs().max()), 1e-3 * gmax)
        assert float((g1[k] - g0[k]).abs().max()) < 2e-4 * scale, k
