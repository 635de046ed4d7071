"""Data-parallel path on CPU: world_size 2, gloo.  Each rank runs the step on its own
scene shard through the CPU oracle; after the flat-bucket all-reduce every rank must hold
the mean of the two per-rank gradients (BatchNorm is per-rank, as in the reference's DDP
use with plain BN)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import oracle
    from nesie_amd import dp, kernels
    from nesie_amd.votenet.nesie_head import GTBatch
    from tests import _small
    r, w, _ = dp.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    model = _small.small_model(seed=0)  # identical weights on every rank
    model.train_cfg['pos_distance_thr'] = 1.0
    model.train_cfg['neg_distance_thr'] = 1.5
    pts, boxes, labels = _small.small_batch(seed=7, batch=4, n=2048)  # global batch
    lo, hi = dp.shard_range(4, rank, world)
    model.bbox_head.jitter_noise = tuple(t[lo:hi] for t in _small.fixed_noise(4, 32))
    bucket = dp.FlatGradBucket(model.parameters())
    with kernels.use_backend(oracle.OracleKernels()):
        bucket.zero_()
        gt = GTBatch.collate(boxes[lo:hi], labels[lo:hi], pts.device)
        losses = model.forward_train(pts[lo:hi], None, gt, None)
        model.parse_losses(losses).backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    torch.save(dict(local=local, reduced=bucket.flat.clone()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_is_mean_of_rank_gradients(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "rank0.pt")
    b = torch.load(tmp_path / "rank1.pt")
    assert 2640477 <= a["local"].numel() < 2640477 + 4 * 221      # (+ the padding that keeps every parameter 16-byte aligned)
    assert not torch.equal(a["local"], b["local"])  # different shards -> different grads
    want = (a["local"] + b["local"]) / 2
    torch.testing.assert_close(a["reduced"], want, rtol=1e-6, atol=1e-7)
    assert torch.equal(a["reduced"], b["reduced"])  # every rank holds the same mean


def test_shard_range_partitions_the_global_batch():
    from nesie_amd import dp
    spans = [dp.shard_range(64, r, 8) for r in range(8)]
    assert spans[0] == (0, 8) and spans[-1] == (56, 64)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
    with pytest.raises(AssertionError):
        dp.shard_range(10, 0, 4)


def test_flat_bucket_views_track_autograd():
    from nesie_amd import dp
    lin = torch.nn.Linear(3, 2)
    bucket = dp.FlatGradBucket(lin.parameters())
    lin(torch.ones(4, 3)).sum().backward()
    assert bucket.flat.abs().sum() > 0
    assert lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    bucket.zero_()
    assert bucket.flat.abs().sum() == 0 and lin.weight.grad.abs().sum() == 0
    assert bucket.offsets == [0, 8] and bucket.nbytes() == (8 + 4) * 4      # (6 + 2 floats, each start 16-byte aligned)


def test_flat_train_state_matches_per_tensor_adamw():
    """dp.FlatTrainState: stolen gradients gathered into the flat vector + AdamW/clip on ONE
    flat parameter == zeroed .grad accumulation + per-tensor AdamW/clip (the reference's
    optimizer config: AdamW, grad_clip max_norm=10)."""
    import copy
    from nesie_amd import dp
    torch.manual_seed(3)
    a = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3),
                            torch.nn.Linear(3, 3))
    unused = torch.nn.Parameter(torch.randn(4))       # a parameter the loss never reaches
    a.register_parameter('unused', unused)
    b = copy.deepcopy(a)
    state = dp.FlatTrainState(a.parameters())
    opt_a = torch.optim.AdamW([state.flat_param], lr=1e-2, weight_decay=0.05)
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=0.05)
    for p in b.parameters():
        p.grad = torch.zeros_like(p)
    for it, max_norm in enumerate([1e9, 1e9, 0.05, 0.05]):
        x = torch.randn(7, 6)
        state.begin()
        a[:3](x).square().sum().backward()
        state.collect()
        for p in b.parameters():
            p.grad.zero_()
        b[:3](x).square().sum().backward()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert pa.grad.data_ptr() >= state.flat.data_ptr()
            assert torch.equal(pa.grad, pb.grad)
        state.all_reduce_mean()
        na = torch.nn.utils.clip_grad_norm_([state.flat_param], max_norm)
        nb = torch.nn.utils.clip_grad_norm_(b.parameters(), max_norm)
        torch.testing.assert_close(na, nb, rtol=1e-6, atol=0)
        opt_a.step()
        opt_b.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert pa.data_ptr() >= state.flat_param.data_ptr()
            if max_norm > 1:
                assert torch.equal(pa, pb), it          # no clipping: bit-identical
            else:
                torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7)
    assert a.state_dict().keys() == b.state_dict().keys()


def test_two_phase_backward_equals_one_backward(oracle_kernels):
    """dp.backward_head + backward_rest (the cut bench.py's g1a / g1b graphs are made of) give
    the gradients of one ``backward()``: every path from the loss into the backbone runs
    through ``model.head_inputs``."""
    import copy
    from nesie_amd import dp, kernels
    from nesie_amd.votenet.nesie_head import GTBatch
    from tests import _small
    model = _small.small_model()
    model.train_cfg.update(pos_distance_thr=1.0, neg_distance_thr=1.5)
    model.bbox_head.jitter_noise = _small.fixed_noise(2, 32)
    twin = copy.deepcopy(model)
    pts, boxes, labels = _small.small_batch()
    gt = GTBatch.collate(boxes, labels, pts.device)
    with kernels.use_backend(oracle_kernels):
        twin.parse_losses(twin.forward_train(pts, None, gt, None)).backward()
        state = dp.FlatTrainState(model.parameters())
        n_bb, e_bb = state.split_after(model.backbone.parameters())
        assert 0 < n_bb < len(state.params) and 0 < e_bb < state.flat.numel()
        state.begin()
        model.keep_head_inputs = True
        total = model.parse_losses(model.forward_train(pts, None, gt, None))
        seen = []
        dp.backward_in_two_phases(
            total, model.take_head_inputs(), state.params[n_bb:], state.params[:n_bb],
            between=lambda: (state.collect(n_bb, None),
                             seen.append([p.grad is None for p in state.params[:n_bb]])))
        state.collect(0, n_bb)
    assert all(seen[0])                     # phase 1 did not touch the backbone
    for (n, p), q in zip(model.named_parameters(), twin.parameters()):
        assert p.grad.data_ptr() >= state.flat.data_ptr(), n
        want = q.grad if q.grad is not None else torch.zeros_like(q)
        torch.testing.assert_close(p.grad, want, rtol=1e-6, atol=1e-7, msg=n)


def _semi_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import oracle
    from nesie_amd import dp, kernels
    from nesie_amd.votenet import semi
    from nesie_amd.votenet.nesie_head import GTBatch
    from tests import _small
    dp.init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = semi.build_nesie_votenet_semi(_small.small_cfg())   # identical weights on every rank
    with torch.no_grad():   # a teacher that passes some proposals at random init
        model.bbox_head.conv_pred.conv_cls.bias[1] += 2.3
        model.bbox_head.conv_pred.conv_cls.bias[2] += 0.9
    model.teacher.resync()
    model.init_label_state(12, 108, torch.device('cpu'))       # every rank: its own state
    pts, boxes, labels = _small.small_batch(seed=7, batch=6, n=2048)   # global batch
    lo, hi = dp.shard_range(6, rank, world)                    # 1 labeled + 2 unlabeled per rank
    g = torch.Generator().manual_seed(1 + rank)
    meta_t = semi.AugMeta.random(3, pts.device, g, strong=False)
    meta_s = semi.AugMeta.random(3, pts.device, g, strong=True)
    model.bbox_head.jitter_noise = tuple(t[lo:hi] for t in _small.fixed_noise(6, 32))
    gt = GTBatch.collate(boxes[lo:lo + 1], labels[lo:lo + 1], pts.device)
    rows = torch.tensor([10 * rank + 5, 10 * rank + 7])        # this rank's unlabeled scenes
    state = dp.FlatTrainState(model.parameters())
    opt = torch.optim.AdamW([state.flat_param], lr=1e-3, weight_decay=0.01)
    n_bb, e_bb = state.split_after(model.backbone.parameters())
    comm = dp.SegmentedAllReduce(state.flat)
    assert comm.world == world
    model.keep_head_inputs = True
    with kernels.use_backend(oracle.OracleKernels()):
        state.begin()
        losses = model.forward_train(meta_s.apply_points(pts[lo:hi]), meta_t.apply_points(pts[lo:hi]),
                                     gt, [True, False, False], meta_s, meta_t, rows)
        total = model.parse_losses(losses)
        local = {}
        def between():
            state.collect(n_bb, None)
            local['head'] = state.flat[e_bb:].clone()
            comm.launch(e_bb, state.flat.numel())
        dp.backward_in_two_phases(total, model.take_head_inputs(), state.params[n_bb:],
                                  state.params[:n_bb], between)
        state.collect(0, n_bb)
        local['backbone'] = state.flat[:e_bb].clone()
        comm.launch(0, e_bb)
        assert comm.wait() == [(e_bb, state.flat.numel()), (0, e_bb)]
    reduced = state.flat.clone()
    torch.nn.utils.clip_grad_norm_([state.flat_param], max_norm=10)
    opt.step()
    model.teacher.update(1000)
    torch.save(dict(local=torch.cat([local['backbone'], local['head']]), reduced=reduced,
                    params=state.flat_param.detach().clone(),
                    ema=torch.cat([b.flatten() for b in model.teacher.emas]),
                    ulb_flag=model.state.ulb_flag.clone(), ulb_list=model.state.ulb_list.clone(),
                    loss=float(total)), os.path.join(out_dir, f"semi{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_semi_supervised_step(tmp_path):
    """The student/teacher step on two ranks (gloo): per-rank scenes, augmentation and
    pseudo-label state; the gradient crosses ranks in two segments (head first, then backbone:
    dp.SegmentedAllReduce); after clip + AdamW + EMA every rank holds the same student and the
    same teacher, and each rank's pseudo-label state shows only its own unlabeled rows."""
    port = _free_port()
    mp.spawn(_semi_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(tmp_path / "semi0.pt")
    b = torch.load(tmp_path / "semi1.pt")
    assert a["loss"] != b["loss"] and not torch.equal(a["local"], b["local"])
    torch.testing.assert_close(a["reduced"], (a["local"] + b["local"]) / 2, rtol=1e-6, atol=1e-7)
    assert torch.equal(a["reduced"], b["reduced"])
    assert torch.equal(a["params"], b["params"]) and torch.equal(a["ema"], b["ema"])
    # per-rank pseudo-label state: rank r marked rows 10 r + {5, 7} and nothing else
    for r, s in enumerate((a, b)):
        touched = (s["ulb_flag"] == 0).nonzero().flatten().tolist()
        assert touched == [10 * r + 5, 10 * r + 7], touched
    assert not torch.equal(a["ulb_list"], b["ulb_list"])


def test_bench_launcher_returns_the_first_failing_ranks_code_promptly(tmp_path):
    """`python bench.py --gpus N` without a launcher: a rank that dies while the others are still
    inside a collective must end the job with ITS exit code within seconds (the launcher polls every
    child; it used to wait for rank 0 first, i.e. for the collective time-out)."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location(
        'bench_launcher', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    script = tmp_path / 'rank.py'
    script.write_text(
        'import os, sys, time\n'
        'assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["WORLD_SIZE"] == "3"\n'
        'r = int(os.environ["RANK"])\n'
        'open(sys.argv[1] + f"/started{r}", "w").close()\n'
        'if r == 2:\n'
        '    time.sleep(0.5); sys.exit(7)\n'
        'time.sleep(120)\n')
    t0 = time.monotonic()
    rc = bench.spawn_ranks(3, [str(tmp_path)], script=str(script))
    assert rc == 7
    assert time.monotonic() - t0 < 30
    assert all((tmp_path / f'started{r}').exists() for r in range(3))
    ok = tmp_path / 'ok.py'
    ok.write_text('import sys; sys.exit(0)\n')
    assert bench.spawn_ranks(2, [], script=str(ok)) == 0
