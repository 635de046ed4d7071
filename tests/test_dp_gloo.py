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
    assert a["local"].numel() == 2640477
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
    assert bucket.nbytes() == (6 + 2) * 4


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
