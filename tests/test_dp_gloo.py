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
