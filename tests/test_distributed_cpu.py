"""N > 1 path on CPU: two gloo ranks, bucketed gradient all-reduce launched from autograd hooks
(afcm_amd.distributed.GradientBuckets) must reproduce the single-process gradient of the full batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.LeakyReLU(0.2), torch.nn.Linear(64, 64), torch.nn.LeakyReLU(0.2),
                               torch.nn.Linear(64, 4), torch.nn.Linear(4, 4, bias=False))


def _worker(rank, world, port, bucket_bytes, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    if rank != 0:                      # broadcast must overwrite this
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    m[5].weight.requires_grad_(True)
    buckets = GradientBuckets(m.parameters(), bucket_bytes=bucket_bytes)
    buckets.broadcast_parameters(m)
    torch.manual_seed(123)
    x = torch.randn(8, 16)
    y = torch.randn(8, 4)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for it in range(2):                # two iterations: bucket state must reset
        for p in m.parameters():
            p.grad = None
        out = m[:5](xs)                # the last layer is unused -> its parameter gets no gradient (reduced as zeros)
        loss = (out - ys).abs().mean()
        loss.backward()
        buckets.finish()
    torch.save({n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()}, os.path.join(out_dir, f'g{rank}.pt'))
    # third iteration through finish_flat(): the optimizer path that reads the summed buckets in place
    for p in m.parameters():
        p.grad = None
    (m[:5](xs) - ys).abs().mean().backward()
    views, scale = buckets.finish_flat()
    assert abs(scale - 1.0 / world) < 1e-12
    torch.save({n: (views[p] * scale).clone() for n, p in m.named_parameters()}, os.path.join(out_dir, f'f{rank}.pt'))
    torch.save(buckets.num_buckets, os.path.join(out_dir, f'nb{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [1 << 20, 4096])
def test_bucketed_allreduce_matches_full_batch(tmp_path, bucket_bytes):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, bucket_bytes, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / 'g0.pt')
    g1 = torch.load(tmp_path / 'g1.pt')
    m = _model()
    torch.manual_seed(123)
    x = torch.randn(8, 16)
    y = torch.randn(8, 4)
    # mean over ranks of per-rank mean losses == full-batch mean loss (equal shard sizes)
    loss = (m[:5](x) - y).abs().mean()
    loss.backward()
    for n, p in m.named_parameters():
        if p.grad is None:
            assert g0[n] is None and g1[n] is None
            continue
        assert torch.allclose(g0[n], p.grad, atol=1e-6), n
        assert torch.allclose(g1[n], p.grad, atol=1e-6), n
    f0 = torch.load(tmp_path / 'f0.pt')
    for n, p in m.named_parameters():
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(f0[n], want, atol=1e-6), ('finish_flat', n)
    nb = torch.load(tmp_path / 'nb0.pt')
    assert nb >= (2 if bucket_bytes == 4096 else 1)


def test_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    b = GradientBuckets(m.parameters())
    (m(torch.randn(2, 16)).sum()).backward()
    g = m[0].weight.grad.clone()
    b.finish()
    assert torch.equal(g, m[0].weight.grad)
