"""N > 1 path on CPU: two gloo ranks, bucketed gradient all-reduce launched from autograd hooks
(afcm_amd.distributed.GradientBuckets) must reproduce the single-process gradient of the full batch --
with one backward per iteration (the G update), with the reference's two-backward D update
(models/comodgan_model.py:128-149: loss_D_fake.backward(), then (loss_D_real + R1).backward()) in each of its
three spellings (no_sync, passes=2, undeclared), with a parameter that only one rank uses, and with 16-bit buckets."""
import os
import socket
import sys
import warnings

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


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _worker(rank, world, port, bucket_bytes, out_dir):
    _init(rank, world, port)
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
        out = m[:5](xs)                # the last layer is unused on every rank -> its parameter keeps .grad = None
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
    assert m[5].weight not in views                      # unused everywhere: skipped like torch.optim.Adam skips .grad = None
    torch.save({n: (views[p] * scale).clone() for n, p in m.named_parameters() if p in views}, os.path.join(out_dir, f'f{rank}.pt'))
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
        if p.grad is None:
            assert n not in f0
            continue
        assert torch.allclose(f0[n], p.grad, atol=1e-6), ('finish_flat', n)
    nb = torch.load(tmp_path / 'nb0.pt')
    assert nb >= (2 if bucket_bytes == 4096 else 1)


def _d_losses(m, xf, xr, lambda_r1=10.0):
    """The reference's D update in miniature (comodgan_model.py:128-149): softplus(D(fake)).mean() is back-propagated first,
    then softplus(-D(real)).mean() + lambda_r1 * R1, R1 from a double backward through D."""
    loss_fake = torch.nn.functional.softplus(m(xf)).mean()
    xr = xr.detach().requires_grad_(True)
    logits = m(xr)
    loss_real = torch.nn.functional.softplus(-logits).mean()
    r1, = torch.autograd.grad([logits.sum()], [xr], create_graph=True)
    return loss_fake, loss_real + lambda_r1 * 0.5 * r1.square().sum(1).mean()


def _worker_two_backward(rank, world, port, mode, bucket_bytes, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    m = _model().double()
    buckets = GradientBuckets(m.parameters(), bucket_bytes=bucket_bytes, passes=2 if mode == 'passes' else 1)
    buckets.broadcast_parameters(m)
    torch.manual_seed(7)
    xf, xr = torch.randn(8, 16, dtype=torch.float64), torch.randn(8, 16, dtype=torch.float64)
    sl = slice(rank * 4, (rank + 1) * 4)
    caught = []
    for it in range(3):                # several iterations: per-iteration state must reset
        for p in m.parameters():
            p.grad = None
        a, b = _d_losses(m, xf[sl], xr[sl])
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            if mode == 'no_sync':
                with buckets.no_sync():
                    a.backward()
            else:
                a.backward()
            b.backward()
            if it % 2 == 0:
                buckets.finish()
                got = {n: p.grad.clone() for n, p in m.named_parameters()}
            else:
                views, scale = buckets.finish_flat()
                got = {n: (views[p] * scale).clone() for n, p in m.named_parameters()}
            caught += [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    torch.save(got, os.path.join(out_dir, f'g{rank}.pt'))
    torch.save(caught, os.path.join(out_dir, f'w{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['no_sync', 'passes', 'undeclared'])
@pytest.mark.parametrize('bucket_bytes', [1 << 20, 2048])
def test_two_backward_d_update_matches_full_batch(tmp_path, mode, bucket_bytes):
    """with the hooks launching after ONE accumulation the second backward overwrote a bucket in flight
    (max error 0.71 on a gradient of scale 1.1).  All three spellings must give the full-batch gradient at 1e-6 (here: 1e-12,
    float64) on every rank."""
    world = 2
    mp.spawn(_worker_two_backward, args=(world, _free_port(), mode, bucket_bytes, str(tmp_path)), nprocs=world, join=True)
    m = _model().double()
    torch.manual_seed(7)
    xf, xr = torch.randn(8, 16, dtype=torch.float64), torch.randn(8, 16, dtype=torch.float64)
    a, b = _d_losses(m, xf, xr)
    a.backward()
    b.backward()
    for r in range(world):
        g = torch.load(tmp_path / f'g{r}.pt')
        for n, p in m.named_parameters():
            assert (g[n] - p.grad).abs().max().item() <= 1e-12 * max(1.0, p.grad.abs().max().item()), (mode, r, n)
        w = torch.load(tmp_path / f'w{r}.pt')
        if mode == 'undeclared':
            assert any('accumulated after its bucket' in s for s in w)      # loud about the extra collective
        else:
            assert not w, w


def _worker_partial(rank, world, port, static_graph, comm_dtype, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    buckets = GradientBuckets(m.parameters(), bucket_bytes=4096, static_graph=static_graph,
                              comm_dtype=getattr(torch, comm_dtype) if comm_dtype else None)
    buckets.broadcast_parameters(m)
    torch.manual_seed(5)
    x, y = torch.randn(8, 16), torch.randn(8, 4)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for it in range(3):
        for p in m.parameters():
            p.grad = None
        # the last layer runs on rank 0 only: rank 1 has .grad = None for it and must still receive the reduced gradient
        out = m(xs) if rank == 0 else m[:5](xs)
        (out - ys).abs().mean().backward()
        if it < 2:
            buckets.finish()
            got = {n: p.grad.clone() for n, p in m.named_parameters()}
        else:
            views, scale = buckets.finish_flat()
            got = {n: (views[p] * scale).clone() for n, p in m.named_parameters()}
    torch.save(got, os.path.join(out_dir, f'g{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('static_graph,comm_dtype,tol', [(True, None, 1e-6), (False, None, 1e-6), (True, 'bfloat16', 2e-2)])
def test_parameter_used_on_one_rank_and_16bit_buckets(tmp_path, static_graph, comm_dtype, tol):
    """a parameter with .grad = None on some ranks gets the reduced gradient on all of them, through both
    finish() and finish_flat().  comm_dtype=bfloat16 halves the bytes on the wire: the reduced gradient then carries bf16
    rounding, relative 2^-8 per addend -> 2e-2 of the gradient's scale is the stated tolerance."""
    world = 2
    mp.spawn(_worker_partial, args=(world, _free_port(), static_graph, comm_dtype, str(tmp_path)), nprocs=world, join=True)
    m = _model()
    torch.manual_seed(5)
    x, y = torch.randn(8, 16), torch.randn(8, 4)
    # rank 0: mean |m(x0) - y0| through all six layers; rank 1: through five; averaged over ranks
    ((m(x[:4]) - y[:4]).abs().mean() * 0.5 + (m[:5](x[4:]) - y[4:]).abs().mean() * 0.5).backward()
    for r in range(world):
        g = torch.load(tmp_path / f'g{r}.pt')
        for n, p in m.named_parameters():
            assert (g[n] - p.grad).abs().max().item() <= tol * max(1.0, p.grad.abs().max().item()), (r, n)


def test_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    b = GradientBuckets(m.parameters())
    (m(torch.randn(2, 16)).sum()).backward()
    g = m[0].weight.grad.clone()
    b.finish()
    assert torch.equal(g, m[0].weight.grad)
    with b.no_sync():
        pass


def _worker_rebucket(rank, world, port, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    torch.manual_seed(0)
    # `first` is registered LAST but runs FIRST in forward, so its gradient arrives last (the generator's mapping network)
    body = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4))
    first = torch.nn.Linear(8, 8)
    params = list(body.parameters()) + list(first.parameters())
    buckets = GradientBuckets(params, bucket_bytes=1)                       # one parameter per bucket
    fired, log = [0], []
    on_grad, launch = buckets._on_grad, buckets._launch
    for h in buckets._hooks:
        h.remove()
    def counting_on_grad(p):
        fired[0] += 1
        on_grad(p)
    def logging_launch(b):
        log.append(fired[0])
        launch(b)
    buckets._launch = logging_launch
    buckets._hooks = [p.register_post_accumulate_grad_hook(counting_on_grad) for p in params]
    x = torch.randn(4, 8)
    per_iter = []
    for it in range(3):
        for p in params:
            p.grad = None
        fired[0] = 0
        del log[:]
        body(first(x)).square().mean().backward()
        buckets.finish()
        per_iter.append(list(log))
    full = [p.grad.clone() for p in params]
    torch.save(dict(per_iter=per_iter, n=len(params), grads=full), os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


def test_buckets_are_rebuilt_in_gradient_arrival_order(tmp_path):
    """Collectives go out in bucket order, so a bucket that completes last must not come first.  Iteration 1 (reverse registration
    order: the late `first` layer leads) can issue nothing before the end of backward; from iteration 2 on the buckets follow the
    observed arrival order and each goes out as soon as its own gradient is final.  Gradients stay those of the plain model."""
    world = 2
    mp.spawn(_worker_rebucket, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = torch.load(tmp_path / 'r0.pt')
    n = r['n']
    it1, it2, it3 = r['per_iter']
    assert len(it1) == len(it2) == len(it3) == n
    assert min(it1) >= n - 1                    # nothing before (almost) every gradient had arrived
    assert it2 == list(range(1, n + 1)) == it3  # bucket k issued right after the k-th gradient
    r1 = torch.load(tmp_path / 'r1.pt')
    for a, b in zip(r['grads'], r1['grads']):
        assert torch.equal(a, b)


def _worker_asymmetric(rank, world, port, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    m = _model().double()
    buckets = GradientBuckets(m.parameters(), bucket_bytes=2048, static_graph=False)
    buckets.broadcast_parameters(m)
    torch.manual_seed(7)
    xf, xr = torch.randn(8, 16, dtype=torch.float64), torch.randn(8, 16, dtype=torch.float64)
    sl = slice(rank * 4, (rank + 1) * 4)
    for it in range(3):
        for p in m.parameters():
            p.grad = None
        a, b = _d_losses(m, xf[sl], xr[sl])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if rank == 0:           # undeclared second pass on rank 0 only; rank 1 back-propagates the sum in one pass
                a.backward()
                b.backward()
            else:
                (a + b).backward()
            buckets.finish()
    torch.save({n: p.grad.clone() for n, p in m.named_parameters()}, os.path.join(out_dir, f'g{rank}.pt'))
    dist.destroy_process_group()


def test_rank_dependent_extra_pass_is_settled_globally_without_static_graph(tmp_path):
    """the decision to reduce a bucket again must be the same on every rank.  With static_graph=False the marks ride in
    the flag vector: rank 0's undeclared second backward makes BOTH ranks repeat the bucket (a per-rank decision would issue a
    collective the peer does not -- a hang or a size mismatch)."""
    world = 2
    mp.spawn(_worker_asymmetric, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m = _model().double()
    torch.manual_seed(7)
    xf, xr = torch.randn(8, 16, dtype=torch.float64), torch.randn(8, 16, dtype=torch.float64)
    a, b = _d_losses(m, xf, xr)
    (a + b).backward()
    for r in range(world):
        g = torch.load(tmp_path / f'g{r}.pt')
        for n, p in m.named_parameters():
            assert (g[n] - p.grad).abs().max().item() <= 1e-12 * max(1.0, p.grad.abs().max().item()), (r, n)


def _worker_newly_used(rank, world, port, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    buckets = GradientBuckets(m.parameters(), bucket_bytes=4096, static_graph=True)
    buckets.broadcast_parameters(m)
    torch.manual_seed(5)
    x, y = torch.randn(8, 16), torch.randn(8, 4)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    state = []
    for it in range(4):
        for p in m.parameters():
            p.grad = None
        # iterations 0, 1: the last layer is unused everywhere; from iteration 2 on rank 0 uses it
        out = m(xs) if (rank == 0 and it >= 2) else m[:5](xs)
        (out - ys).abs().mean().backward()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            buckets.finish()
        state.append(None if m[5].weight.grad is None else m[5].weight.grad.clone())
    torch.save(state, os.path.join(out_dir, f's{rank}.pt'))
    dist.destroy_process_group()


def test_finish_never_leaves_an_unreduced_gradient_on_one_rank(tmp_path):
    """with static_graph=True the used-parameter mask lags one iteration.  In the iteration where a parameter first
    receives a gradient on ONE rank, finish() must clear it there too (every rank skips the same set); once the mask has caught up
    every rank holds the same reduced gradient."""
    world = 2
    mp.spawn(_worker_newly_used, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    s0, s1 = torch.load(tmp_path / 's0.pt'), torch.load(tmp_path / 's1.pt')
    for it in range(4):
        assert (s0[it] is None) == (s1[it] is None), it           # never a gradient on one rank only
        if s0[it] is not None:
            assert torch.equal(s0[it], s1[it])
    assert s0[0] is None and s0[1] is None and s0[3] is not None


def _worker_ragged(rank, world, port, sizes, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    m = _model()
    buckets = GradientBuckets(m.parameters(), bucket_bytes=8192)
    buckets.broadcast_parameters(m)
    torch.manual_seed(321)
    total = sum(sizes)
    x, y = torch.randn(total, 16), torch.randn(total, 4)
    lo = sum(sizes[:rank])
    xs, ys = x[lo:lo + sizes[rank]], y[lo:lo + sizes[rank]]
    for it in range(3):                # three iterations: the bucket rebuild after the first one must keep working with four ranks
        for p in m.parameters():
            p.grad = None
        # the buckets average over RANKS (1 / world); a rank holding n_r of the N samples weighs its mean loss by n_r world / N so that
        # the result is the mean over SAMPLES, whatever the split (the last, short batch of an epoch: data/cmsr_dataset.py has no drop_last)
        loss = (m(xs) - ys).abs().mean() * (sizes[rank] * world / total)
        loss.backward()
        buckets.finish()
    torch.save({n: p.grad.clone() for n, p in m.named_parameters()}, os.path.join(out_dir, f'g{rank}.pt'))
    torch.save(buckets.num_buckets, os.path.join(out_dir, f'nb{rank}.pt'))
    dist.destroy_process_group()


def test_four_ranks_with_ragged_per_rank_batches_match_the_full_batch(tmp_path):
    """World size 4, per-rank batches 3 / 2 / 2 / 1: after finish() every rank holds the gradient of the mean loss
    over all 8 samples -- the loss of a rank is weighted by its share of the samples, the buckets average over ranks."""
    world, sizes = 4, (3, 2, 2, 1)
    port = _free_port()
    mp.spawn(_worker_ragged, args=(world, port, sizes, str(tmp_path)), nprocs=world, join=True)
    m = _model()
    torch.manual_seed(321)
    x, y = torch.randn(sum(sizes), 16), torch.randn(sum(sizes), 4)
    (m(x) - y).abs().mean().backward()
    want = {n: p.grad for n, p in m.named_parameters()}
    g = [torch.load(tmp_path / f'g{r}.pt') for r in range(world)]
    for n, w in want.items():
        for r in range(world):
            assert torch.allclose(g[r][n], w, rtol=1e-5, atol=1e-7), (n, r, (g[r][n] - w).abs().max().item())
            assert torch.equal(g[r][n], g[0][n]), f'{n}: rank {r} differs from rank 0'
    assert torch.load(tmp_path / 'nb0.pt') > 1


def _worker_world8(rank, world, port, sizes, out_dir):
    _init(rank, world, port)
    from afcm_amd.distributed import GradientBuckets
    torch.manual_seed(0)
    # a deeper model than _model(): enough parameters for several regular buckets AND a tail of small ones; the last layer is used by
    # the even ranks only except on rank 3, which never uses `extra` at all
    m = torch.nn.Sequential(torch.nn.Linear(16, 96), torch.nn.LeakyReLU(0.2), torch.nn.Linear(96, 96), torch.nn.LeakyReLU(0.2),
                            torch.nn.Linear(96, 64), torch.nn.LeakyReLU(0.2), torch.nn.Linear(64, 4))
    extra = torch.nn.Linear(4, 4, bias=False)
    params = list(extra.parameters()) + list(m.parameters())      # registered first, produced first: the arrival-order rebuild moves it
    buckets = GradientBuckets(params, bucket_bytes=8192, static_graph=False, min_bucket_bytes=256)
    buckets.broadcast_parameters(m)
    buckets.broadcast_parameters(extra)
    torch.manual_seed(99)
    total = sum(sizes)
    x, y = torch.randn(total, 16), torch.randn(total, 4)
    lo = sum(sizes[:rank])
    xs, ys = x[lo:lo + sizes[rank]], y[lo:lo + sizes[rank]]
    layouts = []
    pid = {id(p): i for i, p in enumerate(params)}
    for it in range(3):
        for p in params:
            p.grad = None
        out = m(xs)
        if rank % 2 == 0 and rank != 3:
            out = out + 0.1 * extra(out)
        loss = (out - ys).abs().sum() * (world / total)      # sum over the rank's samples x world / N: the buckets' 1 / world makes it the mean over samples
        loss.backward()
        buckets.finish()
        layouts.append([[pid[id(p)] for p in b['params']] for b in buckets._buckets])
    torch.save(dict(grads=[None if p.grad is None else p.grad.clone() for p in params], layouts=layouts,
                    bytes=[sum(p.numel() * 4 for p in b['params']) for b in buckets._buckets], counts=[len(b['params']) for b in buckets._buckets],
                    tail=(buckets.effective_tail_bytes, buckets.tail_bucket_bytes, buckets.min_bucket_bytes)), os.path.join(out_dir, f'w8_{rank}.pt'))
    dist.destroy_process_group()


def test_eight_ranks_ragged_batches_unused_parameter_and_tail_buckets(tmp_path):
    """World size 8 (the driver's scaling run is 8 ranks on one node): per-rank batches 3/2/2/1/2/1/3/2, one parameter used by
    some ranks only, three iterations (the buckets are rebuilt in gradient-arrival order after the first): every rank ends with the
    gradient of the mean loss over all 16 samples, bit-identical across ranks; the bucket LAYOUT is identical on all ranks in every
    iteration (it decides the order of the collectives); and the last `tail_bytes` of the order sit in buckets of at most
    `tail_bucket_bytes` (what is reduced at the end of backward has nothing left to overlap with: keep it small)."""
    world, sizes = 8, (3, 2, 2, 1, 2, 1, 3, 2)
    mp.spawn(_worker_world8, args=(world, _free_port(), sizes, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f'w8_{r}.pt') for r in range(world)]
    for r in range(1, world):
        assert res[r]['layouts'] == res[0]['layouts'], f'rank {r}: bucket layout differs from rank 0'
        for a, b in zip(res[r]['grads'], res[0]['grads']):
            assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), f'rank {r}: gradients differ from rank 0'
    assert res[0]['layouts'][0] != res[0]['layouts'][2], 'the arrival-order rebuild did not happen'
    # reference: one process, all 16 samples
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(16, 96), torch.nn.LeakyReLU(0.2), torch.nn.Linear(96, 96), torch.nn.LeakyReLU(0.2),
                            torch.nn.Linear(96, 64), torch.nn.LeakyReLU(0.2), torch.nn.Linear(64, 4))
    extra = torch.nn.Linear(4, 4, bias=False)
    torch.manual_seed(99)
    x, y = torch.randn(sum(sizes), 16), torch.randn(sum(sizes), 4)
    out = m(x)
    use = torch.zeros(sum(sizes), 1)
    lo = 0
    for r, n in enumerate(sizes):
        if r % 2 == 0 and r != 3:
            use[lo:lo + n] = 1.0
        lo += n
    out = out + 0.1 * extra(out) * use
    ((out - y).abs().sum() / sum(sizes)).backward()
    want = [p.grad for p in list(extra.parameters()) + list(m.parameters())]
    for i, (g, w) in enumerate(zip(res[0]['grads'], want)):
        assert g is not None and torch.allclose(g, w, rtol=1e-5, atol=1e-7), (i, None if g is None else (g - w).abs().max().item())
    # tail: walking the bucket list from the end, everything inside the last tail_bytes is in buckets <= tail_bucket_bytes
    tail_bytes, tail_bucket, min_bucket = res[0]['tail']
    nbytes = res[0]['bytes']
    assert len(nbytes) >= 4
    assert min(nbytes) >= min_bucket, nbytes           # no collective for a handful of biases
    acc = 0
    for b, k in zip(reversed(nbytes), reversed(res[0]['counts'])):
        if acc + b > tail_bytes:
            break
        assert b <= 2 * tail_bucket + min_bucket or k == 1, (b, k)   # (a single parameter larger than the limit is a bucket of its own; a bucket closes where its size is nearest the limit; a group below the minimum joined its predecessor)
        acc += b
    assert acc > 0, 'no tail bucket at all'


def test_bench_network_bucket_layout_has_no_tiny_collectives():
    """The bench network's own parameter set (58.5 M parameters, 234 MB of fp32 gradients; shapes only: meta device) under the default
    bucket rules, in registration order reversed (first iteration) and in an arrival-like order (decoder back to front, encoder back to
    front, the mapping network last: what the rebuild produces): at most 14 collectives per step, none below 256 KB, the tail in
    pieces of at most an eighth of a bucket (+ a merged remainder), every parameter in exactly one bucket (the r05
    rehearsal issued 20 collectives, three of them of 0.0 MB)."""
    from afcm_amd import layer_schedule as sched
    from afcm_amd.distributed import GradientBuckets
    from afcm_amd.networks_stylegan3 import Stylegan3Generator
    G = Stylegan3Generator(z_dim=512, c_dim=1, w_dim=512, img_resolution=256, img_channels_in=4, img_channels_out=1,
                           mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(sched.DEFAULT_SYNTHESIS_KWARGS))
    named = [(n, torch.nn.Parameter(torch.empty(p.shape, device='meta'))) for n, p in G.named_parameters()]       # shapes only
    del G
    assert sum(p.numel() for _, p in named) > 58e6
    dec = [p for n, p in named if n.startswith('synthesis.L')]
    enc = [p for n, p in named if n.startswith('synthesis.') and not n.startswith('synthesis.L')]
    mp_ = [p for n, p in named if n.startswith('mapping.')]
    arrival = list(reversed(dec)) + list(reversed(enc)) + list(reversed(mp_))
    for order in (None, arrival):
        b = GradientBuckets([p for _, p in named])
        if order is not None:
            b._build(order)
        sizes = [sum(p.numel() * 4 for p in bk['params']) for bk in b._buckets]
        assert b.num_buckets <= 14, sizes
        assert min(sizes) >= 256 * 1024, sizes
        assert sorted(id(p) for bk in b._buckets for p in bk['params']) == sorted(id(p) for _, p in named)
        assert b.effective_tail_bytes <= sum(sizes) // 4


def test_small_parameter_set_is_not_all_tail():
    """gradients totalling <= 1.25 buckets used to be cut into ~10 eighth-of-a-bucket collectives; the tail is now a quarter
    of the bytes at most."""
    from afcm_amd.distributed import GradientBuckets
    params = [torch.nn.Parameter(torch.zeros(1 << 20, device='meta')) for _ in range(6)]       # 24 MB of fp32 gradients, 25 MB buckets
    b = GradientBuckets(params)
    sizes = [sum(p.numel() * 4 for p in bk['params']) for bk in b._buckets]
    assert b.num_buckets <= 3, sizes
