"""Fused gradient scrub + Adam (C ABI afcm_adam_multi) vs the eager sequence it replaces: torch.nan_to_num on every
gradient, then torch.optim.Adam(betas=(0, 0.99)) -- models/stylegan3_model.py:132-135, models/comodgan_model.py:19-20."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    torch.manual_seed(seed)
    shapes = [(512, 512, 3, 3), (64,), (181, 128, 3, 3), (1, 64, 1, 1), (7,), (1536, 91), (16385,), (3, 5, 7)]
    return [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]


@pytest.mark.parametrize('betas', [(0.0, 0.99), (0.9, 0.999)])
def test_fused_scrub_adam_matches_eager(betas):
    from afcm_amd.optim import FusedScrubAdam
    ref = _params(0)
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=0.0025, betas=betas, eps=1e-8)
    o_got = FusedScrubAdam(got, lr=0.0025, betas=betas, eps=1e-8, scrub=True, write_grad=True)
    for it in range(4):
        torch.manual_seed(100 + it)
        for pr, pg in zip(ref, got):
            g = torch.randn_like(pr) * (10.0 ** (it - 2))
            if it == 1:                       # exercise the scrub: NaN, +inf, -inf
                flat = g.view(-1)
                flat[0] = float('nan')
                if flat.numel() > 2:
                    flat[1] = float('inf')
                    flat[2] = float('-inf')
            pr.grad = g.clone()
            pg.grad = g.clone()
        for p in ref:
            if p.grad is not None:
                torch.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)
        o_ref.step()
        o_got.step()
        for i, (pr, pg) in enumerate(zip(ref, got)):
            assert torch.isfinite(pg).all()
            err = (pr - pg).abs().max().item()
            assert err <= 2e-6 * max(1.0, pr.abs().max().item()), (it, i, err)
            assert torch.equal(torch.nan_to_num(pr.grad), pg.grad), 'scrubbed gradient written back'
            for k in ('exp_avg', 'exp_avg_sq'):
                a, b = o_ref.state[pr][k], o_got.state[pg][k]
                assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item()), (it, i, k)


def test_fused_adam_consumes_external_gradients_with_scale():
    """The DDP path: gradients live in bucket slices (summed over ranks) and are averaged inside the kernel."""
    from afcm_amd.optim import FusedScrubAdam
    ref = _params(1)
    got = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=0.01, betas=(0.0, 0.99), eps=1e-8)
    o_got = FusedScrubAdam(got, lr=0.01, betas=(0.0, 0.99), eps=1e-8)
    flat = torch.randn(sum(p.numel() for p in ref), device='cuda')
    views, off = {}, 0
    for pr, pg in zip(ref, got):
        views[pg] = flat[off:off + pg.numel()].view(pg.shape)
        pr.grad = views[pg].clone() / 8
        off += pg.numel()
    o_ref.step()
    o_got.step(grads=views, grad_scale=1.0 / 8)
    for pr, pg in zip(ref, got):
        assert (pr - pg).abs().max().item() <= 2e-6 * max(1.0, pr.abs().max().item())
        assert pg.grad is None


def test_state_dict_round_trips_with_torch_adam():
    from afcm_amd.optim import FusedScrubAdam
    ps = _params(2)
    o = FusedScrubAdam(ps, lr=0.0025, betas=(0.0, 0.99))
    for p in ps:
        p.grad = torch.randn_like(p)
    o.step()
    sd = o.state_dict()
    t = torch.optim.Adam(ps, lr=0.0025, betas=(0.0, 0.99))
    t.load_state_dict(sd)
    assert set(t.state[ps[0]].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert float(t.state[ps[0]]['step']) == 1.0


def test_parameter_without_gradient_is_left_alone():
    from afcm_amd.optim import FusedScrubAdam
    ps = _params(3)
    o = FusedScrubAdam(ps, lr=0.1, betas=(0.0, 0.99))
    before = [p.detach().clone() for p in ps]
    for p in ps[1:]:
        p.grad = torch.ones_like(p)
    o.step()
    assert torch.equal(ps[0], before[0]) and len(o.state[ps[0]]) in (0, 3)
    assert all(not torch.equal(p, b) for p, b in zip(ps[1:], before[1:]))
