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


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(16, 1, 256, 256), (2, 1, 37, 5), (1, 1, 1, 1)], ids=str)
def test_weighted_l1_matches_l1loss_times_weight(shape):
    """optim.weighted_l1 (C ABI afcm_l1_partials / afcm_l1_grad) against torch.nn.L1Loss()(a, b) * weight (models/stylegan3_model.py:107):
    value, gradient (sign(a - b) * weight / numel, 0 where a == b), a NaN stays a NaN, chained with another differentiable factor."""
    from afcm_amd.optim import weighted_l1
    g = torch.Generator().manual_seed(1)
    a = torch.randn(shape, generator=g).cuda()
    b = torch.randn(shape, generator=g).cuda()
    a.flatten()[0] = b.flatten()[0]                            # an exact tie: gradient 0
    ar = a.clone().requires_grad_(True)
    ref = torch.nn.L1Loss()(ar, b) * 100.0
    gr, = torch.autograd.grad(ref * 0.5, [ar])
    af = a.clone().requires_grad_(True)
    got = weighted_l1(af, b, 100.0)
    gg, = torch.autograd.grad(got * 0.5, [af])
    assert got.shape == ref.shape and abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert torch.allclose(gg, gr, rtol=1e-6, atol=0) and float(gg.flatten()[0]) == 0.0
    if a.numel() > 4:
        an = a.clone()
        an.flatten()[3] = float('nan')
        an.requires_grad_(True)
        ln = weighted_l1(an, b, 1.0)
        gn, = torch.autograd.grad(ln, [an])
        assert bool(ln.isnan()) and bool(gn.flatten()[3].isnan()) and bool(torch.isfinite(gn.flatten()[4:]).all())


@pytest.mark.gpu
def test_captured_step_replays_bit_identically_to_eager_steps():
    """stylegan3_model.capture_step: the whole generator training step (forward, L1 loss, backward, scrub + Adam with the step count and bias
    corrections on the device) captured into ONE hipGraph; four replays leave every parameter bit-identical to four eager steps of the same
    (capturable) build on the same batch (eval mode: no dropout draw), and the device step count advances with the replays."""
    from afcm_amd import synthetic
    from afcm_amd.networks_stylegan3 import Stylegan3Generator
    from afcm_amd.stylegan3_model import StyleGAN3GeneratorStep, capture_step
    tiny = dict(channel_base=256, channel_max=8, num_layers=14, num_critical=2, margin_size=10, output_scale=0.25, skip_resolution=128,
                conv_kernel=3, filter_size=6, lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256, magnitude_ema_beta=0.5 ** (16 / 20e3),
                cond_mod=True, compute_dtype=torch.bfloat16)
    real_A, real_B, z, c = synthetic.generator_inputs(2, size=128, seed=0, device='cuda')
    inputs = (real_A, real_B, z[:, :32].contiguous(), c)
    finals = []
    for mode in ('eager', 'graph'):
        torch.manual_seed(0)
        G = Stylegan3Generator(z_dim=32, c_dim=1, w_dim=32, img_resolution=128, img_channels_in=4, img_channels_out=1, mapping_kwargs=dict(num_layers=2),
                               synthesis_kwargs=dict(tiny)).cuda().eval()
        step = StyleGAN3GeneratorStep(G, lr_G=0.0025, lambda_L1=100.0, capturable=True)
        if mode == 'eager':
            # (a capture records the step without running it: the graph run executes 3 warm-up steps + 4 replays = 7 steps)
            for _ in range(3 + 4):
                step.set_input(*inputs)
                step.optimize_parameters()
            finals.append([p.detach().clone() for p in G.parameters()])
            assert step.optimizer_G.device_step() == 7
        else:
            graph = capture_step(step, inputs, warmup=3)
            for _ in range(4):
                graph.replay()
            torch.cuda.synchronize()
            assert step.optimizer_G.device_step() == 7
            finals.append([p.detach().clone() for p in G.parameters()])
    for a, b in zip(*finals):
        assert torch.equal(a, b)
    assert all(bool(torch.isfinite(p).all()) for p in finals[1])
