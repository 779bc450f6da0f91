"""Row f1 on the GPU: the drop-in CoModDiscriminator (HIP upfirdn2d / bias_act, MFMA or framework conv) vs vectors captured from the
reference -- logits, the gradients of both discriminator loss terms including the R1 double backward
(models/comodgan_model.py:128-149), and the gradient the generator receives through D (models/stylegan3_model.py:93-95)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _build(g):
    from afcm_amd.networks_discriminator import CoModDiscriminator
    res, n, cb, cm, group, clamp = [int(v) for v in g['meta']]
    D = CoModDiscriminator(c_dim=(g['c'].shape[1] if 'c' in g else 0), img_resolution=res, img_channels=5, channel_base=cb, channel_max=cm,
                           conv_clamp=None if clamp < 0 else clamp, epilogue_kwargs=dict(mbstd_group_size=group))
    sd = {k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith('sd/')}
    D.load_state_dict(sd, strict=True)
    return D.cuda()


@pytest.mark.parametrize('mfma_conv', [True, False])
@pytest.mark.parametrize('name', ['D1_tiny64', 'D2_tiny128_clamp', 'D3_tiny64_cond'])
def test_discriminator_matches_reference_golden(name, mfma_conv, monkeypatch):
    """mfma_conv: 1x1 / 3x3 convolutions on the MFMA kernels of csrc/conv2d.hip (second-order through _ScaledConv2d / _ConvWgrad,
    strides by decimation) vs the framework convolution the reference uses."""
    from afcm_amd.torch_utils.ops import conv2d_resample
    monkeypatch.setattr(conv2d_resample, 'USE_MFMA_CONV', mfma_conv)
    monkeypatch.setattr(conv2d_resample, 'MFMA_CONV_FP32', mfma_conv)     # the goldens are fp32 networks
    g = load_golden(name)
    D = _build(g)
    names = [str(k) for k in g['names']]
    params = dict(D.named_parameters())
    fake, real = torch.from_numpy(g['fake']).cuda(), torch.from_numpy(g['real']).cuda()
    c = torch.from_numpy(g['c']).cuda() if 'c' in g else None          # D3: the conditional form (cmsr.yml:13), state dict incl. mapping.*
    # fake half
    gen_logits = D(fake, c)
    assert np.abs(gen_logits.detach().cpu().numpy() - g['gen_logits']).max() <= 1e-4
    loss_fake = torch.nn.functional.softplus(gen_logits).mean()
    gf = torch.autograd.grad(loss_fake, [params[k] for k in names])
    # real half + R1 (double backward through upfirdn2d / bias_act / conv)
    real_tmp = real.detach().requires_grad_(True)
    real_logits = D(real_tmp, c)
    loss_real = torch.nn.functional.softplus(-real_logits).mean()
    r1, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp], create_graph=True, only_inputs=True)
    loss_r1 = r1.square().sum([1, 2, 3]).mean() * 0.5
    assert np.abs(r1.detach().cpu().numpy() - g['r1_grads']).max() <= 1e-6 + 1e-3 * np.abs(g['r1_grads']).max()
    assert abs(loss_r1.item() - float(g['loss_r1'])) <= 1e-3 * float(g['loss_r1']) + 1e-8
    gr = torch.autograd.grad(loss_real + loss_r1 * 10.0, [params[k] for k in names], retain_graph=True)
    # the R1 term is 0-6 % of greal, so the 2e-4 bound on the sum sees R1 errors above ~1 % only -- the double
    # backward is asserted on its own, at 1e-3 of its own scale per tensor (plus a floor for tensors whose R1 gradient is ~0)
    g1 = torch.autograd.grad(loss_r1, [params[k] for k in names], allow_unused=True)
    r1_scale = max(float(np.abs(g['gr1/' + k]).max()) for k in names)
    for k, a in zip(names, g1):
        want = g['gr1/' + k]
        got = np.zeros_like(want) if a is None else a.cpu().numpy()
        err = float(np.abs(got - want).max())
        assert err <= 1e-3 * float(np.abs(want).max()) + 2e-5 * r1_scale, (name, k, 'r1 alone', err, float(np.abs(want).max()))
    for k, a, b in zip(names, gf, gr):
        for got, want, what in ((a, g['gfake/' + k], 'fake'), (b, g['greal/' + k], 'real+r1')):
            tol = 2e-4 * max(1e-3, float(np.abs(want).max()))
            err = float(np.abs(got.cpu().numpy() - want).max())
            assert err <= tol, (name, k, what, err, tol)
    # generator term through D
    img = fake.clone().requires_grad_(True)
    lg = torch.nn.functional.softplus(-D(img, c)).mean()
    gi, = torch.autograd.grad(lg, img)
    d = gi.cpu().numpy().astype(np.float64) - g['g_img']
    rel = float(np.sqrt((d ** 2).sum() / (g['g_img'].astype(np.float64) ** 2).sum()))
    assert rel <= 1e-3, f'{name}: image gradient of the G term, relative L2 {rel:.3e}'
    assert np.abs(d).max() <= 1e-2 * float(np.abs(g['g_img']).max())      # isolated leaky-ReLU kink flips, cf. test_gpu_generator


def test_discriminator_full_width_state_dict_and_step():
    """The shipped configuration (models/stylegan3_model.py:66-78): 24.0 M parameters, one D loss evaluation with R1 at batch 4."""
    from afcm_amd.networks_discriminator import CoModDiscriminator
    D = CoModDiscriminator(c_dim=0, img_resolution=256, img_channels=5, channel_base=int(0.5 * 32768), channel_max=512,
                           epilogue_kwargs=dict(mbstd_group_size=16)).cuda()
    n = sum(p.numel() for p in D.parameters())
    assert n == 24001217, n        # the reference class with these kwargs (c_dim = 0: no mapping network)
    x = torch.randn(4, 5, 256, 256, device='cuda', requires_grad=True)
    logits = D(x, None)
    assert logits.shape == (4, 1)
    r1, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    (torch.nn.functional.softplus(-logits).mean() + 5.0 * r1.square().sum([1, 2, 3]).mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in D.parameters())


@pytest.mark.parametrize('mfma_conv', [True, False])
def test_resampling_conv2dlayer_matches_reference_golden(mfma_conv, monkeypatch):
    """The generator-side Conv2dLayer (afcm_amd/networks_stylegan3.py; CoModGAN/layers.py:115-162) with up / down != 1 -- every branch
    of conv2d_resample -- vs vectors captured from the reference layer: output and the three gradients."""
    from afcm_amd.networks_stylegan3 import Conv2dLayer
    from afcm_amd.torch_utils.ops import conv2d_resample
    monkeypatch.setattr(conv2d_resample, 'USE_MFMA_CONV', mfma_conv)
    monkeypatch.setattr(conv2d_resample, 'MFMA_CONV_FP32', mfma_conv)
    g = load_golden('C1_conv2dlayer_resample')
    for n, (k, up, down) in enumerate(g['cases']):
        k, up, down = int(k), int(up), int(down)
        layer = Conv2dLayer(3, 5, kernel_size=k, up=up, down=down, activation='lrelu', conv_clamp=(2.0 if n == 0 else None))
        layer.load_state_dict({'weight': torch.from_numpy(g[f'{n}/w']), 'bias': torch.from_numpy(g[f'{n}/b']),
                               'resample_filter': layer.resample_filter}, strict=True)
        layer = layer.cuda()
        x = torch.from_numpy(g[f'{n}/x']).cuda().requires_grad_(True)
        y = layer(x, gain=0.7)
        assert tuple(y.shape) == g[f'{n}/y'].shape, (k, up, down)
        assert np.abs(y.detach().cpu().numpy() - g[f'{n}/y']).max() <= 2e-5, (k, up, down)
        gx, gw, gb = torch.autograd.grad((y * torch.from_numpy(g[f'{n}/r']).cuda()).sum(), [x, layer.weight, layer.bias])
        for got, key in ((gx, 'gx'), (gw, 'gw'), (gb, 'gb')):
            want = g[f'{n}/{key}']
            assert np.abs(got.cpu().numpy() - want).max() <= 1e-4 * max(1.0, float(np.abs(want).max())), (k, up, down, key)


def test_conditional_discriminator_full_width():
    """The ADNI / in-house configurations' discriminator (configs/adni/stylegan3/cmsr.yml:11-16: c_dim = 1, mbstd_group_size 16): the
    label mapping network adds 8 FC layers of 512 + the embedding; one conditioned loss evaluation with R1 runs and every
    parameter -- mapping network included -- receives a finite gradient."""
    from afcm_amd.networks_discriminator import CoModDiscriminator
    D = CoModDiscriminator(c_dim=1, img_resolution=256, img_channels=5, channel_base=int(0.5 * 32768), channel_max=512,
                           epilogue_kwargs=dict(mbstd_group_size=16)).cuda()
    n = sum(p.numel() for p in D.parameters())
    # c_dim = 0 network (24,001,217) with b4.out widened 1 -> 512 (+ 511 x 513) + embed (1 x 512 + 512) + 8 x (512 x 512 + 512)
    assert n == 24001217 + 511 * 513 + 1024 + 8 * (512 * 512 + 512), n
    assert [k for k in D.state_dict() if k.startswith('mapping.')][:2] == ['mapping.embed.weight', 'mapping.embed.bias']
    x = torch.randn(4, 5, 256, 256, device='cuda', requires_grad=True)
    c = torch.rand(4, 1, device='cuda')
    logits = D(x, c)
    assert logits.shape == (4, 1)
    r1, = torch.autograd.grad(logits.sum(), x, create_graph=True)
    (torch.nn.functional.softplus(-logits).mean() + 5.0 * r1.square().sum([1, 2, 3]).mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in D.parameters())
    # the label matters (at initialisation -- zero embedding bias, second-moment normalisation -- only through its sign; the golden
    # D3 network has random biases and pins the general case)
    assert not torch.allclose(D(x.detach(), c), D(x.detach(), -c))


def test_full_training_iteration_matches_oracle():
    """The whole `--model stylegan3` iteration against the CPU oracle on the same tensors and weights (the reference-captured
    state dicts of the G1 generator and D2 discriminator goldens, fp32, loss-side blur active):
    D half (models/comodgan_model.py:128-149)  -- loss_D_fake, loss_D_real, R1 penalty, and the gradient of every D parameter
                                                   after BOTH backward passes (fake term; real term + lambda_r1 x R1);
    G half (models/stylegan3_model.py:89-111)   -- loss_G_GAN through D, lambda_L1 x L1 on the blurred images, and the gradient
                                                   of every G parameter.
    Oracle: oracle/generator.generator + oracle/discriminator.discriminator composed as those reference lines compose them."""
    from afcm_amd.stylegan3_model import StyleGAN3Step
    from afcm_amd.torch_utils.ops import conv2d_resample
    from oracle import aten_ops as ops
    from oracle import discriminator as odisc
    from oracle import generator as ogen
    from test_gpu_generator import TINY, _build as build_G
    gG, gD = load_golden('G1_tiny128'), load_golden('D2_tiny128_clamp')
    res, _, _, _, group, clamp = [int(v) for v in gD['meta']]
    sdG = {k[3:]: torch.from_numpy(np.array(v)) for k, v in gG.items() if k.startswith('sd/')}
    sdD = {k[3:]: torch.from_numpy(np.array(v)) for k, v in gD.items() if k.startswith('sd/')}
    G = build_G(128).eval()
    G.load_state_dict(sdG, strict=True)
    D = _build(gD)
    lam_l1, lam_r1, sigma = 100.0, 10.0, 2.0
    step = StyleGAN3Step(G.cuda(), D, lambda_L1=lam_l1, lambda_r1=lam_r1, blur_init_sigma=sigma, blur_fade_kimg=1.0)
    step.blur_sigma = sigma
    z, c, a = (torch.from_numpy(gG[k]) for k in ('z', 'c', 'x'))
    torch.manual_seed(11)
    b = (torch.from_numpy(gD['real'][:, 4:5]) * 0.5).clamp(-1, 1)                 # real_B: the golden's real target channel
    step.set_input(a, b, z, c)
    # ---- GPU: D half, then G half (no optimizer step in between: both halves are compared on the same weights)
    D.requires_grad_(True)
    with torch.no_grad():
        step.forward()
    step.backward_D()
    d_names = [n for n, _ in D.named_parameters()]
    d_grads = [p.grad.detach().cpu().clone() for p in D.parameters()]
    got_d = [step.loss_D_fake.item(), step.loss_D_real.item(), step.loss_Dr1.item()]
    D.requires_grad_(False)
    step.forward()
    step.backward_G()
    g_names = [n for n, p in G.named_parameters() if p.grad is not None]
    g_grads = [p.grad.detach().cpu().clone() for p in G.parameters() if p.grad is not None]
    got_g = [step.loss_G_GAN.item(), step.loss_G_L1.item()]
    # ---- oracle
    pl = ogen.plan(128, 4, 1, dict(TINY))
    f = torch.arange(-np.floor(sigma * 3), np.floor(sigma * 3) + 1).div(sigma).square().neg().exp2().float()
    f = f / f.sum()
    blur = lambda t: ops.filter2d(t, f)                                            # models/stylegan3_model.py:24-30,97-103
    oD = {k: v.clone().requires_grad_(True) for k, v in sdD.items()}
    oG = {k: (v.clone().requires_grad_(True) if k in dict(G.named_parameters()) else v.clone()) for k, v in sdG.items()}
    kw = dict(mbstd_group_size=group, conv_clamp=clamp)
    fake = ogen.generator(oG, pl, z, c, a, mapping_layers=2)
    gen_logits = odisc.discriminator(oD, blur(torch.cat([a, fake.detach()], 1)), res, **kw)
    loss_fake = torch.nn.functional.softplus(gen_logits).mean()
    real_tmp = torch.cat([a, b], 1).requires_grad_(True)
    real_logits = odisc.discriminator(oD, blur(real_tmp), res, **kw)
    loss_real = torch.nn.functional.softplus(-real_logits).mean()
    r1, = torch.autograd.grad([real_logits.sum()], [real_tmp], create_graph=True)
    loss_r1 = r1.square().sum([1, 2, 3]).mean() * 0.5
    want_d = torch.autograd.grad(loss_fake + loss_real + lam_r1 * loss_r1, [oD[k] for k in d_names])
    loss_gan = torch.nn.functional.softplus(-odisc.discriminator(oD, blur(torch.cat([a, fake], 1)), res, **kw)).mean()
    loss_l1 = (blur(fake) - blur(b)).abs().mean() * lam_l1
    want_g = torch.autograd.grad(loss_gan + loss_l1, [oG[k] for k in g_names])
    for what, got, want in zip(('loss_D_fake', 'loss_D_real', 'loss_Dr1', 'loss_G_GAN', 'loss_G_L1'), got_d + got_g,
                               [loss_fake, loss_real, loss_r1, loss_gan, loss_l1]):
        assert abs(got - want.item()) <= 2e-5 * max(1.0, abs(want.item())), (what, got, want.item())

    def check(names, got, want, rel_tol, what):
        for k, x_, y_ in zip(names, got, want):
            d = x_.double() - y_.double()
            rel = float(d.norm() / y_.double().norm().clamp_min(1e-30))
            assert rel <= rel_tol, f'{what} gradient {k}: relative L2 {rel:.3e}'
            assert float(d.abs().max()) <= 2e-2 * max(1e-6, float(y_.abs().max())), f'{what} gradient {k}: max-abs'
    # D: two accumulated backward passes incl. the R1 double backward; G: through D and through the blurred L1 term.  Relative
    # L2 because isolated leaky-ReLU kink flips are allowed (test_generator_gradients_with_the_kernels_branch_decisions_imposed)
    check(d_names, d_grads, want_d, 2e-3, 'D')
    check(g_names, g_grads, want_g, 1e-2, 'G')
    # and the step itself: scrub + Adam on those gradients moves every parameter, D first then G
    d0 = [p.detach().clone() for p in D.parameters()]
    g0 = [p.detach().clone() for p in G.parameters()]
    step.optimize_parameters(cur_nimg=800)
    assert abs(step.blur_sigma - 0.4) < 1e-9
    assert all((p - q).abs().max().item() > 0 for p, q in zip(D.parameters(), d0))
    assert sum(int((p - q).abs().max().item() > 0) for p, q in zip(G.parameters(), g0)) >= len(g0) - 2
    assert all(torch.isfinite(p).all() for p in list(G.parameters()) + list(D.parameters()))


def test_full_training_iteration_runs_in_bf16():
    """Smoke only (no oracle comparison): one full iteration with bf16 generator activations on tiny networks stays finite and
    moves every parameter; the fade schedule of the loss-side blur follows cur_nimg."""
    from afcm_amd.networks_discriminator import CoModDiscriminator
    from afcm_amd.networks_stylegan3 import Stylegan3Generator
    from afcm_amd.stylegan3_model import StyleGAN3Step
    from afcm_amd import synthetic
    from test_gpu_generator import TINY
    torch.manual_seed(0)
    G = Stylegan3Generator(z_dim=32, c_dim=1, w_dim=32, img_resolution=128, img_channels_in=4, img_channels_out=1,
                           mapping_kwargs=dict(num_layers=2), synthesis_kwargs=dict(TINY, compute_dtype=torch.bfloat16)).cuda().train()
    D = CoModDiscriminator(c_dim=0, img_resolution=128, img_channels=5, channel_base=1024, channel_max=16,
                           epilogue_kwargs=dict(mbstd_group_size=2)).cuda()
    step = StyleGAN3Step(G, D, blur_init_sigma=2.0, blur_fade_kimg=1.0)
    a, b, z, c = synthetic.generator_inputs(4, size=128, z_dim=32, seed=0, device='cuda')
    g0 = [p.detach().clone() for p in G.parameters()]
    d0 = [p.detach().clone() for p in D.parameters()]
    step.set_input(a, b, z, c)
    step.optimize_parameters(cur_nimg=0)
    assert step.blur_sigma == 2.0
    for t in (step.loss_D_fake, step.loss_D_real, step.loss_Dr1, step.loss_G_GAN, step.loss_G_L1):
        assert torch.isfinite(t).all()
    assert all((p - q).abs().max().item() > 0 for p, q in zip(D.parameters(), d0))
    assert sum(int((p - q).abs().max().item() > 0) for p, q in zip(G.parameters(), g0)) >= len(g0) - 2
    step.set_input(a, b, z, c)
    step.optimize_parameters(cur_nimg=800)
    assert abs(step.blur_sigma - 0.4) < 1e-9
    assert all(torch.isfinite(p).all() for p in list(G.parameters()) + list(D.parameters()))


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 2e-2), (torch.bfloat16, 8e-2)])
def test_discriminator_16bit_blocks_track_the_fp32_reference(dtype, tol):
    """num_fp16_res (generator.py:808,819): the highest-resolution blocks in 16 bit (HIP upfirdn2d / bias_act in that dtype,
    16-bit convs on the MFMA kernels) vs the fp32 golden logits and R1 gradient; looser bound = 16-bit rounding only."""
    from afcm_amd.networks_discriminator import CoModDiscriminator
    g = load_golden('D2_tiny128_clamp')
    res, n, cb, cm, group, clamp = [int(v) for v in g['meta']]
    D = CoModDiscriminator(c_dim=0, img_resolution=res, img_channels=5, channel_base=cb, channel_max=cm, conv_clamp=clamp, num_fp16_res=3,
                           block_kwargs=dict(fp16_dtype=dtype), epilogue_kwargs=dict(mbstd_group_size=group))
    D.load_state_dict({k[3:]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith('sd/')}, strict=True)
    D = D.cuda()
    assert [getattr(D, f'b{r}').use_fp16 for r in (128, 64, 32, 16, 8)] == [True, True, True, False, False]
    real = torch.from_numpy(g['real']).cuda().requires_grad_(True)
    logits = D(real, None)
    assert logits.dtype == torch.float32
    scale = max(1.0, float(np.abs(g['real_logits']).max()))
    assert np.abs(logits.detach().cpu().numpy() - g['real_logits']).max() <= tol * scale
    r1, = torch.autograd.grad(logits.sum(), real, create_graph=True)
    d = r1.detach().cpu().numpy() - g['r1_grads']
    assert np.sqrt((d ** 2).sum() / (g['r1_grads'] ** 2).sum()) <= 4 * tol
    (r1.square().sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in D.parameters() if p.grad is not None)
