"""The `--model stylegan3` training step: generator-only step and the full D + G iteration.

Mirrors the reference's model wrapper for the path SURVEY.md section 8 scopes in: ``set_input``
(models/pix2pix_model.py:111-113 + models/comodgan_model.py:93-99), ``run_G`` / ``forward``
(models/stylegan3_model.py:13-22,85-87), the G update of ``optimize_parameters`` (:127-135: zero_grad,
forward, backward_G, NaN/Inf scrub of the gradients, Adam(beta=(0, 0.99)) step -- comodgan_model.py:19-20).
``StyleGAN3GeneratorStep`` is the generator-only step the headline metric times: its loss is the lambda_L1-weighted L1
term of models/stylegan3_model.py:107 (+ a caller-supplied extra term).  ``StyleGAN3Step`` adds the
discriminator update and the adversarial term of ``backward_G``: the full iteration of the reference.

Unlike the reference (which dereferences ``netG.module`` and therefore cannot run without DataParallel,
comodgan_model.py:14), the wrapper owns a bare module and shards by batch across ranks through
``afcm_amd.distributed.GradientBuckets``.
"""
import contextlib
import copy
import os

import numpy as np
import torch

from .distributed import GradientBuckets
from .torch_utils.ops import upfirdn2d
from .optim import FusedScrubAdam, weighted_l1


class StyleGAN3GeneratorStep:
    def __init__(self, netG, lr_G=0.0025, lambda_L1=100.0, distributed=False, bucket_bytes=25 * 1024 * 1024, style_mixing_prob=0,
                 force_collectives=False, blur_init_sigma=0.0, blur_fade_kimg=0.0, ema=False, comm_dtype=None, eval_dtype='auto', capturable=False):
        self.netG = netG
        # `ema`: keep the evaluation copy the reference creates unconditionally (models/comodgan_model.py:16-17); off by default
        # because the throughput path never reads it (234 MB)
        self.model_names = ['G']
        self.netG_ema = None
        if ema:
            self.netG_ema = copy.deepcopy(netG).eval()
            self.model_names.append('G_ema')
            # Accuracy budget of the evaluation path (north star: PSNR within 0.05 dB of the reference).  With uncorrelated
            # errors a 16-bit forward of PSNR P_e against the fp32 forward costs 10 log10(1 + 10^((P_t - P_e) / 10)) dB at a task
            # PSNR P_t: <= 0.05 dB needs P_e >= P_t + 19.4 dB, i.e. >= 52 dB for tasks up to 32.6 dB.  Full-width bf16 is
            # 44 dB (8 mantissa bits): fine for training throughput, over budget for reported metrics.  So `test()` /
            # `forward_ema()` run the EMA copy in fp16 when training runs in bf16 (11 mantissa bits, same kernels, same speed;
            # measured PSNR in tests/test_gpu_generator.py::test_full_width_16bit_accuracy_budget), else in the training dtype;
            # eval_dtype=torch.float32 selects the exact kernels.  The weights are fp32 masters either way.
            if eval_dtype == 'auto':
                eval_dtype = torch.float16 if netG.synthesis.compute_dtype == torch.bfloat16 else netG.synthesis.compute_dtype
            self.netG_ema.synthesis.compute_dtype = eval_dtype
        # loss-side blur schedule (models/stylegan3_model.py:80-81,115-116): sigma fades linearly to 0 over blur_fade_kimg
        self.blur_init_sigma, self.blur_fade_kimg, self.blur_sigma = float(blur_init_sigma), float(blur_fade_kimg), 0.0
        self.G_mapping = netG.mapping
        self.G_synthesis = netG.synthesis
        # nan_to_num of every gradient + Adam(betas=(0, 0.99)) (stylegan3_model.py:132-135, comodgan_model.py:19-20) as one HIP launch
        # (capturable: step count and bias corrections on the device, so that optimize_parameters() can be captured into a hipGraph: capture_step())
        self.optimizer_G = FusedScrubAdam(netG.parameters(), lr=lr_G, betas=(0.0, 0.99), eps=1e-8, scrub=True, posinf=1e5, neginf=-1e5, capturable=capturable)
        self.criterionL1 = torch.nn.L1Loss()
        self.lambda_L1 = lambda_L1
        self.style_mixing_prob = style_mixing_prob
        self.real_A = self.real_B = self.fake_B = None
        self.gen_z = self.gen_c = None
        self.buckets = GradientBuckets(netG.parameters(), bucket_bytes=bucket_bytes, force=force_collectives, comm_dtype=comm_dtype) if distributed else None
        if self.buckets is not None:
            self.buckets.broadcast_parameters(netG)

    def set_input(self, real_A, real_B, gen_z=None, gen_c=None):
        dev = next(self.netG.parameters()).device
        self.real_A = real_A.to(dev)
        self.real_B = real_B.to(dev)
        self.gen_z = torch.randn([real_A.shape[0], self.netG.z_dim], device=dev) if gen_z is None else gen_z.to(dev)
        self.gen_c = gen_c.to(dev) if gen_c is not None else torch.zeros([real_A.shape[0], self.netG.c_dim], device=dev)

    def run_G(self, cond_img, update_emas=False, noise_mode='random'):
        ref_img = self.real_B
        ws = self.G_mapping(z=self.gen_z, c=self.gen_c, img_in=ref_img, update_emas=False)
        if self.style_mixing_prob > 0:
            cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
            cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
            ws[:, cutoff:] = self.G_mapping(z=torch.randn_like(self.gen_z), c=self.gen_c, img_in=ref_img)[:, cutoff:]
        return self.G_synthesis(ws, cond_img, update_emas=False, noise_mode=noise_mode)

    def forward(self, update_emas=False):
        self.fake_B = self.run_G(self.real_A, update_emas=update_emas)

    @torch.no_grad()
    def forward_ema(self):
        """models/comodgan_model.py:114-116: the EMA generator, whole forward (mapping + synthesis), noise_mode='const'."""
        if self.netG_ema is None:
            raise RuntimeError('this step was built without the EMA generator (ema=True)')
        self.fake_B = self.netG_ema(z=self.gen_z, c=self.gen_c, cond_img=self.real_A, ref_img=self.real_B, noise_mode='const')

    def test(self):
        """models/comodgan_model.py:118-126: evaluation forward under no_grad."""
        self.forward_ema()

    def update_ema(self, batch_size, total_iters, ema_kimgs=10.0, ramp=None):
        """The G_ema update of the reference's train loop (train.py:67-77)."""
        return update_ema(self.netG_ema, self.netG, batch_size, total_iters, ema_kimgs, ramp)

    def save_networks(self, epoch, save_dir):
        """models/base_model.py:144-159: one '<epoch>_net_<name>.pth' per network holding the bare module's CPU state-dict
        (the reference unwraps DataParallel before saving, so the keys carry no 'module.' prefix)."""
        os.makedirs(save_dir, exist_ok=True)
        for name in self.model_names:
            net = getattr(self, 'net' + name)
            torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, os.path.join(save_dir, '%s_net_%s.pth' % (epoch, name)))

    def load_networks(self, epoch, save_dir, strict=True):
        """models/base_model.py:176-199: load '<epoch>_net_<name>.pth' for every network of this step; also accepts state-dicts
        saved from a DataParallel / DDP wrapper ('module.' prefix)."""
        for name in self.model_names:
            net = getattr(self, 'net' + name)
            path = os.path.join(save_dir, '%s_net_%s.pth' % (epoch, name))
            state = torch.load(path, map_location=next(net.parameters()).device, weights_only=True)
            if hasattr(state, '_metadata'):
                del state._metadata
            if all(k.startswith('module.') for k in state):
                state = {k[len('module.'):]: v for k, v in state.items()}
            net.load_state_dict(state, strict=strict)

    def _blur(self, img):
        """Gaussian blur of the loss inputs while blur_sigma > 0 (models/stylegan3_model.py:97-103): 2*floor(3 sigma)+1 taps,
        separable, through the HIP upfirdn2d kernel (filter2d)."""
        blur_size = np.floor(self.blur_sigma * 3)
        if blur_size <= 0:
            return img
        f = torch.arange(-blur_size, blur_size + 1, device=img.device).div(self.blur_sigma).square().neg().exp2()
        return upfirdn2d.filter2d(img, f / f.sum())

    def backward_G(self, extra_loss=None):
        # criterionL1(fake_B, real_B) * lambda_L1 (models/stylegan3_model.py:107), three launches instead of eight (optim.weighted_l1)
        self.loss_G_L1 = weighted_l1(self._blur(self.fake_B), self._blur(self.real_B), self.lambda_L1)
        self.loss_G = self.loss_G_L1 if extra_loss is None else self.loss_G_L1 + extra_loss
        self.loss_G.backward()

    def optimize_parameters(self, cur_nimg=None):
        """G half of models/stylegan3_model.py:113-135.  `cur_nimg` drives the blur fade as in the reference (:115-116);
        None keeps the current blur_sigma."""
        if cur_nimg is not None:
            self.blur_sigma = (max(1 - cur_nimg / (self.blur_fade_kimg * 1e3), 0) * self.blur_init_sigma) if self.blur_fade_kimg > 0 else 0.0
        ev = getattr(self, 'phase_events', None)      # a dict: record CUDA events at the phase boundaries of THIS step (bench.py's bucket timeline)
        mark = (lambda k: ev.__setitem__(k, _recorded_event())) if ev is not None else (lambda k: None)
        self.optimizer_G.zero_grad(set_to_none=True)
        mark('forward')
        self.forward(update_emas=False)
        mark('backward')
        self.backward_G()
        mark('finish')
        grads, scale = (None, 1.0) if self.buckets is None else self.buckets.finish_flat()
        # averaging (1 / world), the NaN/Inf scrub and the Adam update all happen inside the optimizer kernel; with buckets the
        # reduced gradients are read where the all-reduce left them
        self.optimizer_G.step(grads=grads, grad_scale=scale)
        mark('end')


def capture_step(step, inputs, warmup=3):
    """The whole training step -- set_input, forward, loss, backward, scrub + Adam -- as ONE hipGraph (r06; the step must have been built with
    capturable=True and without gradient buckets).  Every launch of the step goes to the current stream (the HIP kernels through the C ABI's
    stream argument, the framework's own), nothing in it reads the device from the host, and its allocations come from the graph's private
    pool, so the captured graph replays the step on the SAME input tensors (refresh them in place between replays).  Returns the
    torch.cuda.CUDAGraph; `graph.replay()` runs one step.  What a replay does not do: Python-side bookkeeping (optimizer state['step'] --
    FusedScrubAdam.device_step() has the count --, loss tensors are those of the captured step's buffers, refreshed by every replay)."""
    if step.buckets is not None or not step.optimizer_G.capturable or not getattr(getattr(step, 'optimizer_D', None), 'capturable', True):
        raise RuntimeError('capture_step needs a single-process step built with capturable=True')
    real_A, real_B, z, c = inputs

    def one():
        step.set_input(real_A, real_B, z, c)
        step.optimize_parameters()

    # warm-up AND capture on one side stream (allocator pools, workspace caches, pointer tables: everything lazy happens in the warm-up).  The same
    # stream for both, and the step's FIRST use of autograd should be this warm-up: a parameter's AccumulateGrad node runs on the stream that was
    # current when the node was made (the first time the parameter entered a graph) -- a step that has already run on the default stream would
    # accumulate its gradients there, outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(1, warmup)):
            one()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
        one()
    return graph


def _recorded_event():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


@torch.no_grad()
def update_ema(net_ema, net, batch_size, total_iters, ema_kimgs=10.0, ramp=None):
    """Exponential moving average of the generator as in the reference's train loop (train.py:67-77):
    beta = 0.5 ** (batch_size / ema_nimg), ema_nimg = min(ema_kimgs * 1000, total_iters * ramp); parameters are lerped,
    buffers copied.  One multi-tensor lerp instead of a Python loop of copies."""
    ema_nimg = ema_kimgs * 1000
    if ramp is not None:
        ema_nimg = min(ema_nimg, total_iters * ramp)
    beta = 0.5 ** (batch_size / max(ema_nimg, 1e-8))
    p_ema, p = list(net_ema.parameters()), list(net.parameters())
    # p_ema <- p.lerp(p_ema, beta) = p_ema + (1 - beta) * (p - p_ema)
    torch._foreach_lerp_(p_ema, p, 1.0 - beta)
    for b_ema, b in zip(net_ema.buffers(), net.buffers()):
        b_ema.copy_(b)
    return beta


class StyleGAN3Step(StyleGAN3GeneratorStep):
    """The full `--model stylegan3` iteration: discriminator update, then generator update, in the order and
    with the loss arithmetic of the reference (models/stylegan3_model.py:113-135; backward_D: models/comodgan_model.py:128-149;
    backward_G: models/stylegan3_model.py:89-111).  D sees cat(real_A, fake_B) (combine_ab) blurred by the same fading Gaussian.

    The D update's generator forward runs without a graph (the reference builds one and detaches the result,
    comodgan_model.py:131-133: same numbers, less memory).  Both updates use the fused scrub + Adam launch."""

    def __init__(self, netG, netD, lr_G=0.0002, lr_D=0.0002, lambda_L1=100.0, lambda_r1=10.0, combine_ab=True, **kw):
        super().__init__(netG, lr_G=lr_G, lambda_L1=lambda_L1, **kw)
        self.netD = netD
        self.model_names.insert(1, 'D')                             # ['G', 'D'(, 'G_ema')] as pix2pix_model.py:80 + comodgan_model.py:17
        self.lambda_r1 = float(lambda_r1)
        self.combine_ab = bool(combine_ab)
        self.optimizer_D = FusedScrubAdam(netD.parameters(), lr=lr_D, betas=(0.0, 0.99), eps=1e-8, scrub=True, posinf=1e5, neginf=-1e5,
                                          capturable=kw.get('capturable', False))
        self.buckets_D = None
        if self.buckets is not None:
            self.buckets_D = GradientBuckets(netD.parameters(), bucket_bytes=kw.get('bucket_bytes', 25 * 1024 * 1024),
                                             force=kw.get('force_collectives', False), comm_dtype=kw.get('comm_dtype'))
            self.buckets_D.broadcast_parameters(netD)

    def run_D(self, img, **kwargs):
        return self.netD(self._blur(img), **kwargs)                 # models/stylegan3_model.py:24-30

    def _pair(self, b):
        return torch.cat((self.real_A.to(b.dtype), b), 1) if self.combine_ab else b

    def backward_D(self):
        gen_logits = self.run_D(self._pair(self.fake_B).detach(), c=self.gen_c)
        self.loss_D_fake = torch.nn.functional.softplus(gen_logits).mean()
        # two backward passes per D update, as in the reference (comodgan_model.py:136,149): the first only accumulates into
        # .grad; the bucket all-reduces go out during the second (the longer one: it carries the R1 double backward)
        with (self.buckets_D.no_sync() if self.buckets_D is not None else contextlib.nullcontext()):
            self.loss_D_fake.backward()
        real_img_tmp = self._pair(self.real_B).detach().requires_grad_(True)
        real_logits = self.run_D(real_img_tmp, c=self.gen_c)
        self.loss_D_real = torch.nn.functional.softplus(-real_logits).mean()
        self.loss_D = self.loss_D_real
        if self.lambda_r1 > 0:
            r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp], create_graph=True, only_inputs=True)
            self.loss_Dr1 = r1_grads.square().sum([1, 2, 3]).mean() * 0.5
            self.loss_D = self.loss_D + self.loss_Dr1 * self.lambda_r1
        self.loss_D.backward()

    def backward_G(self, extra_loss=None):
        gen_logits = self.run_D(self._pair(self.fake_B), c=self.gen_c)
        self.loss_G_GAN = torch.nn.functional.softplus(-gen_logits).mean()
        super().backward_G(extra_loss=self.loss_G_GAN if extra_loss is None else self.loss_G_GAN + extra_loss)

    def optimize_parameters(self, cur_nimg=None):
        if cur_nimg is not None:
            self.blur_sigma = (max(1 - cur_nimg / (self.blur_fade_kimg * 1e3), 0) * self.blur_init_sigma) if self.blur_fade_kimg > 0 else 0.0
        # update D
        self.optimizer_D.zero_grad(set_to_none=True)
        self.netD.requires_grad_(True)
        with torch.no_grad():
            self.forward(update_emas=False)
        self.backward_D()
        self.netD.requires_grad_(False)
        grads, scale = (None, 1.0) if self.buckets_D is None else self.buckets_D.finish_flat()
        self.optimizer_D.step(grads=grads, grad_scale=scale)
        # update G
        super().optimize_parameters(cur_nimg=None)
