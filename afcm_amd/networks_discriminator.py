"""Drop-in discriminator of the `--model stylegan3` path: ``CoModDiscriminator`` and its blocks with the
reference's class names, constructor arguments, ``forward()`` signatures and state-dict keys
(models/networks/CoModGAN/generator.py:613-836; layers of models/networks/CoModGAN/layers.py:81-162).

Resampling FIRs and bias/activation run on the HIP kernels (``upfirdn2d``, ``bias_act``: both arbitrarily differentiable, which
the R1 penalty's double backward needs); the contraction is the framework convolution, as in the reference (see
torch_utils/ops/conv2d_resample.py).  ``num_fp16_res`` runs the N highest-resolution blocks in 16 bit exactly as the reference
does (generator.py:808,819: ``use_fp16 = res >= fp16_resolution``; 0 in the shipped configurations,
models/stylegan3_model.py:69); the 16-bit type is float16 as in the reference unless ``fp16_dtype=torch.bfloat16`` is passed
(new: bf16 needs no loss scaling).  The channels-last switch is accepted and ignored (contiguous NCHW kernels).
"""
import numpy as np
import torch

from .torch_utils.ops import bias_act, conv2d_resample, upfirdn2d


class FullyConnectedLayer(torch.nn.Module):
    """layers.py:81-111."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)


class Conv2dLayer(torch.nn.Module):
    """layers.py:115-162 (with resampling)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation = activation
        self.up, self.down = up, down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
        b = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(b) if b is not None else None
        else:
            self.register_buffer('weight', weight)
            if b is not None:
                self.register_buffer('bias', b)
            else:
                self.bias = None

    def forward(self, x, gain=1):
        w = self.weight * self.weight_gain
        b = self.bias.to(x.dtype) if self.bias is not None else None
        flip_weight = (self.up == 1)
        x = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=flip_weight)
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=act_gain, clamp=act_clamp)


class DiscriminatorBlock(torch.nn.Module):
    """generator.py:613-692."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, architecture='resnet',
                 activation='lrelu', resample_filter=[1, 3, 3, 1], conv_clamp=None, use_fp16=False, fp16_channels_last=False,
                 freeze_layers=0, fp16_dtype=torch.float16):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.use_fp16 = use_fp16
        self.fp16_dtype = fp16_dtype
        self.in_channels = in_channels
        self.resolution = resolution
        self.img_channels = img_channels
        self.first_layer_idx = first_layer_idx
        self.architecture = architecture
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_layers = 0

        def trainable_gen():
            while True:
                layer_idx = self.first_layer_idx + self.num_layers
                self.num_layers += 1
                yield layer_idx >= freeze_layers
        trainable_iter = trainable_gen()
        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation, trainable=next(trainable_iter),
                                       conv_clamp=conv_clamp)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation, trainable=next(trainable_iter),
                                 conv_clamp=conv_clamp)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2, trainable=next(trainable_iter),
                                 resample_filter=resample_filter, conv_clamp=conv_clamp)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, trainable=next(trainable_iter),
                                    resample_filter=resample_filter)

    def _checked(self, t, channels):
        if list(t.shape[1:]) != [channels, self.resolution, self.resolution]:
            raise AssertionError(f'b{self.resolution}: expected [N, {channels}, {self.resolution}, {self.resolution}], got {list(t.shape)}')
        return t

    def forward(self, x, img, force_fp32=False):
        """generator.py:661-692: (features or None, image) -> (features at half the resolution, the image the next block takes or None).
        Restated, not mirrored: the image branch first (what the block takes in), then the trunk by architecture."""
        dtype = torch.float32 if (force_fp32 or not self.use_fp16) else self.fp16_dtype
        feat = None if x is None else self._checked(x, self.in_channels).to(dtype)
        if self.in_channels == 0 or self.architecture == 'skip':
            rgb = self._checked(img, self.img_channels).to(dtype)
            from_img = self.fromrgb(rgb)
            feat = from_img if feat is None else feat + from_img
            # only the 'skip' architecture hands a (half-resolution) image on to the next block
            img = upfirdn2d.downsample2d(rgb, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture != 'resnet':
            out = self.conv1(self.conv0(feat))
        else:
            half = np.sqrt(0.5)                    # shortcut and trunk each carry sqrt(1/2): unit variance after the sum
            shortcut = self.skip(feat, gain=half)
            out = shortcut.add_(self.conv1(self.conv0(feat), gain=half))
        assert out.dtype == dtype
        return out, img


class MinibatchStdLayer(torch.nn.Module):
    """generator.py:696-718."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = min(self.group_size, N) if self.group_size is not None else N
        F = self.num_channels
        c = C // F
        y = x.reshape(G, -1, F, c, H, W)
        y = y - y.mean(dim=0)
        y = y.square().mean(dim=0)
        y = (y + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4])
        y = y.reshape(-1, F, 1, 1)
        y = y.repeat(G, 1, H, W)
        return torch.cat([x, y], dim=1)


class DiscriminatorEpilogue(torch.nn.Module):
    """generator.py:722-776."""

    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4, mbstd_num_channels=1,
                 activation='lrelu', conv_clamp=None):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.cmap_dim = cmap_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.architecture = architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        """generator.py:755-776, always in float32: (+ the image through fromrgb, 'skip' only) -> minibatch std -> 3x3 conv -> two FC
        layers; with a label, the ``cmap_dim`` outputs are projected onto the mapped label (scaled by cmap_dim^-1/2)."""
        want = [self.in_channels, self.resolution, self.resolution]
        assert list(x.shape[1:]) == want, f'b{self.resolution}: expected [N, {want}], got {list(x.shape)}'
        feat = x.to(torch.float32)
        if self.architecture == 'skip':
            assert list(img.shape[1:]) == [self.img_channels, self.resolution, self.resolution]
            feat = feat + self.fromrgb(img.to(torch.float32))
        if self.mbstd is not None:
            feat = self.mbstd(feat)
        logits = self.out(self.fc(self.conv(feat).flatten(1)))
        if self.cmap_dim > 0:
            assert list(cmap.shape[1:]) == [self.cmap_dim]
            logits = (logits * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        return logits


class MappingNetwork(torch.nn.Module):
    """The StyleGAN2-style mapping network the discriminator conditions through (layers.py:540-609): optional latent z and label c
    -> normalised, concatenated -> ``num_layers`` FC + lrelu layers at lr_multiplier 0.01.  Same constructor arguments and
    state-dict keys (``embed``, ``fc0`` ... , ``w_avg``) as the reference class."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation='lrelu',
                 lr_multiplier=0.01, w_avg_beta=0.995, **kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        features = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(features[idx], features[idx + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    @staticmethod
    def _normalize_2nd_moment(x, eps=1e-8):
        return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()                    # layers.py:15-16

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False, **kwargs):
        x = None
        if self.z_dim > 0:
            assert list(z.shape[1:]) == [self.z_dim]
            x = self._normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            assert list(c.shape[1:]) == [self.c_dim]
            y = self._normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update and hasattr(self, 'w_avg'):
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class CoModDiscriminator(torch.nn.Module):
    """generator.py:780-836.  ``c_dim > 0`` -- the ADNI / in-house configurations (configs/adni/stylegan3/cmsr.yml:13: the slice
    fraction conditions D as it conditions G) -- maps the label through ``MappingNetwork(z_dim=0)`` and projects the epilogue's
    ``cmap_dim`` outputs onto it (generator.py:822-823,833-835,771-773)."""

    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=0,
                 conv_clamp=None, cmap_dim=None, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs={}, **kwargs):
        super().__init__()
        self.c_dim = c_dim
        self.img_resolution = img_resolution
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.img_channels = img_channels
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels_dict = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        if cmap_dim is None:
            cmap_dim = channels_dict[4]
        if c_dim == 0:
            cmap_dim = 0
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)                  # generator.py:808
        common_kwargs = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        cur_layer_idx = 0
        for res in self.block_resolutions:
            in_channels = channels_dict[res] if res < img_resolution else 0
            block = DiscriminatorBlock(in_channels, channels_dict[res], channels_dict[res // 2], resolution=res,
                                       first_layer_idx=cur_layer_idx, use_fp16=(res >= fp16_resolution), **block_kwargs, **common_kwargs)
            setattr(self, f'b{res}', block)
            cur_layer_idx += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(channels_dict[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **common_kwargs)

    def forward(self, img, c, **block_kwargs):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
        cmap = self.mapping(None, c) if self.c_dim > 0 else None
        return self.b4(x, img, cmap)
