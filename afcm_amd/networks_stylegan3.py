"""Conditional, U-shaped alias-free (StyleGAN3-style) generator of AFCM on the MI355X kernels.

Drop-in for the reference module ``models/networks/stylegan3/networks_stylegan3.py`` (NET): the class
names, constructor arguments, ``forward()`` signatures, parameter/buffer names (state-dict keys) and
the layer schedule are the reference's, so reference checkpoints load and ``--model stylegan3`` code can
swap the import.  What differs is underneath: every hot op goes to the HIP library --
``filtered_lrelu`` / ``bias_act`` (afcm_amd/csrc/*.hip) and the MFMA convolution with the
shared-weight form of the style modulation (see torch_utils/ops/conv2d.py).

New capability relative to the reference: ``compute_dtype`` (fp32 default, bf16/fp16 optional) selects the
storage type of the activation stream; the reference hard-codes fp32 (NET:619,653).  Weights, styles,
demodulation coefficients and every accumulation stay fp32.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import layer_schedule as sched
from .torch_utils.ops import affine_bank, bias_act, conv2d_gradfix, fc_bank, filtered_lrelu, fused_layer, modulation_bank
from .torch_utils.ops import conv2d as _conv_ops
from .torch_utils.ops.conv2d import modulation_coefficients_fused, scaled_conv2d
from .torch_utils.ops.conv2d import modulated_conv2d  # noqa: F401  (re-exported: NET:25 lives in this module)


def _assert_shape(t, ref):
    if t.ndim != len(ref):
        raise AssertionError(f'Wrong number of dimensions: got {t.ndim}, expected {len(ref)}')
    for idx, (size, want) in enumerate(zip(t.shape, ref)):
        if want is not None and size != want:
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {want}')


class _ScaledLinear(torch.autograd.Function):
    """y = alpha * x @ w.t() (+ beta_b * b): the equalised-lr gains ride as the GEMMs' alpha in the forward AND in both
    gradients (the framework's addmm backward multiplies each gradient by alpha in a separate launch; the reference's
    ``w * weight_gain`` costs a pass over w each way, NET:97-100).  backward is made of differentiable ops (no
    once_differentiable), so higher-order gradients still work."""

    @staticmethod
    def forward(ctx, x, w, b, alpha, bias_gain):
        ctx.save_for_backward(x, w)
        ctx.gains = (float(alpha), float(bias_gain), b is not None)
        wt = w.t()
        if b is None:
            return torch.addmm(x.new_empty([w.shape[0]]), x, wt, beta=0.0, alpha=float(alpha))
        return torch.addmm(b.unsqueeze(0), x, wt, beta=float(bias_gain), alpha=float(alpha))

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        alpha, bias_gain, has_b = ctx.gains
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.addmm(g.new_empty([w.shape[1]]), g, w, beta=0.0, alpha=alpha)
        if ctx.needs_input_grad[1]:
            dw = torch.addmm(g.new_empty([w.shape[1]]), g.t(), x, beta=0.0, alpha=alpha)
        if has_b and ctx.needs_input_grad[2]:
            db = g.sum(0)
            if bias_gain != 1:
                db = db * bias_gain
        return dx, dw, db, None, None


class _ScaleCast(torch.autograd.Function):
    """(x * scale).to(out_dtype) as ONE pass each way (C ABI afcm_scale_planes with a constant per-plane factor) instead of a multiply and a
    cast: the generator's last two ops (NET:703-705: `x * output_scale`, `.to(float32)`).  The product is formed in fp32 (the reference rounds
    it to the compute dtype first; with output_scale a power of two, as shipped, the results are identical)."""
    _planes = {}

    @staticmethod
    def _factor(n, scale, device):
        key = (n, float(scale), device)
        t = _ScaleCast._planes.get(key)
        if t is None:
            if len(_ScaleCast._planes) > 64:
                _ScaleCast._planes.clear()
            t = _ScaleCast._planes[key] = torch.full([n], float(scale), dtype=torch.float32, device=device)
        return t

    @staticmethod
    def forward(ctx, x, scale, out_dtype):
        ctx.cfg = (float(scale), x.dtype)
        return _conv_ops.scale_planes(x, _ScaleCast._factor(x.shape[0] * x.shape[1], scale, x.device).view(x.shape[0], x.shape[1]), out_dtype)

    @staticmethod
    def backward(ctx, g):
        scale, dt = ctx.cfg
        return _ScaleCast.apply(g, scale, dt), None, None


class _PoolBlocks(torch.autograd.Function):
    """[N, C, H, W] (16-bit or fp32 device tensor, H % 4 == W % 4 == 0) -> [N, C, 4, 4] fp32 block means (C ABI afcm_pool_blocks_fwd / _bwd): the
    evenly dividing case of AdaptiveAvgPool2d((4, 4)) (NET:636,683) in one launch each way, straight from the 16-bit activations."""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty([n, c, 4, 4], dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().afcm_pool_blocks_fwd(y.data_ptr(), x.data_ptr(), _lib.dtype_code(x), n * c, h, w, _lib.stream_ptr(x)), 'pool_blocks_fwd')
        ctx.cfg = (tuple(x.shape), x.dtype)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        from . import _lib
        shape, dtype = ctx.cfg
        n, c, h, w = shape
        dx = torch.empty(shape, dtype=dtype, device=gy.device)
        gy = gy.to(torch.float32).contiguous()
        _lib.check(_lib.load().afcm_pool_blocks_bwd(dx.data_ptr(), gy.data_ptr(), _lib._DTYPES[dtype], n * c, h, w, _lib.stream_ptr(gy)), 'pool_blocks_bwd')
        return dx


class FullyConnectedLayer(torch.nn.Module):
    """Equalised-learning-rate dense layer (NET:69-104)."""

    def __init__(self, in_features, out_features, activation='linear', bias=True, lr_multiplier=1, weight_init=1, bias_init=0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) * (weight_init / lr_multiplier))
        bias_init = np.broadcast_to(np.asarray(bias_init, dtype=np.float32), [out_features])
        self.bias = torch.nn.Parameter(torch.from_numpy(bias_init / lr_multiplier)) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        if fc_bank.supported(x, self.weight, self.activation):
            # GEMM + gains + bias + activation as one launch (and one backward): torch_utils/ops/fc_bank.py
            return fc_bank.fc_act(x, self.weight, self.bias, self.weight_gain, self.bias_gain, self.activation)
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
        w = self.weight.to(x.dtype)
        if x.ndim == 2:
            # the equalised-lr gains as GEMM alpha / beta: same values as x @ (w * gain).t() + b * bias_gain (NET:97-100)
            if self.activation == 'linear':
                return _ScaledLinear.apply(x, w, b, self.weight_gain, self.bias_gain)
            if b is not None and self.bias_gain != 1:
                b = b * self.bias_gain
            return bias_act.bias_act(_ScaledLinear.apply(x, w, None, self.weight_gain, 1.0), b, act=self.activation)
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        w = w * self.weight_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class MappingNetwork(torch.nn.Module):
    """z, c -> ws (NET:109-164)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=2, lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers = num_layers
        self.w_avg_beta = w_avg_beta
        self.embed = FullyConnectedLayer(c_dim, w_dim) if c_dim > 0 else None
        features = [z_dim + (w_dim if c_dim > 0 else 0)] + [w_dim] * num_layers
        for idx, (fin, fout) in enumerate(zip(features[:-1], features[1:])):
            setattr(self, f'fc{idx}', FullyConnectedLayer(fin, fout, activation='lrelu', lr_multiplier=lr_multiplier))
        self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **kwargs):
        _assert_shape(z, [None, self.z_dim])
        if truncation_cutoff is None:
            truncation_cutoff = self.num_ws
        x = z.to(torch.float32)
        if self.c_dim > 0:
            _assert_shape(c, [None, self.c_dim])
        cc = c.to(torch.float32) if self.c_dim > 0 else None
        if fc_bank.mapping_input_supported(x, cc, self.embed):
            # both normalisations, the embedding and the cat as one launch (NET:143-150)
            x = fc_bank.mapping_input(x, cc, self.embed)
        else:
            x = x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
            if self.c_dim > 0:
                y = self.embed(cc)
                y = y * (y.square().mean(1, keepdim=True) + 1e-8).rsqrt()
                x = torch.cat([x, y], dim=1)
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x)
        if update_emas:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x

    def extra_repr(self):
        return f'z_dim={self.z_dim:d}, c_dim={self.c_dim:d}, w_dim={self.w_dim:d}, num_ws={self.num_ws:d}'


class _ResampleGeometry:
    """Shared constructor arithmetic of SynthesisLayer / EncoderLayer (NET:294-334, 453-489)."""

    def _setup_resampling(self, in_size, out_size, in_sampling_rate, out_sampling_rate, in_cutoff, out_cutoff, in_half_width,
                          out_half_width, conv_kernel, filter_size, lrelu_upsampling, use_radial_filters, is_torgb,
                          is_critically_sampled):
        self.in_size = np.broadcast_to(np.asarray(in_size), [2])
        self.out_size = np.broadcast_to(np.asarray(out_size), [2])
        self.in_sampling_rate, self.out_sampling_rate = in_sampling_rate, out_sampling_rate
        self.in_cutoff, self.out_cutoff = in_cutoff, out_cutoff
        self.in_half_width, self.out_half_width = in_half_width, out_half_width
        (self.tmp_sampling_rate, self.up_factor, self.up_taps, self.down_factor, self.down_taps,
         self.padding) = sched.resample_geometry(in_size, out_size, in_sampling_rate, out_sampling_rate, conv_kernel, filter_size,
                                                 lrelu_upsampling, is_torgb)
        self.down_radial = use_radial_filters and not is_critically_sampled
        self.register_buffer('up_filter', sched.design_lowpass_filter(
            numtaps=self.up_taps, cutoff=in_cutoff, width=in_half_width * 2, fs=self.tmp_sampling_rate))
        self.register_buffer('down_filter', sched.design_lowpass_filter(
            numtaps=self.down_taps, cutoff=out_cutoff, width=out_half_width * 2, fs=self.tmp_sampling_rate, radial=self.down_radial))

    @staticmethod
    def design_lowpass_filter(numtaps, cutoff, width, fs, radial=False):
        return sched.design_lowpass_filter(numtaps, cutoff, width, fs, radial)


class SynthesisLayer(torch.nn.Module, _ResampleGeometry):
    """Style-modulated conv + filtered leaky ReLU (+ encoder skip) (NET:253-412)."""

    def __init__(self, w_dim, global_w_dim, is_torgb, is_critically_sampled, use_fp16, in_channels, out_channels, in_size, out_size,
                 in_sampling_rate, out_sampling_rate, in_cutoff, out_cutoff, in_half_width, out_half_width, conv_kernel=3,
                 filter_size=6, lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256, magnitude_ema_beta=0.999,
                 cond_mod=False):
        super().__init__()
        self.w_dim = w_dim
        self.is_torgb = is_torgb
        self.is_critically_sampled = is_critically_sampled
        self.use_fp16 = use_fp16
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_kernel = 1 if is_torgb else conv_kernel
        self.conv_clamp = conv_clamp
        self.magnitude_ema_beta = magnitude_ema_beta
        self.cond_mod = cond_mod
        if not cond_mod:
            global_w_dim = 0
        self.affine = FullyConnectedLayer(self.w_dim + global_w_dim, self.in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([self.out_channels, self.in_channels, self.conv_kernel, self.conv_kernel]))
        self.bias = torch.nn.Parameter(torch.zeros([self.out_channels]))
        self.register_buffer('magnitude_ema', torch.ones([]))
        self._setup_resampling(in_size, out_size, in_sampling_rate, out_sampling_rate, in_cutoff, out_cutoff, in_half_width,
                               out_half_width, self.conv_kernel, filter_size, lrelu_upsampling, use_radial_filters, is_torgb,
                               is_critically_sampled)

    def modulation(self, w, global_w):
        """Style path of the layer (NET:349-352 + the small-tensor half of modulated_conv2d, NET:41-57): returns
        (w_hat, in_scale [N, Cin], out_scale [N, Cout] or None).  Depends on the latents only, not on the activations."""
        if self.cond_mod:
            w = torch.cat((w, global_w), 1)
        styles = self.affine(w)
        if self.is_torgb:
            styles = styles * self.styles_scale()
        return self.modulation_from_styles(styles)

    def styles_scale(self):
        """ToRGB's factor on the styles (NET:351); 1 for every other layer."""
        return float(1 / np.sqrt(self.in_channels * (self.conv_kernel ** 2))) if self.is_torgb else 1.0

    def modulation_from_styles(self, styles):
        """The weight-side half of modulation(): styles -> (w_hat, in_scale, out_scale).  SynthesisNetwork computes the styles of all
        its layers in one launch (torch_utils/ops/affine_bank.py) and enters here."""
        return modulation_coefficients_fused(self.weight, styles, demodulate=(not self.is_torgb), magnitude=self.magnitude_ema)

    def _act_args(self):
        return dict(up=self.up_factor, down=self.down_factor, padding=self.padding, gain=(1 if self.is_torgb else np.sqrt(2)),
                    slope=(1 if self.is_torgb else 0.2), clamp=self.conv_clamp)

    def fusable(self, x):
        """Can this call run as the single fused node (torch_utils/ops/fused_layer.py)?"""
        return fused_layer.available(x, self.weight, self.up_filter, self.down_filter, conv_pad=self.conv_kernel - 1, **self._act_args())

    def forward(self, x, w, global_w, E_features=None, include_skip=True, noise_mode='random', force_fp32=False, update_emas=False,
                _mod=None, _prescaled=False, _next_scale=None, _packed=None, _link_in=None, _link_out=None):
        """Reference signature (NET:336).  The underscore arguments are SynthesisNetwork's fusion hooks: `_mod` = this layer's
        precomputed modulation(), `_prescaled` = x already carries this layer's styles, `_next_scale` = the next layer's
        styles to fold into this layer's output (fused node only), `_packed` = the two MFMA images of this layer's normalised weight
        from the network's multi-layer pack (fused node only), `_link_in` / `_link_out` = the fused_layer.LayerLink shared with the layer
        before / after this one (x / the result have no other consumer)."""
        assert noise_mode in ['random', 'const', 'none']  # unused, as in the reference
        _assert_shape(x, [None, self.in_channels, int(self.in_size[1]), int(self.in_size[0])])
        _assert_shape(w, [x.shape[0], self.w_dim])
        if update_emas:
            assert not _prescaled, 'magnitude EMA needs the unscaled activations'
            with torch.autograd.profiler.record_function('update_magnitude_ema'):
                magnitude_cur = x.detach().to(torch.float32).square().mean()
                self.magnitude_ema.copy_(magnitude_cur.lerp(self.magnitude_ema, self.magnitude_ema_beta))
            _mod = None
        w_hat, in_scale, out_scale = _mod if _mod is not None else self.modulation(w, global_w)
        dtype = x.dtype
        x_skip = E_features[self.out_size[0]].to(dtype) if (E_features is not None and include_skip) else None
        act = self._act_args()
        if self.fusable(x):
            x = fused_layer.conv_filtered_lrelu(x, w_hat, in_scale, out_scale, self.bias, self.up_filter, self.down_filter,
                                                conv_pad=self.conv_kernel - 1, skip=x_skip, next_scale=_next_scale,
                                                prescaled=_prescaled, packed=_packed, link_in=_link_in, link_out=_link_out, **act)
        else:
            assert _next_scale is None, 'only the fused node can pre-scale its output'
            with torch.autograd.profiler.record_function('modulated_conv2d'):
                x = scaled_conv2d(x, w_hat, in_scale, out_scale, self.conv_kernel - 1, prescaled=_prescaled, link_in=_link_in)
            x = filtered_lrelu.filtered_lrelu(x=x, fu=self.up_filter, fd=self.down_filter, b=self.bias.to(x.dtype), **act)
            if x_skip is not None:
                x = x + x_skip
        _assert_shape(x, [None, self.out_channels, int(self.out_size[1]), int(self.out_size[0])])
        assert x.dtype == dtype
        return x

    def extra_repr(self):
        return (f'w_dim={self.w_dim:d}, is_torgb={self.is_torgb}, in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, '
                f'in_size={list(self.in_size)}, out_size={list(self.out_size)}, up={self.up_factor}, down={self.down_factor}')


class EncoderLayer(torch.nn.Module, _ResampleGeometry):
    """Plain conv + filtered leaky ReLU of the alias-free encoder (NET:417-549)."""

    def __init__(self, is_critically_sampled, use_fp16, in_channels, out_channels, in_size, out_size, in_sampling_rate,
                 out_sampling_rate, in_cutoff, out_cutoff, in_half_width, out_half_width, conv_kernel=3, filter_size=6,
                 lrelu_upsampling=1, use_radial_filters=False, conv_clamp=256, magnitude_ema_beta=0.999, cond_mod=False):
        super().__init__()
        self.is_critically_sampled = is_critically_sampled
        self.use_fp16 = use_fp16
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_kernel = conv_kernel
        self.conv_clamp = conv_clamp
        self.magnitude_ema_beta = magnitude_ema_beta
        self.weight = torch.nn.Parameter(torch.randn([self.out_channels, self.in_channels, self.conv_kernel, self.conv_kernel]))
        self.weight_gain = 1 / np.sqrt(in_channels * (conv_kernel ** 2))
        self._gain_planes = {}
        self.bias = torch.nn.Parameter(torch.zeros([self.out_channels]))
        self.register_buffer('magnitude_ema', torch.ones([]))
        self._setup_resampling(in_size, out_size, in_sampling_rate, out_sampling_rate, in_cutoff, out_cutoff, in_half_width,
                               out_half_width, conv_kernel, filter_size, lrelu_upsampling, use_radial_filters, False,
                               is_critically_sampled)

    def forward(self, x, force_fp32=False, update_emas=False, _packed=None):
        _assert_shape(x, [None, self.in_channels, int(self.in_size[1]), int(self.in_size[0])])
        if update_emas:
            with torch.autograd.profiler.record_function('update_magnitude_ema'):
                magnitude_cur = x.detach().to(torch.float32).square().mean()
                self.magnitude_ema.copy_(magnitude_cur.lerp(self.magnitude_ema, self.magnitude_ema_beta))
        dtype = x.dtype
        act = dict(up=self.up_factor, down=self.down_factor, padding=self.padding, gain=np.sqrt(2), slope=0.2, clamp=self.conv_clamp)
        if fused_layer.available(x, self.weight, self.up_filter, self.down_filter, conv_pad=self.conv_kernel - 1, **act):
            # conv (+ bias in its epilogue) -> filtered_lrelu as one autograd node (16-bit activations); the equalised-lr
            # weight gain (NET:503) rides in the conv epilogue as a constant per-plane factor instead of a pass over w
            key = (int(x.shape[0]), x.device)
            if key not in self._gain_planes:
                self._gain_planes[key] = torch.full([x.shape[0], self.out_channels], float(self.weight_gain), dtype=torch.float32, device=x.device)
            x = fused_layer.conv_filtered_lrelu(x, self.weight, None, self._gain_planes[key], self.bias, self.up_filter, self.down_filter,
                                                conv_pad=self.conv_kernel - 1, packed=_packed, **act)
        else:
            w = self.weight * self.weight_gain
            x = conv2d_gradfix.conv2d(input=x, weight=w, padding=self.conv_kernel - 1)
            x = filtered_lrelu.filtered_lrelu(x=x, fu=self.up_filter, fd=self.down_filter, b=self.bias.to(x.dtype), **act)
        _assert_shape(x, [None, self.out_channels, int(self.out_size[1]), int(self.out_size[0])])
        assert x.dtype == dtype
        return x

    def extra_repr(self):
        return (f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, in_size={list(self.in_size)}, '
                f'out_size={list(self.out_size)}, up={self.up_factor}, down={self.down_factor}')


class Conv2dLayer(torch.nn.Module):
    """The bottleneck's 3x3 conv + bias + activation (`e_16x16`, NET:635): models/networks/CoModGAN/layers.py:115-162.  The generator
    uses it without resampling (up = down = 1: straight onto the MFMA conv); with up / down it takes the reference's decomposition
    (torch_utils/ops/conv2d_resample.py: blur + strided / transposed conv, the route the discriminator's layers use)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        from .torch_utils.ops import upfirdn2d
        self.activation = activation
        self.up, self.down = up, down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
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
        if self.up == 1 and self.down == 1:
            x = conv2d_gradfix.conv2d(input=x, weight=w, padding=self.padding)
        else:
            from .torch_utils.ops import conv2d_resample
            x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                                padding=self.padding, flip_weight=(self.up == 1))          # layers.py:156-157
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=act_gain, clamp=act_clamp)


class SynthesisNetwork(torch.nn.Module):
    """Alias-free encoder (14 layers) -> global vector -> co-modulated alias-free decoder (15 layers) (NET:556-712)."""

    def __init__(self, w_dim, img_resolution, img_channels_in, img_channels_out, channel_base=32768, channel_max=512, num_layers=14,
                 num_critical=2, first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, margin_size=10,
                 output_scale=0.25, num_fp16_res=4, dropout_rate=0.5, skip_resolution=256, compute_dtype=torch.float32,
                 **layer_kwargs):
        super().__init__()
        self.w_dim = w_dim
        self.num_ws = num_layers + 2
        self.img_resolution = img_resolution
        self.img_channels_in, self.img_channels_out = img_channels_in, img_channels_out
        self.num_layers, self.num_critical = num_layers, num_critical
        self.margin_size = margin_size
        self.output_scale = output_scale
        self.num_fp16_res = num_fp16_res
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.compute_dtype = compute_dtype

        if skip_resolution >= 4:
            final_skip = int(np.log2(skip_resolution))
            self.skip_connects = [True] * (final_skip - 1) + [False] * (self.img_resolution_log2 - final_skip)
        else:
            self.skip_connects = [False] * self.img_resolution_log2

        bs = sched.band_schedule(img_resolution, img_channels_out, num_layers, num_critical, first_cutoff, first_stopband,
                                 last_stopband_rel, margin_size, channel_base, channel_max)
        cutoffs, half_widths, sampling_rates = bs['cutoffs'], bs['half_widths'], bs['sampling_rates']
        sizes, sizes_for_encoder = bs['sizes'], bs['sizes_for_encoder']
        self.sizes = sizes
        self.channels = channels = bs['channels']

        for idx in range(num_layers):
            rev_idx = num_layers - idx - 1
            rev_prev = num_layers - max(idx - 1, 0) - 1
            layer = EncoderLayer(
                is_critically_sampled=(idx < num_layers - num_critical), use_fp16=False,
                in_channels=img_channels_in if idx == 0 else int(channels[rev_prev]), out_channels=int(channels[rev_idx]),
                in_size=int(sizes_for_encoder[rev_prev]), out_size=int(sizes_for_encoder[rev_idx]),
                in_sampling_rate=int(sampling_rates[rev_prev]), out_sampling_rate=int(sampling_rates[rev_idx]),
                in_cutoff=cutoffs[rev_prev], out_cutoff=cutoffs[rev_idx],
                in_half_width=half_widths[rev_prev], out_half_width=half_widths[rev_idx], **layer_kwargs)
            setattr(self, f'encoder_{idx}', layer)

        self.e_16x16 = Conv2dLayer(int(channels[0]), int(channels[0]), kernel_size=3, activation='lrelu', conv_clamp=None)
        self.pool = torch.nn.AdaptiveAvgPool2d((4, 4))
        self.fc_in = FullyConnectedLayer(int(channels[0]) * (4 ** 2), 512 * 2, activation='lrelu')
        self.dropout = torch.nn.Dropout(p=dropout_rate)

        self.layer_names = []
        for idx in range(num_layers + 1):
            prev = max(idx - 1, 0)
            layer = SynthesisLayer(
                w_dim=self.w_dim, global_w_dim=512 * 2, is_torgb=(idx == num_layers),
                is_critically_sampled=(idx >= num_layers - num_critical), use_fp16=False,
                in_channels=int(channels[prev]), out_channels=int(channels[idx]),
                in_size=int(sizes[prev]), out_size=int(sizes[idx]),
                in_sampling_rate=int(sampling_rates[prev]), out_sampling_rate=int(sampling_rates[idx]),
                in_cutoff=cutoffs[prev], out_cutoff=cutoffs[idx],
                in_half_width=half_widths[prev], out_half_width=half_widths[idx], **layer_kwargs)
            name = f'L{idx}_{layer.out_size[0]}_{layer.out_channels}'
            setattr(self, name, layer)
            self.layer_names.append(name)

    def _pool4(self, x):
        """AdaptiveAvgPool2d((4, 4)) (NET:636,683).  When the plane divides evenly (36 -> 9x9 bins at every shipped
        resolution) the adaptive bins are plain blocks: a reshape + mean, whose backward is a broadcast instead of the
        atomic scatter kernel of adaptive_avg_pool2d_backward (0.38 ms per step at batch 16)."""
        h, w = x.shape[-2:]
        if h % 4 == 0 and w % 4 == 0:
            if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16, torch.float32):
                return _PoolBlocks.apply(x)                      # fp32 block means straight from the activations: one launch each way
            x = x.to(torch.float32)
            return x.reshape(x.shape[0], x.shape[1], 4, h // 4, 4, w // 4).mean(dim=(3, 5))
        return self.pool(x.to(torch.float32))

    def forward(self, ws, img_in, **layer_kwargs):
        _assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws_stack = ws
        ws = ws.to(torch.float32).unbind(dim=1)
        img_in = F.pad(img_in.to(self.compute_dtype), [self.margin_size] * 4, 'constant', 0)

        # every conv weight of the step exists before its first convolution: the 16-bit MFMA images (forward + data gradient) of all
        # 3x3 layers from one launch per half of the network (conv2d.pack_weights_bank) instead of one 8 us launch per layer
        bank16 = self.compute_dtype in (torch.bfloat16, torch.float16) and img_in.is_cuda
        enc_layers = [getattr(self, f'encoder_{idx}') for idx in range(self.num_layers)]
        enc_packs = [None] * len(enc_layers)
        if bank16 and all(layer.conv_kernel == 3 for layer in enc_layers) and len(enc_layers) <= _conv_ops.PACK_MAX:
            # (the data-gradient images only where a backward can follow)
            enc_packs = _conv_ops.pack_weights_bank([layer.weight for layer in enc_layers], self.compute_dtype, need_dgrad=torch.is_grad_enabled())

        E_features = {}
        for idx in range(self.num_layers):
            rev_idx = self.num_layers - idx - 1
            rev_prev = self.num_layers - max(idx - 1, 0) - 1
            img_in = enc_layers[idx](img_in, _packed=enc_packs[idx])   # the reference passes no kwargs to the encoder (NET:678)
            if (self.sizes[rev_idx] != self.sizes[rev_prev]) and self.sizes[rev_prev] != self.sizes[0]:
                # the feature map has two consumers (the next encoder layer, a decoder layer's skip input): fork it, so that its two
                # gradients meet in one fused pass (torch_utils/ops/fused_layer.py skip_fork)
                img_in, E_features[self.sizes[rev_idx]] = fused_layer.skip_fork(img_in)

        img_pool = self.e_16x16(img_in)
        img_pool = self._pool4(img_pool)                         # (fp32 out)
        img_pool = self.fc_in(img_pool.flatten(1))
        img_global = self.dropout(img_pool)

        x = img_in
        res_idx = 1
        # Every layer's styles depend on the latents only: compute them up front, so that a layer running as the fused node
        # can fold the NEXT layer's style factor into its own output epilogue (no separate pass over the activations).
        fuse = not layer_kwargs.get('update_emas', False)
        layers = [getattr(self, name) for name in self.layer_names]
        mods = None
        if fuse and all(layer.cond_mod for layer in layers):
            # every layer's styles = affine_l(cat(ws[:, 1 + l], img_global)) from ONE launch (and three backward) instead of 15 cat +
            # 15 GEMM (and 30 GEMM + 15 reductions + 14 accumulations): torch_utils/ops/affine_bank.py
            specs = [affine_bank.Spec(layer.affine, 1 + l, layer.styles_scale()) for l, layer in enumerate(layers)]
            ws_all = ws_stack.to(torch.float32)
            g32 = img_global.to(torch.float32).contiguous()
            if affine_bank.supported(ws_all, g32, specs):
                styles = affine_bank.affine_bank(ws_all, g32, specs)
                # ... and every layer's (w_hat, in_scale, out_scale) from two (three backward): torch_utils/ops/modulation_bank.py
                items = [modulation_bank.Item(layer.weight, st, layer.magnitude_ema, not layer.is_torgb) for layer, st in zip(layers, styles)]
                if modulation_bank.supported(items):
                    mods = modulation_bank.modulation_bank(items)
                else:
                    mods = [layer.modulation_from_styles(st) for layer, st in zip(layers, styles)]
        if mods is None:
            mods = [layer.modulation(w, img_global) for layer, w in zip(layers, ws[1:])] if fuse else [None] * len(layers)
        dec_packs = [None] * len(layers)
        if bank16 and fuse and mods[0] is not None:
            k3 = [i for i, layer in enumerate(layers) if layer.conv_kernel == 3]
            if 0 < len(k3) <= _conv_ops.PACK_MAX:
                for i, pk in zip(k3, _conv_ops.pack_weights_bank([mods[i][0] for i in k3], self.compute_dtype, need_dgrad=torch.is_grad_enabled())):
                    dec_packs[i] = pk
        prescaled = False
        links = [fused_layer.LayerLink() for _ in layers]        # links[i]: between layer i and layer i + 1 (its only consumer)
        for idx, (layer, w) in enumerate(zip(layers, ws[1:])):
            nxt = min(idx + 1, len(self.layer_names) - 1)
            if (self.sizes[idx] != self.sizes[nxt]) and self.sizes[idx] != self.sizes[0]:
                include_skip = self.skip_connects[res_idx]
                res_idx += 1
            else:
                include_skip = False
            next_scale = mods[idx + 1][1] if (fuse and idx + 1 < len(layers) and layer.fusable(x)) else None
            x = layer(x, w, img_global, E_features, include_skip, _mod=mods[idx], _prescaled=prescaled, _next_scale=next_scale,
                      _packed=dec_packs[idx], _link_in=(links[idx - 1] if prescaled else None),
                      _link_out=(links[idx] if next_scale is not None else None), **layer_kwargs)
            prescaled = next_scale is not None
        _assert_shape(x, [None, self.img_channels_out, self.img_resolution, self.img_resolution])
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16):
            return _ScaleCast.apply(x, self.output_scale, torch.float32)             # the multiply and the cast as one pass (and one backward)
        if self.output_scale != 1:
            x = x * self.output_scale
        return x.to(torch.float32)

    def extra_repr(self):
        return (f'w_dim={self.w_dim:d}, num_ws={self.num_ws:d}, img_resolution={self.img_resolution:d}, '
                f'num_layers={self.num_layers:d}, num_critical={self.num_critical:d}, margin_size={self.margin_size:d}, '
                f'compute_dtype={self.compute_dtype}')


class Stylegan3Generator(torch.nn.Module):
    """Mapping + synthesis; ``forward(z, c, cond_img, ref_img=None, truncation_psi=1, truncation_cutoff=None,
    update_emas=False, **synthesis_kwargs)`` as NET:717-740."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels_in, img_channels_out, mapping_kwargs={}, synthesis_kwargs={}):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution = img_resolution
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels_in=img_channels_in,
                                          img_channels_out=img_channels_out, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, cond_img, ref_img=None, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, img_in=ref_img, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, cond_img, update_emas=update_emas, **synthesis_kwargs)
